"""Developer probe (GPU): BASELINE config 2 - fit N=50k, M=500 (float32), then Jacobian + curl (+ all evaluators) on a
64^3 grid; also times apply / rhs / E-step kernels at that size.  Prints achieved pair-rate and VALU-roofline fraction."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
import spateo_amd as st
from spateo_amd import _lib
from spateo_amd._kernels import HipKernels
from spateo_amd._synthetic import make_config

X, V, M = make_config("C2")
t0 = time.perf_counter()
vf = st.SparseVFC(X, V, None, M=M, lambda_=0.02, lstsq_method="scipy", dtype="float32", device="cuda:0", MaxIter=50)
torch.cuda.synchronize()
print(f"C2 fit: N={len(X)} M={M} iterations={vf['iteration']+1} wall={time.perf_counter()-t0:.2f}s sigma2={vf['sigma2']:.4g}")
g = np.linspace(-1, 1, 64)
G = np.stack(np.meshgrid(g * 303, g * 202, g * 151.5), -1).reshape(-1, 3)
for dtype in ("float32", "float64"):
    k = HipKernels("cuda:0", dtype)
    c = vf["X_ctrl"].mean(0)
    x4, c4 = k.to_x4(G, c), k.to_x4(vf["X_ctrl"], c)
    Cd = torch.from_numpy(np.ascontiguousarray(vf["C"])).cuda()
    for name, flags in (("jac+curl", _lib.EVAL_JAC | _lib.EVAL_CURL), ("all", 255)):
        k.eval(x4, c4, vf["beta"], Cd, flags); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): k.eval(x4, c4, vf["beta"], Cd, flags)
        e1.record(); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        pairs = len(G) * M
        print(f"eval[{dtype}] {name:9s} 64^3 x {M}: {ms:.3f} ms  {pairs/ms/1e6:.1f} Gpair/s  (~{pairs*38/ms/1e9:.1f} TFLOP/s at 38 flop/pair)")
