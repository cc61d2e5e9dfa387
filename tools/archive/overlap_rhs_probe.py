"""A/B: rhs kernels on a side stream next to the Gram tile stage (HipKernels.overlap_rhs) vs behind it; 8 M x 3000 float32."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
from spateo_amd._kernels import HipKernels
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8_000_000
X, V, M = make_config("C4", N=n)
valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=M, seed=0)
out = {}
for dtype in ("float32", "float64"):
    res = {}
    for overlap in (False, True, False, True):
        k = HipKernels("cuda:0", dtype); k.overlap_rhs = overlap
        eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype=dtype, device="cuda:0", kernels=k)
        eng.init_state(0.9)
        for _ in range(2): eng.em_step(lambda_=0.02)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(4): eng.em_step(lambda_=0.02)
        torch.cuda.synchronize(); ms = 1e3 * (time.perf_counter() - t0) / 4
        res.setdefault(overlap, []).append(ms)
        key = (eng.G.cpu().numpy().tobytes(), eng.R[0].cpu().numpy().tobytes(), eng.sigma2)
        assert res.setdefault("bits", key) == key, "overlap changed the result"
        k.drop_ublk(); del eng, k; torch.cuda.empty_cache()
    out[dtype] = {"behind_ms": res[False], "overlapped_ms": res[True]}
print(json.dumps(out))
