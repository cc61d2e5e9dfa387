"""Developer probe (GPU): where the wall time of the coefficient solve goes inside an EM iteration at M control points - the C call
(mvf_solve_minnorm_lrd, host-synchronous), the Python around it (_solve_all: wrapper, status read), against the kernel phases the
library itself reports (developer option lr_timing, separate iterations).

    python tools/solve_wall_probe.py M N [steps] [dtype]
"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
from spateo_amd import _lib
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess

M, N = int(sys.argv[1]), int(sys.argv[2])
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
dtype = sys.argv[4] if len(sys.argv) > 4 else "float32"
X, Y, _ = make_config("C4" if M >= 3000 else "C2", N=N)
valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, Y, M=M, seed=0)
eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype=dtype, device="cuda:0")
eng.init_state()
k = eng.k
t_c, t_all = [], []
inner_c, inner_all = k.solve_minnorm_lr, eng._solve_all

def timed_c(*a, **kw):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = inner_c(*a, **kw); t_c.append(time.perf_counter() - t0); return r

def timed_all(ls2):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = inner_all(ls2); t_all.append(time.perf_counter() - t0); return r

k.solve_minnorm_lr, eng._solve_all = timed_c, timed_all
for i in range(steps):
    eng.em_step(lambda_=0.02)
torch.cuda.synchronize()
print(f"M = {M}, N = {N}, {dtype}: _solve_all wall (last 5) {[round(1e3 * t, 3) for t in t_all[-5:]]} ms; the C call inside it "
      f"{[round(1e3 * t, 3) for t in t_c[-5:]]} ms; rank {eng.solver_stats['rank'][-3:]}, block {eng.solver_stats.get('block', [])[-3:]}")
k.solve_minnorm_lr, eng._solve_all = inner_c, inner_all
_lib.debug_option("lr_timing", 1)
for i in range(2):
    eng.em_step(lambda_=0.02)
torch.cuda.synchronize()
