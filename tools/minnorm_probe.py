"""Timing probe of mvf_solve_minnorm (hand-written symmetric eigensolver + truncated solve) on rank-deficient
SparseVFC systems.  python tools/minnorm_probe.py M [N]  ->  one line of JSON per repetition."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
from spateo_amd._kernels import HipKernels
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess

M = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 20 * M
X, Y, _ = make_config("C2", N=N)
valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, Y, M=M, seed=0)
eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype="float64", device="cuda:0")
eng.init_state()
for _ in range(3):
    eng.em_step(lambda_=0.02)
k = eng.k
ls2 = 0.02 * eng.sigma2
C = torch.empty_like(eng.R[0])
for rep in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    k.solve_minnorm(eng.G, eng.K, ls2, 2.0 ** -36, eng.R[0], C, eng.info, eng.einfo)
    torch.cuda.synchronize(); t_mn = time.perf_counter() - t0
    t0 = time.perf_counter()
    k.solve(eng.G, eng.K, ls2, 1e-12, eng.R[0], C, eng.info, eng.pivots)
    torch.cuda.synchronize(); t_ch = time.perf_counter() - t0
    e = eng.einfo.cpu().numpy(); pv = eng.pivots.cpu().numpy()
    print(json.dumps({"M": M, "N": N, "minnorm_ms": 1e3 * t_mn, "sweeps": e[0], "rank": e[1], "lmax": e[2], "min_kept": e[3],
                      "delta": e[4], "lmin": e[5], "cholesky_ms": 1e3 * t_ch, "piv_ratio": pv[0] / pv[1],
                      "solver_stats": {a: (b if not isinstance(b, list) else b[-3:]) for a, b in eng.solver_stats.items()}}), flush=True)
