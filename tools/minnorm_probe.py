"""Timing probe of mvf_solve_minnorm (hand-written symmetric eigensolver + truncated solve) inside the EM loop:
python tools/minnorm_probe.py M [N] [steps] -> per EM step: solve ms, Jacobi sweeps, kept rank (cold first, then warm)."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess

M = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 20 * M
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
shift = 2.0 ** -int(sys.argv[4]) if len(sys.argv) > 4 else None
X, Y, _ = make_config("C2", N=N)
valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, Y, M=M, seed=0)
out = {"M": M, "N": N}
for warm in (True, False):
    eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype="float64", device="cuda:0")
    eng.warm_start = warm
    if shift: eng.mn_shift = shift
    eng.init_state()
    orig = eng._solve_all
    ms = []
    def timed(ls2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); h = orig(ls2); torch.cuda.synchronize(); ms.append(1e3 * (time.perf_counter() - t0)); return h
    eng._solve_all = timed
    for _ in range(steps):
        eng.em_step(lambda_=0.02)
    out["warm" if warm else "cold"] = {"solve_ms": [round(x, 2) for x in ms], "sweeps": eng.solver_stats["sweeps"],
                                       "rank": eng.solver_stats["rank"], "cholesky_steps": eng.solver_stats["cholesky"],
                                       "sigma2": eng.sigma2}
    V = eng.results()[0]
    out.setdefault("V", []).append(V)
    del eng
    torch.cuda.empty_cache()
Vw, Vc = out.pop("V")
out["warm_vs_cold_field_maxrel"] = float(np.abs(Vw - Vc).max() / np.abs(Vc).max())
print(json.dumps(out), flush=True)
