"""Timing probe of the minimum-norm coefficient solve inside the EM loop:
python tools/minnorm_probe.py M [N] [steps] [lambda] -> per EM step: solve ms, Jacobi sweeps, kept rank, factor rank for
mn_method = "lowrank" (mvf_solve_minnorm_lr: pivoted-Cholesky factor, Jacobi on its r columns) and "full"
(mvf_solve_minnorm: all M columns of the shifted factor, warm-started), and the field deviation between the two."""
import json, os, sys, time
import numpy as np
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess

M = int(sys.argv[1]); N = int(sys.argv[2]) if len(sys.argv) > 2 else 20 * M
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 8
lam = float(sys.argv[4]) if len(sys.argv) > 4 else 0.02
methods = sys.argv[5].split(",") if len(sys.argv) > 5 else ["lowrank", "full"]
X, Y, _ = make_config("C2", N=N)
valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, Y, M=M, seed=0)
out = {"M": M, "N": N, "lambda": lam}
fields = {}
for method in methods:
    eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype="float64", device="cuda:0")
    eng.mn_method = method
    eng.init_state()
    orig = eng._solve_all
    ms = []
    def timed(ls2):
        torch.cuda.synchronize(); t0 = time.perf_counter(); h = orig(ls2); torch.cuda.synchronize(); ms.append(1e3 * (time.perf_counter() - t0)); return h
    eng._solve_all = timed
    for _ in range(steps):
        eng.em_step(lambda_=lam)
    out[method] = {"solve_ms": [round(x, 2) for x in ms], "sweeps": eng.solver_stats["sweeps"],
                   "rank": eng.solver_stats["rank"], "factor_rank": eng.solver_stats.get("factor_rank"),
                   "cholesky_steps": eng.solver_stats["cholesky"], "sigma2": eng.sigma2}
    fields[method] = eng.results()[0]
    del eng
    torch.cuda.empty_cache()
if len(fields) == 2:
    a, b = (fields[m] for m in methods)
    out["field_maxrel_between_methods"] = float(np.abs(a - b).max() / np.abs(b).max())
print(json.dumps(out), flush=True)
