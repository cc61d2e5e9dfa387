"""Where the FIRST EM iteration of a C5 organ (250 k x 500) spends its host time (engine construction + em_step #1)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess
X, V, _ = make_config("C2", N=250_000, seed=101)
valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=500, seed=0, device="cuda:0")
def one(profile=False):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype="float32", device="cuda:0")
    eng.init_state(gamma=0.9)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    eng.em_step(a=5, lambda_=0.02, minP=1e-5, theta=0.75)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    eng.em_step(a=5, lambda_=0.02, minP=1e-5, theta=0.75)
    torch.cuda.synchronize(); t3 = time.perf_counter()
    return 1e3 * (t1 - t0), 1e3 * (t2 - t1), 1e3 * (t3 - t2)
for rep in range(4):
    print("engine+init %.2f ms, step 1 %.2f ms, step 2 %.2f ms" % one())
pr = cProfile.Profile(); pr.enable(); one(); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(22); print(s.getvalue()[:5000])
