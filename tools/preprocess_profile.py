"""Host profile of sparsevfc_preprocess at 8 M cells x 3000 control points (what is left on the host)."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import sparsevfc_preprocess
X, V, M = make_config("C4")
sparsevfc_preprocess(X[:300000], V[:300000], M=M, seed=0, device="cuda:0")
for rep in range(2):
    t0 = time.perf_counter(); out = sparsevfc_preprocess(X, V, M=M, seed=0, device="cuda:0"); print(f"preprocess 8 M: {time.perf_counter()-t0:.3f} s")
pr = cProfile.Profile(); pr.enable(); sparsevfc_preprocess(X, V, M=M, seed=0, device="cuda:0"); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3500])
