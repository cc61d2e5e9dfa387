"""Host profile of ONE C5 organ (250 k cells, M = 500): sparsevfc_preprocess and the whole SparseVFC call, single thread."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
import spateo_amd as st
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import sparsevfc_preprocess
X, V, _ = make_config("C2", N=250_000, seed=101)
kw = dict(M=500, lambda_=0.02, MaxIter=30, dtype="float32", device="cuda:0")
st.SparseVFC(X, V, None, **kw)
for rep in range(3):
    t0 = time.perf_counter(); sparsevfc_preprocess(X, V, M=500, seed=0, device="cuda:0"); t1 = time.perf_counter()
    st.SparseVFC(X, V, None, **kw); t2 = time.perf_counter()
    print(f"preprocess {1e3*(t1-t0):.2f} ms, whole call {1e3*(t2-t1):.2f} ms")
for what, f in (("preprocess", lambda: sparsevfc_preprocess(X, V, M=500, seed=0, device="cuda:0")), ("whole call", lambda: st.SparseVFC(X, V, None, **kw))):
    pr = cProfile.Profile(); pr.enable(); f(); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(16); print("====", what); print(s.getvalue()[:3800])
