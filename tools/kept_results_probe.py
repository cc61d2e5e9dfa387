"""Whole SparseVFC calls of one C5 organ with the caller KEEPING its results (every array of the next call at a fresh address) and
discarding them: the pageable-upload pathology (tools/, DESIGN section 2.2 "what the host costs")."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, "spateo-release_amd")]
import numpy as np, torch
import spateo_amd as st
from spateo_amd._synthetic import make_config
X, V, _ = make_config("C2", N=250_000, seed=101)
kw = dict(M=500, lambda_=0.02, lstsq_method="scipy", seed=0, MaxIter=30, dtype="float32", device="cuda:0")
st.SparseVFC(X, V, None, **kw); st.SparseVFC(X, V, None, **kw)
for mode in ("discarding", "keeping"):
    keep, ts = [], []
    for rep in range(12):
        t0 = time.perf_counter(); r = st.SparseVFC(X, V, None, **kw); ts.append(1e3 * (time.perf_counter() - t0))
        if mode == "keeping":
            keep.append(r)
    print(f"{mode} results: {[round(t, 1) for t in ts]} ms")
# fresh input arrays per call (what a loop over organs does)
ts = []
for rep in range(8):
    Xn, Vn, _ = make_config("C2", N=250_000, seed=200 + rep)
    t0 = time.perf_counter(); r = st.SparseVFC(Xn, Vn, None, **kw); ts.append(1e3 * (time.perf_counter() - t0))
print(f"fresh inputs per call: {[round(t, 1) for t in ts]} ms")
