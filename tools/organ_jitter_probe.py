"""Whole calls of one C5 organ (250 k x 500) twenty times with the phase profile on: which phase carries the occasional + 25 ms?"""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path[:0] = [R, os.path.join(R, "spateo-release_amd")]
import numpy as np, torch
import spateo_amd as st
from spateo_amd import _runtime as rt
from spateo_amd._synthetic import make_config
import spateo_amd.preprocess as pre
X, V, _ = make_config("C2", N=250_000, seed=101)
kw = dict(M=500, lambda_=0.02, lstsq_method="scipy", seed=0, MaxIter=30, dtype="float32", device="cuda:0")
st.SparseVFC(X, V, None, **kw)
rt.PROFILE_FITS = True
for rep in range(16):
    t0 = time.perf_counter(); st.SparseVFC(X, V, None, **kw); t = time.perf_counter() - t0
    ph = rt._TLS.fit_profile
    print(f"{1e3 * t:7.2f} ms: " + ", ".join(f"{k[:-2]} {1e3 * v:.2f}" for k, v in ph.items() if k.endswith("_s")))
rt.PROFILE_FITS = False
# inside the preprocessing
import cProfile, pstats, io
for rep in range(6):
    t0 = time.perf_counter(); pre.sparsevfc_preprocess(X, V, M=500, seed=0, device="cuda:0"); t = time.perf_counter() - t0
    print(f"preprocess alone {1e3 * t:.2f} ms")
pr = cProfile.Profile()
for rep in range(8):
    pr.enable(); pre.sparsevfc_preprocess(X, V, M=500, seed=0, device="cuda:0"); pr.disable()
s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(12); print(s.getvalue()[:2500])
