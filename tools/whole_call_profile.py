"""Whole SparseVFC calls (host arrays in, host dict out) at BASELINE config 2 and one C5 organ: wall time, the phases of
`spateo_amd._runtime._Phases` (preprocessing / upload + U cache / EM / download) and a host profile."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
import spateo_amd as st
from spateo_amd import _runtime as rt
from spateo_amd._synthetic import make_config
for n, seed in ((50_000, 0), (250_000, 101)):
    X, V, _ = make_config("C2", N=n, seed=seed)
    kw = dict(M=500, lambda_=0.02, lstsq_method="scipy", seed=0, MaxIter=30, dtype="float32", device="cuda:0")
    st.SparseVFC(X, V, None, **kw)
    ts = []
    for rep in range(5):
        t0 = time.perf_counter(); r = st.SparseVFC(X, V, None, **kw); ts.append(time.perf_counter() - t0)
    rt.PROFILE_FITS = True
    st.SparseVFC(X, V, None, **kw)
    ph = dict(rt._TLS.fit_profile)
    rt.PROFILE_FITS = False
    print(f"n = {n}: whole call {1e3 * min(ts):.2f} ms (min of 5; all {[round(1e3 * t, 2) for t in ts]}), {int(r['iteration']) + 1} EM iterations; "
          f"phases (synchronised, ms): " + ", ".join(f"{k} {1e3 * v:.2f}" if k.endswith('_s') else f"{k} {v}" for k, v in ph.items()))
    pr = cProfile.Profile(); pr.enable(); st.SparseVFC(X, V, None, **kw); pr.disable()
    s = io.StringIO(); pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(14); print(s.getvalue()[:3000])
