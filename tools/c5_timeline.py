"""Per-thread phase timeline of SparseVFC_many on 4 organs x 4 streams (where do the 0.17 - 0.22 s go?)."""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np, torch
from spateo_amd._synthetic import make_config
import spateo_amd.vectorfield as vfm
ev = []
def wrap(obj, name, tag):
    f = getattr(obj, name)
    def g(*a, **k):
        t0 = time.perf_counter(); r = f(*a, **k); ev.append((threading.get_ident() % 1000, tag, t0, time.perf_counter())); return r
    setattr(obj, name, g)
wrap(vfm, "sparsevfc_preprocess", "preprocess")
wrap(vfm.SparseVFCEngine, "__init__", "engine_init")
wrap(vfm.SparseVFCEngine, "fit", "fit")
wrap(vfm.SparseVFCEngine, "results", "results")
organs = [(*make_config("C2", N=250_000, seed=100 + s)[:2], None) for s in range(4)]
kw = dict(M=500, lambda_=0.02, MaxIter=30, dtype="float32")
vfm.SparseVFC_many(organs[:1], n_streams=1, **dict(kw, MaxIter=2))
for rep in range(2):
    ev.clear()
    T0 = time.perf_counter(); vfm.SparseVFC_many(organs, n_streams=4, **kw); T1 = time.perf_counter()
    print(f"rep {rep}: wall {1e3*(T1-T0):.1f} ms")
    for th, tag, a, b in sorted(ev, key=lambda e: e[2]):
        print(f"   thread {th:3d} {tag:12s} {1e3*(a-T0):7.1f} -> {1e3*(b-T0):7.1f} ms ({1e3*(b-a):6.1f})")
