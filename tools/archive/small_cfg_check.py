"""Developer probe (GPU): bench.py's small_configs measurement, stand-alone, with the per-step solve times listed."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "spateo-release_amd"))
import numpy as np
import torch

from spateo_amd import _lib
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess

step_kw = dict(a=5.0, lambda_=0.02, minP=1e-5, theta=0.75)


def small(cfg_name, n_small, m_small, seed_small, timing=False):
    Xs, Vs, _ = make_config(cfg_name, N=n_small, seed=seed_small)
    _, Xsv, Ysv, _, ctrl_s, beta_s = sparsevfc_preprocess(Xs, Vs, M=m_small, seed=0)
    eng = SparseVFCEngine(Xsv, Ysv, ctrl_s, beta_s, dtype="float32", device="cuda:0")
    eng.init_state(gamma=0.9)
    evs, inner = [], eng._solve_all

    def timed(ls2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        h_ = inner(ls2)
        e1.record()
        evs.append((e0, e1))
        return h_

    eng._solve_all = timed
    if timing:
        _lib.debug_option("lr_timing", 1)
    walls = []
    for _ in range(34):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.em_step(**step_kw)
        torch.cuda.synchronize()
        walls.append(1e3 * (time.perf_counter() - t0))
    _lib.debug_option("lr_timing", 0)
    print(cfg_name, n_small, "solve ms:", [round(a.elapsed_time(b), 2) for a, b in evs], file=sys.stderr)
    print(cfg_name, n_small, "step ms:", [round(w, 2) for w in walls], file=sys.stderr)
    eng.k.drop_ublk()


small("C2", 50_000, 500, 2, timing=len(sys.argv) > 1)
small("C2", 250_000, 500, 100)
small("C2", 50_000, 500, 2)
