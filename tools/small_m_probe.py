#!/usr/bin/env python
"""Developer probe (GPU): the coefficient solve at M <= 640 control points (BASELINE configs 2 and 5) - the full-width
warm-started Jacobi eigensolver against the deflated solve with its 64-vector block (round 5), on the same fits.

    python tools/small_m_probe.py [--out gpurun_out/r05_small_m_probe.json]
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "spateo-release_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)


def run(X, V, M, dtype, method, steps=12, lambda_=0.02, timing=False):
    import torch
    from spateo_amd import _lib
    from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess

    valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=M, seed=0)
    SparseVFCEngine.minnorm_method = method
    try:
        eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype=dtype, device="cuda:0")
    finally:
        SparseVFCEngine.minnorm_method = None
    eng.init_state(0.9)
    kw = dict(a=5.0, lambda_=lambda_, minP=1e-5, theta=0.75)
    ev, inner = [], eng._solve_all

    def timed(ls2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        h = inner(ls2)
        e1.record()
        ev.append((e0, e1))
        return h

    eng._solve_all = timed
    walls = []
    for i in range(steps):
        if timing and i == steps - 1:
            _lib.debug_option("lr_timing", 1)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        eng.em_step(**kw)
        torch.cuda.synchronize()
        walls.append(1e3 * (time.perf_counter() - t0))
    _lib.debug_option("lr_timing", 0)
    solve = [a.elapsed_time(b) for a, b in ev]
    st = eng.solver_stats
    Vg = eng.results()[0]
    rec = dict(cells=len(Xv), ctrl=M, dtype=dtype, method=eng.mn_method, step_ms=[round(w, 3) for w in walls],
               solve_ms=[round(s, 3) for s in solve], steady_step_ms=float(np.median(walls[4:])),
               steady_solve_ms=float(np.median(solve[4:])), cholesky=st["cholesky"], minnorm=st["minnorm"],
               rank=st["rank"][-3:], factor_rank=(st.get("factor_rank") or [None])[-3:], block=(st.get("block") or [None])[-3:],
               sweeps=st["sweeps"][-3:], sigma2=eng.sigma2, async_calls=st.get("async", 0))
    eng.k.drop_ublk()
    return rec, Vg


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_small_m_probe.json"))
    args = ap.parse_args()
    import torch

    from spateo_amd._synthetic import make_config

    torch.zeros(1, device="cuda:0")
    res = {}
    cases = [("C2_50k", make_config("C2")[:2], 500), ("C5_organ_250k", make_config("C2", N=250_000, seed=100)[:2], 500),
             ("C2_50k_M300", make_config("C2")[:2], 300), ("C2_50k_M200", make_config("C2")[:2], 200),
             ("n30k_M640", make_config("C2", N=30_000)[:2], 640)]
    for name, (X, V), M in cases:
        for dtype in ("float32", "float64"):
            out = {}
            fields = {}
            for method in ("full", "deflated"):
                rec, Vg = run(X, V, M, dtype, method, timing=(method == "deflated"))
                out[method] = rec
                fields[method] = Vg
            out["field_maxrel_between_methods"] = float(np.abs(fields["full"] - fields["deflated"]).max() /
                                                        np.abs(fields["full"]).max())
            res[f"{name}_{dtype}"] = out
            print(name, dtype, json.dumps(out), flush=True)
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
