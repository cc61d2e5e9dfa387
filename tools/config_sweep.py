#!/usr/bin/env python
"""Measure the BASELINE.json configurations that are NOT the bench line (C2, C3, C5) end to end on one MI355X.

    python tools/config_sweep.py [--out gpurun_out/config_sweep.json]

Reports, per configuration: host preprocessing time, upload + cache build, ms per EM iteration, whole `SparseVFC`
call wall time (host arrays in -> host dict out), and for C2 the Jacobian + curl evaluation on a 64^3 grid.
Developer tool (numbers quoted in DESIGN.md section 6); not part of the bench contract.
"""
from __future__ import annotations

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


def sync():
    import torch

    torch.cuda.synchronize()


def engine_steps(X, V, M, dtype, steps, lambda_=0.02):
    from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess

    t0 = time.perf_counter()
    valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=M, seed=0)
    t_pre = time.perf_counter() - t0
    t0 = time.perf_counter()
    eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype=dtype, device="cuda:0")
    eng.init_state(0.9)
    sync()
    t_up = time.perf_counter() - t0
    kw = dict(a=5.0, lambda_=lambda_, minP=1e-5, theta=0.75)
    for _ in range(2):
        eng.em_step(**kw)
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        eng.em_step(**kw)
    sync()
    ms = 1e3 * (time.perf_counter() - t0) / steps
    import torch

    # the coefficient solve alone (events on the stream), a few more steps
    ev = []
    inner = eng._solve_all

    def timed(ls2):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        h = inner(ls2)
        e1.record()
        ev.append((e0, e1))
        return h

    eng._solve_all = timed
    for _ in range(min(steps, 10)):
        eng.em_step(**kw)
    sync()
    solve_ms = float(np.mean([a.elapsed_time(b) for a, b in ev]))
    st = eng.solver_stats
    out = dict(cells=len(Xv), ctrl=len(ctrl), dtype=dtype, preprocess_s=t_pre, upload_and_cache_s=t_up, ms_per_em_step=ms,
               cells_per_s=len(Xv) / (ms * 1e-3), cached_u=bool(eng.cached_u), solve_ms=solve_ms,
               solver=dict(cholesky=st["cholesky"], minnorm=st["minnorm"], last_sweeps=st["sweeps"][-3:],
                           last_rank=st["rank"][-3:], method=getattr(eng, "mn_method", None)))
    eng.k.drop_ublk()
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "config_sweep.json"))
    ap.add_argument("--skip-c3", action="store_true")
    args = ap.parse_args()
    import torch

    from spateo_amd._synthetic import make_config
    from spateo_amd.vectorfield import SparseVFC, SparseVFC_many, SvcVectorField

    res = {}
    torch.zeros(1, device="cuda:0")

    # ------------------------------------------------------------------ C2: 50k cells, M = 500, Jacobian + curl on 64^3
    X, V, _ = make_config("C2")
    for dtype in ("float32", "float64"):
        r = engine_steps(X, V, 500, dtype, steps=50)
        SparseVFC(X, V, None, M=500, lambda_=0.02, MaxIter=3, dtype=dtype)  # warm
        t0 = time.perf_counter()
        vf = SparseVFC(X, V, None, M=500, lambda_=0.02, MaxIter=500, dtype=dtype)
        r["fit_wall_s"] = time.perf_counter() - t0
        r["fit_iterations"] = int(vf["iteration"])
        lo, hi = X.min(0), X.max(0)
        ax = [np.linspace(lo[c], hi[c], 64) for c in range(3)]
        grid = np.stack(np.meshgrid(*ax, indexing="ij"), axis=-1).reshape(-1, 3)
        svc = SvcVectorField(dtype=dtype)
        svc.vf_dict = vf
        svc.data = {"X": vf["X"], "V": vf["Y"]}
        svc.get_Jacobian()(grid[:1000])  # warm
        from spateo_amd.vectorfield import clear_eval_cache

        walls = []
        for _ in range(4):  # cold evaluator cache every time; the first call also pays the page-locked allocations
            clear_eval_cache()
            t0 = time.perf_counter()
            J = svc.get_Jacobian()(grid)
            curl = svc.compute_curl(grid)
            walls.append(1e3 * (time.perf_counter() - t0))
        r["jacobian_curl_64cube_first_call_wall_ms"] = walls[0]
        r["jacobian_curl_64cube_wall_ms"] = float(np.median(walls[1:]))
        r["jacobian_shape"] = list(np.shape(J))
        assert np.isfinite(J).all() and np.isfinite(curl).all()
        res[f"C2_{dtype}"] = r
        print(f"C2_{dtype}", json.dumps(r), flush=True)

    # ------------------------------------------------------------------ Spateo's stock call: M = 100 control points
    # (the default of morphofield_sparsevfc, reference sparsevfc.py:248), on the C2 cloud and on a 250 k-cell organ
    for name, (Xd, Vd) in (("default_M100_50k", (X, V)), ("default_M100_250k", make_config("C2", N=250_000, seed=100)[:2])):
        for dtype in ("float32", "float64"):
            r = engine_steps(Xd, Vd, 100, dtype, steps=50)
            res[f"{name}_{dtype}"] = r
            print(f"{name}_{dtype}", json.dumps(r), flush=True)

    # ------------------------------------------------------------------ C3: 2M cells, M = 2000
    if not args.skip_c3:
        X, V, _ = make_config("C3")
        for dtype in ("float32", "float64"):
            r = engine_steps(X, V, 2000, dtype, steps=5)
            res[f"C3_{dtype}"] = r
            print(f"C3_{dtype}", json.dumps(r), flush=True)
        del X, V

    # ------------------------------------------------------------------ C5: organs of 250k cells, M = 500 (4 per GPU)
    organs = []
    for s in range(4):
        Xo, Vo, _ = make_config("C2", N=250_000, seed=100 + s)
        organs.append((Xo, Vo, None))
    kw = dict(M=500, lambda_=0.02, MaxIter=30, dtype="float32")
    SparseVFC_many(organs[:1], n_streams=1, **dict(kw, MaxIter=2))  # warm
    t0 = time.perf_counter()
    seq = [SparseVFC(*o, **kw) for o in organs]
    t_seq = time.perf_counter() - t0
    t_pars = []
    for _ in range(3):  # the first parallel call also pays every thread's first-use costs (kernel objects, workspaces, streams)
        t0 = time.perf_counter()
        par = SparseVFC_many(organs, n_streams=4, **kw)
        t_pars.append(time.perf_counter() - t0)
    t_par = min(t_pars[1:])
    iters = [int(v["iteration"]) for v in par]
    same = all(np.allclose(a["C"], b["C"], rtol=1e-9, atol=1e-12) for a, b in zip(seq, par))
    r = engine_steps(organs[0][0], organs[0][1], 500, "float32", steps=30)
    res["C5_4organs_250k_M500_float32"] = dict(sequential_wall_s=t_seq, four_streams_wall_s=t_par,
                                               four_streams_first_call_wall_s=t_pars[0], iterations=iters,
                                               identical_to_sequential=bool(same), single_organ=r)
    print("C5", json.dumps(res["C5_4organs_250k_M500_float32"]), flush=True)

    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
