#!/usr/bin/env python
"""Developer probe (GPU): is the parity margin the deflated solve "spent" (the float64 field at M = 3000,
lambda_ = 0.02, 20 k cells moved from 1.02 x of the reference floor with the Jacobi solve to 1.19 x with the deflated one)
a property of the deflated solve, or one draw from the noise every mathematically equivalent solver lands in?

The same 10-step fits against the same committed oracle fixtures, with the truncated solve evaluated six ways that agree to
~1e-6 on any ONE system: Jacobi on the pivoted factor ("lowrank"), the deflated solve with its default block plan, with a
256-vector block alone (two and three applications of S2^-1), with a 128-vector block (three and four applications), and
the full-width Jacobi eigensolver.  Prints the field / sigma^2 / P / energy deviation of each as a multiple of the floor.

    python tools/solver_noise_probe.py [--out gpurun_out/r05_solver_noise.json]
"""
import argparse
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for _p in (ROOT, os.path.join(ROOT, "spateo-release_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

VARIANTS = [("lowrank (Jacobi on the factor)", "lowrank", {}),
            ("deflated, default plan", "deflated", {}),
            ("deflated, 256 x 2 applications", "deflated", {"defl_block": 256, "defl_apps": 2}),
            ("deflated, 256 x 3 applications", "deflated", {"defl_block": 256, "defl_apps": 3}),
            ("deflated, 128 x 3 applications", "deflated", {"defl_block": 128, "defl_apps": 3}),
            ("deflated, 128 x 4 applications", "deflated", {"defl_block": 128, "defl_apps": 4}),
            ("full-width Jacobi", "full", {})]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--out", default=os.path.join(ROOT, "gpurun_out", "r05_solver_noise.json"))
    args = ap.parse_args()
    import spateo_amd as st
    import test_gpu_scale as T
    from spateo_amd import _lib
    from spateo_amd.vectorfield import SparseVFCEngine

    res = {}
    for M in (3000, 2000):
        X, V, kw, ref, table = T._large_m_case(M, 0.02)
        for dtype in ("float64", "float32"):
            rows = {}
            for label, method, opts in VARIANTS:
                SparseVFCEngine.minnorm_method = method
                old = {k: _lib.debug_option(k, v) for k, v in opts.items()}
                try:
                    got = st.SparseVFC(X, V, None, dtype=dtype, device="cuda:0", **kw)
                finally:
                    SparseVFCEngine.minnorm_method = None
                    for k, v in old.items():
                        _lib.debug_option(k, v)
                dev = T._fixture_devs(got, ref, 1)
                fl = {k: table[k][0 if dtype == "float64" else 1] for k in dev}
                rows[label] = {k: {"gpu": dev[k], "floor": fl[k], "x": dev[k] / max(fl[k], 1e-300)} for k in dev}
                print(f"M={M} {dtype} {label}: " + ", ".join(f"{k} x{rows[label][k]['x']:.2f}" for k in dev), flush=True)
            xs = [r["V"]["x"] for r in rows.values()]
            res[f"M{M}_lam0.02_20k_{dtype}"] = {"variants": rows, "V_x_min": min(xs), "V_x_max": max(xs),
                                                "V_x_mean": float(np.mean(xs))}
    os.makedirs(os.path.dirname(args.out), exist_ok=True)
    with open(args.out, "w") as f:
        json.dump(res, f, indent=1)


if __name__ == "__main__":
    main()
