#!/usr/bin/env python
"""Golden vectors for the control-point draw of SparseVFC: the REAL
``sample_by_velocity`` of ``spateo/alignment/methods/sampling.py:225-241`` - Spateo's in-tree copy of dynamo's
``dynamo/tools/sampling.py`` (same header, same ``LoggerManager`` import, same function set), the module dynamo's SparseVFC
draws its control points with.  The reference file is executed as it lies under /root/reference (only its three
package-relative imports - logger, kNN helpers it does not use here - are stubbed); nothing of it is copied.

Pinned: the |V|-weighted draw without replacement, the re-seeding of NumPy's GLOBAL generator with the function's own
default seed 19491001 (which is why a caller's ``np.random.seed(seed)`` just before it has no effect), and the state the
global generator is left in.

    python tests/golden/make_golden_sampling.py      -> tests/golden/ref_sampling.npz
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def load_reference_sampling():
    mg.install_stubs()
    for name in ("spateo.alignment", "spateo.alignment.methods"):
        if name not in sys.modules:
            mg._pkg(name)
    lg = types.ModuleType("spateo.alignment.dynamo_logger")
    lg.LoggerManager = type("LoggerManager", (), {})
    sys.modules["spateo.alignment.dynamo_logger"] = lg
    cn = types.ModuleType("spateo.alignment.methods.connectivity")
    cn.k_nearest_neighbors = None
    sys.modules["spateo.alignment.methods.connectivity"] = cn
    ut = types.ModuleType("spateo.alignment.methods.utils")
    ut.nearest_neighbors = None
    ut.timeit = lambda f: f
    sys.modules["spateo.alignment.methods.utils"] = ut
    return mg._load("spateo.alignment.methods.sampling", "spateo/alignment/methods/sampling.py")


def cases():
    rng = np.random.default_rng(20260926)
    V1 = rng.standard_normal((1000, 3))
    V2 = rng.standard_normal((400, 2)) * np.array([5.0, 0.1])
    V3 = rng.standard_normal((300, 3))
    V3[::7] = 0.0                      # zero-velocity rows can never be drawn
    V4 = np.abs(rng.standard_normal((64, 3))) + 0.1
    V5 = rng.standard_normal((5000, 3)) * rng.random((5000, 1)) ** 4   # heavy-tailed weights
    return {"a": (V1, 100, None), "b": (V2, 37, None), "c": (V3, 150, None), "d": (V4, 64, None),   # d: every row drawn
            "e": (V1, 100, 0), "f": (V1, 100, 12345), "g": (V5, 3000, None)}


def main():
    sm = load_reference_sampling()
    out = {}
    for tag, (V, n, seed) in cases().items():
        np.random.seed(4242)           # a caller's seed just before the call: must not matter for seed=None cases
        idx = sm.sample_by_velocity(V, n) if seed is None else sm.sample_by_velocity(V, n, seed=seed)
        nxt = np.random.random(3)      # the state the global generator is left in
        out[f"{tag}_V"], out[f"{tag}_n"], out[f"{tag}_seed"] = V, np.int64(n), np.int64(-1 if seed is None else seed)
        out[f"{tag}_idx"], out[f"{tag}_next"] = np.asarray(idx), nxt
    path = os.path.join(HERE, "ref_sampling.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: np.shape(v) for k, v in out.items() if k.endswith("_idx")})


if __name__ == "__main__":
    main()
