#!/usr/bin/env python
"""Golden vectors for ``cell_directions``: the REAL function of
``spateo/tdr/morphometrics/morphofield/sparsevfc.py:18-100`` is executed with the REAL
``get_optimal_mapping_relationship`` (``spateo/alignment/utils.py:157-193``); only the PASTE optimal-transport solve it
calls first (``paste_pairwise_align``: POT's FGW solver, outside this repo's tier) is replaced by a function that returns
a prepared coupling matrix ``pi`` - the mapping logic after it (row maxima of pi, ties broken by the nearest coordinate
unless ``keep_all``, one partner per cell, ``X_mapping``, ``V_mapping = X_mapping - X``) is what is pinned.

    python tests/golden/make_golden_celldir.py      -> tests/golden/ref_celldir.npz
"""
from __future__ import annotations

import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402


def main():
    mg.install_stubs()
    au = mg._load("spateo.alignment.utils", "spateo/alignment/utils.py")
    sys.modules["spateo.alignment"].get_optimal_mapping_relationship = au.get_optimal_mapping_relationship
    box = {}
    sys.modules["spateo.alignment.methods"].paste_pairwise_align = lambda **kw: (box["pi"], None)
    interp_pkg = mg._pkg("spateo.tdr.interpolations")
    iu = mg._load("spateo.tdr.interpolations.utils", "spateo/tdr/interpolations/utils.py")
    interp_pkg.get_X_Y_grid = iu.get_X_Y_grid
    sv = mg._load("spateo.tdr.morphometrics.morphofield.sparsevfc", "spateo/tdr/morphometrics/morphofield/sparsevfc.py")

    rng = np.random.default_rng(20260928)
    out = {}
    na, nb = 60, 45
    XA = rng.standard_normal((na, 3)) * 10
    XB = XA[rng.choice(na, nb)] + rng.standard_normal((nb, 3))
    pi = rng.random((na, nb)) ** 6
    # exact ties of the row maximum (several partners with the same coupling): rows 3, 10, 11, 40
    for i, js in ((3, (2, 7, 30)), (10, (0, 44)), (11, (5, 6, 7, 8)), (40, (12, 13))):
        pi[i, list(js)] = pi[i].max() * 1.5
    pi /= pi.sum()
    out["XA"], out["XB"], out["pi"] = XA, XB, pi
    for keep_all in (False, True):
        box["pi"] = pi
        A = mg.AnnDataLite(obsm={"align_spatial": XA.copy()})
        B = mg.AnnDataLite(obsm={"align_spatial": XB.copy()})
        ret, pi_out = sv.cell_directions(A, B, keep_all=keep_all, inplace=True)
        assert ret is None and pi_out is pi
        tag = "all" if keep_all else "nearest"
        out[f"{tag}_X_mapping"] = np.asarray(A.obsm["X_mapping"])
        out[f"{tag}_V_mapping"] = np.asarray(A.obsm["V_mapping"])
    path = os.path.join(HERE, "ref_celldir.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
