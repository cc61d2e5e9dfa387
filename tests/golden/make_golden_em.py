#!/usr/bin/env python
"""Golden vectors that pin the M-STEP ARITHMETIC of the SparseVFC path to real reference code.

dynamo (where ``SparseVFC``'s EM loop lives) is not in /root/reference and cannot be installed, but Spateo carries the
same M-step in-tree, in its alignment module:

    spateo/alignment/methods/morpho_class.py:825-875   Morpho_pairwise._construct_kernel
        inducing variables (control points), GammaSparse = con_K(ctrl, ctrl), U = con_K(coordsA, ctrl)
    spateo/alignment/methods/morpho_class.py:1254-1298 Morpho_pairwise._update_nonrigid
        SigmaInv = sigma2 * lambdaVF * GammaSparse + U^T diag(K_NA) U
        PXB_term = P @ coordsB - RnA * K_NA ;  Coff = pinv(SigmaInv) @ (U^T PXB_term) ;  VnA = U @ Coff
        SigmaDiag = sigma2 * diag(U pinv(SigmaInv) U^T)

i.e. lhs = lambda sigma^2 K + U^T P U, rhs = U^T P Y, C = lstsq(lhs, rhs), V = U C  with  Gamma <-> K, K_NA <-> P,
PXB_term <-> P * Y.  This script EXECUTES those two methods (unbound, on a ``SimpleNamespace`` self, NumPy backend, no
guidance, no SVI) from the real file and stores their inputs and outputs in ``tests/golden/ref_em.npz``.  The oracle's
M-step and the HIP path (mvf_gram + mvf_solve / mvf_solve_minnorm + mvf_apply, through ``spateo_amd.align.
update_nonrigid``) are checked against them.  What stays unpinnable (dynamo-only lines): ``get_P``, ``sample_by_velocity``,
``bandwidth_selector``, the sigma^2 / gamma update and the stopping rule.

    python tests/golden/make_golden_em.py
"""
from __future__ import annotations

import os
import sys
import types

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import make_golden as mg  # noqa: E402
import make_golden_align as mga  # noqa: E402


def load_morpho_class():
    utils, _ = mga.load_alignment()
    mc = mg._load("spateo.alignment.methods.morpho_class", "spateo/alignment/methods/morpho_class.py")
    backend = sys.modules["spateo.alignment.methods.backend"]
    return mc, backend, utils


def one_case(mc, backend, rng, n, nb, m, beta, sigma2, lambdaVF, seed, spread):
    """Run the real _construct_kernel + _update_nonrigid on synthetic alignment state."""
    coordsA = rng.standard_normal((n, 3)) * spread
    coordsB = coordsA[rng.choice(n, nb)] + 0.05 * rng.standard_normal((nb, 3))
    # a soft assignment matrix P (n x nb) with some all-zero rows (cells without a partner), as in the alignment
    d2 = ((coordsA[:, None, :] - coordsB[None, :, :]) ** 2).sum(-1)
    P = np.exp(-d2 / (2 * 0.3**2)) * (rng.random((n, nb)) < 0.5)
    P[rng.choice(n, n // 15, replace=False)] = 0.0
    K_NA = P.sum(1)
    RnA = coordsA + 0.02 * rng.standard_normal((n, 3))  # "rigidly transformed" coordinates
    s = types.SimpleNamespace(
        nx=backend.NumpyBackend(), coordsA=coordsA, coordsB=coordsB, kernel_type="euc", kernel_bandwidth=beta,
        guidance_effect=False, guidance=False, X_AI=None, SVI_mode=False, sigma2=sigma2, lambdaVF=lambdaVF, P=P,
        K_NA=K_NA, RnA=RnA, graph=None,
    )
    np.random.seed(seed)  # _construct_kernel draws the inducing variables from NumPy's global RNG
    mc.Morpho_pairwise._construct_kernel(s, m, None)
    mc.Morpho_pairwise._update_nonrigid(s)
    return dict(coordsA=coordsA, coordsB=coordsB, P=P, K_NA=K_NA, RnA=RnA, beta=beta, sigma2=sigma2, lambdaVF=lambdaVF,
                inducing_variables=s.inducing_variables, GammaSparse=s.GammaSparse, U=s.U, SigmaInv=s.SigmaInv,
                PXB_term=s.PXB_term, Coff=s.Coff, VnA=s.VnA, SigmaDiag=s.SigmaDiag)


def main():
    mc, backend, utils = load_morpho_class()
    rng = np.random.default_rng(20260927)
    out = {}
    # case a: well conditioned (strong regulariser, few well-separated inducing points): C itself is determined
    # case b: Spateo-like (wide kernel, many inducing points, weak regulariser): numerically rank deficient, pinv truncates
    # (99 of 120 directions kept at scipy.linalg.pinv's M eps cut-off; a 1e-13 relative perturbation of SigmaInv moves
    # VnA by 6e-4: that is the noise floor any comparison on this case can reach)
    for tag, kw in (("a", dict(n=700, nb=300, m=40, beta=0.5, sigma2=0.5, lambdaVF=100.0, seed=11, spread=1.0)),
                    ("b", dict(n=900, nb=400, m=120, beta=0.1, sigma2=1e-1, lambdaVF=1.0, seed=12, spread=1.0))):
        res = one_case(mc, backend, rng, **kw)
        for k, v in res.items():
            if tag == "b" and k in ("U", "P", "coordsB", "RnA", "GammaSparse"):
                continue  # case b keeps what the solve needs (K_NA, PXB_term) - the composition is pinned by case a
            out[f"{tag}_{k}"] = np.asarray(v)
        w = np.linalg.eigvalsh((res["SigmaInv"] + res["SigmaInv"].T) / 2)
        print(f"case {tag}: M = {len(res['inducing_variables'])}, cond(SigmaInv) = {w.max() / max(w.min(), 1e-300):.2e}, "
              f"|Coff|max = {np.abs(res['Coff']).max():.3g}, |VnA|max = {np.abs(res['VnA']).max():.3g}")
    path = os.path.join(HERE, "ref_em.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, f"({os.path.getsize(path) / 1e6:.2f} MB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
