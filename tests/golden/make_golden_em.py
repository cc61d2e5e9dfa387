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


def guided_svi_case(mc, backend, rng, n, nb, m, n_i, beta, sigma2, lambdaVF, seed, guidance, svi, step_size=0.6,
                    guidance_weight=2.0):
    """The guidance ("nonrigid") and SVI branches of the real _update_nonrigid (morpho_class.py:1269-1284,1293-1294):
    U_I = con_K(X_AI, inducing variables), SigmaInv += w U_I^T U_I, rhs += w U_I^T (X_BI - R_AI), V_AI = U_I Coff with
    w = sigma2 guidance_weight Sp / n_I; SVI: SigmaInv / PXB_term blended with the previous batch's by step_size."""
    coordsA = rng.standard_normal((n, 3))
    coordsB = coordsA[rng.choice(n, nb)] + 0.05 * rng.standard_normal((nb, 3))
    d2 = ((coordsA[:, None, :] - coordsB[None, :, :]) ** 2).sum(-1)
    P = np.exp(-d2 / (2 * 0.3**2)) * (rng.random((n, nb)) < 0.5)
    P[rng.choice(n, n // 15, replace=False)] = 0.0
    K_NA = P.sum(1)
    RnA = coordsA + 0.02 * rng.standard_normal((n, 3))
    X_AI = coordsA[rng.choice(n, n_i, replace=False)] + 0.01 * rng.standard_normal((n_i, 3))
    X_BI = X_AI + 0.1 * rng.standard_normal((n_i, 3))
    R_AI = X_AI + 0.02 * rng.standard_normal((n_i, 3))
    s = types.SimpleNamespace(
        nx=backend.NumpyBackend(), coordsA=coordsA, coordsB=coordsB, kernel_type="euc", kernel_bandwidth=beta,
        guidance_effect="nonrigid" if guidance else False, guidance=bool(guidance), X_AI=X_AI, X_BI=X_BI, R_AI=R_AI,
        guidance_weight=guidance_weight, Sp=float(P.sum()), SVI_mode=bool(svi), step_size=step_size, sigma2=sigma2,
        lambdaVF=lambdaVF, P=P, K_NA=K_NA, RnA=RnA, graph=None, batch_idx=np.arange(nb),
    )
    np.random.seed(seed)
    mc.Morpho_pairwise._construct_kernel(s, m, None)
    prev = {}
    if svi:  # the previous batch's running averages (any symmetric PSD matrix of the right shape / any n x 3 term)
        Z = rng.standard_normal((2 * m, len(s.inducing_variables)))
        prev["SigmaInv_prev"] = Z.T @ Z / (2 * m) * float(K_NA.sum()) / m
        prev["PXB_prev"] = (P @ coordsB - RnA * K_NA[:, None]) * rng.uniform(0.5, 1.5, (n, 1))
        s.SigmaInv, s.PXB_term = prev["SigmaInv_prev"].copy(), prev["PXB_prev"].copy()
    mc.Morpho_pairwise._update_nonrigid(s)
    out = dict(coordsA=coordsA, K_NA=K_NA, beta=beta, sigma2=sigma2, lambdaVF=lambdaVF, Sp=s.Sp,
               inducing_variables=s.inducing_variables, SigmaInv=s.SigmaInv, PXB_term=s.PXB_term, Coff=s.Coff, VnA=s.VnA,
               SigmaDiag=s.SigmaDiag, PXB_new=P @ coordsB - RnA * K_NA[:, None], step_size=step_size,
               guidance_weight=guidance_weight, **prev)
    if guidance:
        out.update(X_AI=X_AI, X_BI=X_BI, R_AI=R_AI, V_AI=s.V_AI)
    return out


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
    # cases c / d / e: the guidance ("nonrigid") branch, the SVI branch, and both together (well conditioned, like a)
    for tag, kw in (("c", dict(guidance=True, svi=False)), ("d", dict(guidance=False, svi=True)),
                    ("e", dict(guidance=True, svi=True))):
        res = guided_svi_case(mc, backend, rng, n=600, nb=250, m=40, n_i=30, beta=0.5, sigma2=0.5, lambdaVF=100.0,
                              seed=13, **kw)
        for k, v in res.items():
            out[f"{tag}_{k}"] = np.asarray(v)
        print(f"case {tag}: M = {len(res['inducing_variables'])}, |Coff|max = {np.abs(res['Coff']).max():.3g}")
    path = os.path.join(HERE, "ref_em.npz")
    np.savez_compressed(path, **out)
    print("wrote", path, f"({os.path.getsize(path) / 1e6:.2f} MB, {len(out)} arrays)")


if __name__ == "__main__":
    main()
