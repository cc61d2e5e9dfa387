"""TEST INFRASTRUCTURE - float64 NumPy restatement of the alignment-side callers of the Gaussian kernel
(SURVEY.md section 8f rank 4).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Parity status: PINNED - ``tests/golden/ref_align.npz`` / ``ref_em.npz`` hold outputs of the real reference functions
executed in the build container (``tests/golden/make_golden_align.py``, ``make_golden_em.py``); ``tests/test_oracle.py``
checks this file against them.
"""
from __future__ import annotations

import numpy as np


def con_K(X, Y, beta=0.01):
    """``spateo/alignment/methods/utils.py:1132-1158`` (con_K) over ``_euc_distance_backend`` (:747-787):
    D = ||x||^2 + ||y||^2 - 2 x.y, clamped at 0, K = exp(-beta D)."""
    X, Y = np.asarray(X, dtype=np.float64), np.asarray(Y, dtype=np.float64)
    assert X.shape[1] == Y.shape[1], "X and Y do not have the same number of features."
    D = np.sum(X**2, 1)[:, None] + np.sum(Y**2, 1)[None, :] - 2 * np.dot(X, Y.T)
    D = np.maximum(D, 0.0)
    return np.exp(-beta * D)


def BA_transform(vecfld, quary_points, deformation_scale=1):
    """``spateo/alignment/transform.py:61-116``: apply the learned non-rigid alignment to query points.
    Returns (XAHat, quary_velocities, quary_optimal_similarity)."""
    f = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731
    scale = f(vecfld["norm_dict"]["scale_transformed"])
    mean_ref = f(vecfld["norm_dict"]["mean_fixed"])
    mean_q = f(vecfld["norm_dict"]["mean_transformed"])
    XA = f(quary_points)
    if vecfld["normalize_c"]:
        XA = (XA - mean_q) / scale
    K = con_K(XA, f(vecfld["inducing_variables"]), vecfld["beta"])
    XA = XA @ f(vecfld["init_R"]).T + f(vecfld["init_t"])
    vel = (K @ f(vecfld["Coff"])) * deformation_scale
    sim = XA @ f(vecfld["R"]).T + f(vecfld["t"])
    opt = XA @ f(vecfld["optimal_R"]).T + f(vecfld["optimal_t"])
    hat = vel + sim
    if vecfld["normalize_c"]:
        hat = hat * scale + mean_ref
        vel = vel * scale
        opt = opt * scale + mean_ref
    return hat, vel, opt


def update_nonrigid(coordsA, coordsB, P, K_NA, RnA, inducing_variables, beta, sigma2, lambdaVF):
    """``Morpho_pairwise._construct_kernel`` kernels (``morpho_class.py:858-860``) + ``_update_nonrigid``
    (``:1254-1298``; no guidance, no SVI): returns dict(GammaSparse, U, SigmaInv, PXB_term, Coff, VnA, SigmaDiag).
    ``_pinv`` on the NumPy backend is ``scipy.linalg.pinv`` (``methods/utils.py:11,1435``)."""
    from scipy.linalg import pinv

    Gamma = con_K(inducing_variables, inducing_variables, beta)
    U = con_K(coordsA, inducing_variables, beta)
    SigmaInv = sigma2 * lambdaVF * Gamma + np.dot(U.T, np.einsum("ij,i->ij", U, K_NA))
    PXB = np.dot(P, coordsB) - np.einsum("ij,i->ij", RnA, K_NA)
    UPXB = np.dot(U.T, PXB)
    Sigma = pinv(SigmaInv)
    Coff = np.dot(Sigma, UPXB)
    VnA = np.dot(U, Coff)
    SigmaDiag = sigma2 * np.einsum("ij->i", np.einsum("ij,ji->ij", U, np.dot(Sigma, U.T)))
    return dict(GammaSparse=Gamma, U=U, SigmaInv=SigmaInv, PXB_term=PXB, Coff=Coff, VnA=VnA, SigmaDiag=SigmaDiag)
