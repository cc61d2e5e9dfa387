"""Alignment-side callers of the same Gaussian kernel and M-step: ``BA_transform`` and
``update_nonrigid`` (the non-rigid update of ``Morpho_pairwise``: the SparseVFC M-step with Gamma <-> K, K_NA <-> P).

Mirror of ``spateo/alignment/transform.py:61-116``: the learned non-rigid alignment ``vecfld`` (output of
``st.align.morpho_align``) applied to query points.  The N x M kernel contraction ``con_K(x, ctrl, beta) @ Coff`` runs
on the MI355X through ``libmvf`` (``mvf_apply``); the three 3 x 3 similarity maps around it are O(N) host NumPy.
``dtype`` selects the device cell dtype like the reference's ``dtype`` selects its backend dtype; outputs are host
float64 arrays ``(XAHat, quary_velocities, quary_optimal_similarity)``.
"""
from __future__ import annotations

import numpy as np

import torch

from . import _lib
from . import _runtime as _rt
from .engine import SparseVFCEngine, _consistent_K
from .vectorfield import vector_field_function

__all__ = ["BA_transform", "update_nonrigid"]


def update_nonrigid(coordsA, inducing_variables, beta, K_NA, PXB_term, sigma2, lambdaVF, dtype: str = "float64",
                    device=None, *, guidance=None, svi=None):
    """The non-rigid update of Spateo's alignment, ``Morpho_pairwise._update_nonrigid``
    (``spateo/alignment/methods/morpho_class.py:1254-1298``), on the MI355X with the kernels of the SparseVFC M-step - it
    is the same computation (``SigmaInv = sigma2 lambdaVF Gamma + U^T diag(K_NA) U``,
    ``Coff = pinv(SigmaInv) U^T PXB_term``, ``VnA = U Coff``; Gamma = con_K(ctrl, ctrl), U = con_K(coordsA, ctrl) as
    ``_construct_kernel`` builds them, ``:825-875``):

    * ``U^T diag(K_NA) U`` and ``U^T PXB_term``  -> ``mvf_gram`` (f64 MFMA; U is never materialised),
    * ``pinv(SigmaInv) @ rhs``                   -> ``mvf_solve_minnorm`` / ``_lr`` with scipy.linalg.pinv's default cut-off
      ``max(M, M) * eps * s_max`` (what ``_pinv`` resolves to on the NumPy backend, ``methods/utils.py:11,1435``),
    * ``U @ Coff``                               -> ``mvf_apply``.

    ``PXB_term`` (n x D) is the reference's ``P @ coordsB - RnA * K_NA[:, None]`` of THIS batch (O(n nb) host work that
    belongs to the assignment step, not to this one).

    ``svi`` (the ``SVI_mode`` branch, ``:1269-1275``): ``dict(step_size=, SigmaInv_prev=, PXB_prev=)`` - the running
    averages of the previous batches; ``SigmaInv`` and ``PXB_term`` are blended ``step new + (1 - step) prev`` before the
    solve (``mvf_lincomb3``) and returned for the next call.
    ``guidance`` (the ``guidance_effect in ("nonrigid", "both")`` branch, ``:1282-1288,1293-1294``):
    ``dict(X_AI=, X_BI=, R_AI=, weight=, Sp=)`` - ``U_I = con_K(X_AI, ctrl)``; ``SigmaInv += w U_I^T U_I``,
    ``rhs += w U_I^T (X_BI - R_AI)`` with ``w = sigma2 weight Sp / n_I`` (a second ``mvf_gram`` with unit weights) and
    ``V_AI = U_I Coff`` (``mvf_apply``) is returned too.

    Returns ``{"SigmaInv", "PXB_term", "Coff", "VnA", "SigmaDiag"[, "V_AI"]}`` as host float64; ``SigmaDiag`` =
    ``sigma2 diag(U pinv(SigmaInv) U^T)`` (the n-vector feeding the alignment's variational sigma^2, ``:1295-1297``) comes
    from the decomposition the solve left on the device (``mvf_pinv_diag``)."""
    if dtype not in ("float32", "float64"):
        raise ValueError("dtype must be 'float32' or 'float64'")
    X = np.asarray(coordsA, dtype=np.float64)
    ctrl = np.asarray(inducing_variables, dtype=np.float64)
    w = np.asarray(K_NA, dtype=np.float64).reshape(-1)
    B = np.asarray(PXB_term, dtype=np.float64)
    if X.ndim != 2 or ctrl.ndim != 2 or X.shape[1] != ctrl.shape[1]:
        raise AssertionError("X and Y do not have the same number of features.")  # con_K's assertion (utils.py:1150)
    if len(w) != len(X) or B.shape[0] != len(X) or B.ndim != 2 or not (1 <= B.shape[1] <= 3):
        raise ValueError("K_NA must be (n,) and PXB_term (n, D) with D <= 3")
    n, m, D = len(X), len(ctrl), B.shape[1]
    k = _rt._make_kernels(device, dtype)
    npdt = np.float32 if dtype == "float32" else np.float64
    center = ctrl.mean(0)
    x4, c4 = k.to_x4(X, center), k.to_x4(ctrl, center)
    Gamma = _consistent_K(k, ctrl, center, float(beta))  # generated like U (see SparseVFCEngine)
    f64 = torch.float64
    G, R = k.zeros(m, m, dtype=f64), k.zeros(m, 3, dtype=f64)
    Pw = k.h2d(w.astype(npdt))
    ls2 = float(sigma2) * float(lambdaVF)
    step = 1.0
    if svi is None:
        # rhs = U^T PXB = U^T diag(K_NA) Y with Y = PXB / K_NA (rows with K_NA == 0 have PXB == 0: cells without a partner)
        Y = np.divide(B, w[:, None], out=np.zeros_like(B), where=w[:, None] != 0)
        k.gram(x4, Pw, k.to_x4(Y), c4, float(beta), G, R)
    else:
        step = float(svi["step_size"])
        S_prev = np.ascontiguousarray(svi["SigmaInv_prev"], dtype=np.float64)
        B_prev = np.asarray(svi["PXB_prev"], dtype=np.float64)
        if S_prev.shape != (m, m) or B_prev.shape != B.shape or not (0.0 < step <= 1.0):
            raise ValueError("svi needs 0 < step_size <= 1, SigmaInv_prev (M, M) and PXB_prev shaped like PXB_term")
        B = step * B + (1.0 - step) * B_prev  # the blended n x D term (host, O(n)): rows without a partner NOW may carry
        # the previous batches' share, so its rhs is taken with unit weights
        k.gram(x4, Pw, k.to_x4(np.zeros_like(B)), c4, float(beta), G, R)
        ones = torch.ones(n, dtype=Pw.dtype, device=k.device)
        k.gram(x4, ones, k.to_x4(B), c4, float(beta), G, R, rhs_only=True)
        # G <- step G + (1 - step) (SigmaInv_prev);  the regulariser enters the solve as (step ls2) Gamma
        k.lincomb3(G, step, G, 1.0 - step, k.h2d(S_prev))
    x4_I = None
    if guidance is not None:
        X_AI = np.asarray(guidance["X_AI"], dtype=np.float64)
        dXI = np.asarray(guidance["X_BI"], dtype=np.float64) - np.asarray(guidance["R_AI"], dtype=np.float64)
        if X_AI.ndim != 2 or X_AI.shape[1] != X.shape[1] or dXI.shape != X_AI.shape:
            raise ValueError("guidance needs X_AI, X_BI, R_AI of one (n_I, D) shape")
        n_i = len(X_AI)
        wg = float(sigma2) * float(guidance["weight"]) * float(guidance["Sp"]) / n_i
        x4_I = k.to_x4(X_AI, center)
        G_I, R_I = k.zeros(m, m, dtype=f64), k.zeros(m, 3, dtype=f64)
        k.gram(x4_I, torch.ones(n_i, dtype=Pw.dtype, device=k.device), k.to_x4(dXI), c4, float(beta), G_I, R_I)
        k.lincomb3(G, 1.0, G, wg, G_I)
        k.lincomb3(R, 1.0, R, wg, R_I)
    C, info, einfo = k.zeros(m, 3, dtype=f64), k.zeros(1, dtype=torch.int32), k.zeros(12, dtype=f64)
    rcond = m * float(np.finfo(np.float64).eps)
    lowrank = m >= 1024 and hasattr(k, "solve_minnorm_lr")
    if lowrank:
        # rank-revealing factor + Jacobi on its columns (the faster path once the factor drops most columns)
        k.solve_minnorm_lr(G, Gamma, step * ls2, R, C, info, einfo, rcond=rcond)
        if int(info.cpu()[0]) != 0:
            raise _lib.MVFError("update_nonrigid: SigmaInv has non-finite entries")
        SparseVFCEngine._check_converged(float(einfo.cpu()[0]))
    else:
        shift = 2.0 ** -36
        while True:
            k.solve_minnorm(G, Gamma, step * ls2, shift, R, C, info, einfo, rcond=rcond)
            if int(info.cpu()[0]) == 0:
                SparseVFCEngine._check_converged(float(einfo.cpu()[0]))
                break
            shift *= 16.0
            if shift > 2.0 ** -12:
                raise _lib.MVFError("update_nonrigid: SigmaInv is not numerically positive semi-definite")
    V4, _ = k.apply(x4, c4, float(beta), C)
    SigmaInv = _rt._d2h(k, G) + step * ls2 * _rt._d2h(k, Gamma)
    # SigmaDiag = sigma2 diag(U pinv(SigmaInv) U^T) (morpho_class.py:1295-1297) from the decomposition the solve left
    diag = k.pinv_diag(x4, c4, float(beta), rcond=rcond, lowrank=lowrank) if hasattr(k, "pinv_diag") else None
    out = {"SigmaInv": SigmaInv, "PXB_term": B, "Coff": _rt._d2h(k, C)[:, :D].copy(),
           "VnA": _rt._d2h(k, V4[:, :D].to(f64))}
    if diag is not None:
        out["SigmaDiag"] = float(sigma2) * _rt._d2h(k, diag)
    if x4_I is not None:
        out["V_AI"] = _rt._d2h(k, k.apply(x4_I, c4, float(beta), C)[0][:, :D].to(f64))
    return out


def BA_transform(vecfld, quary_points, deformation_scale: int = 1, dtype: str = "float64", device=None):
    if dtype not in ("float32", "float64"):
        raise ValueError("dtype must be 'float32' or 'float64'")
    f = lambda a: np.asarray(a, dtype=np.float64)  # noqa: E731
    scale = f(vecfld["norm_dict"]["scale_transformed"])
    mean_ref = f(vecfld["norm_dict"]["mean_fixed"])
    mean_q = f(vecfld["norm_dict"]["mean_transformed"])
    XA = f(quary_points)
    if XA.ndim != 2:
        raise ValueError("quary_points must be (n, d)")
    if vecfld["normalize_c"]:
        XA = (XA - mean_q) / scale
    ctrl = f(vecfld["inducing_variables"])
    if XA.shape[1] != ctrl.shape[1]:  # the reference's con_K assertion (alignment/methods/utils.py:1150)
        raise AssertionError("X and Y do not have the same number of features.")
    field = {"X_ctrl": ctrl, "C": f(vecfld["Coff"]), "beta": float(vecfld["beta"])}
    vel = vector_field_function(XA, field, dtype=dtype, device=device) * deformation_scale
    XA = XA @ f(vecfld["init_R"]).T + f(vecfld["init_t"])
    sim = XA @ f(vecfld["R"]).T + f(vecfld["t"])
    opt = XA @ f(vecfld["optimal_R"]).T + f(vecfld["optimal_t"])
    hat = vel + sim
    if vecfld["normalize_c"]:
        hat = hat * scale + mean_ref
        vel = vel * scale
        opt = opt * scale + mean_ref
    return hat, vel, opt
