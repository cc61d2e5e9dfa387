"""Alignment-side caller of the same Gaussian kernel: ``BA_transform``.

Mirror of ``spateo/alignment/transform.py:61-116``: the learned non-rigid alignment ``vecfld`` (output of
``st.align.morpho_align``) applied to query points.  The N x M kernel contraction ``con_K(x, ctrl, beta) @ Coff`` runs
on the MI355X through ``libmvf`` (``mvf_apply``); the three 3 x 3 similarity maps around it are O(N) host NumPy.
``dtype`` selects the device cell dtype like the reference's ``dtype`` selects its backend dtype; outputs are host
float64 arrays ``(XAHat, quary_velocities, quary_optimal_similarity)``.
"""
from __future__ import annotations

import numpy as np

from .vectorfield import vector_field_function

__all__ = ["BA_transform"]


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
