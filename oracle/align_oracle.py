"""TEST INFRASTRUCTURE - float64 NumPy restatement of the alignment-side callers of the Gaussian kernel
(SURVEY.md section 8f rank 4).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.

Parity status: PINNED - ``tests/golden/ref_align.npz`` holds outputs of the real reference functions executed in the
build container (``tests/golden/make_golden_align.py``); ``tests/test_oracle.py`` checks this file against them.
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
