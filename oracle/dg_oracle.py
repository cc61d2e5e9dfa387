"""CPU ORACLE (test infrastructure, NOT product code) for the differential-geometry evaluators of the learned field.

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import this module.

Restates, in float64 NumPy, the formulas of the in-tree twin
``spateo/tdr/morphometrics/morphofield_dg/GPVectorField.py`` (which is a verbatim copy of dynamo's
``Jacobian_rkhs_gaussian`` / ``compute_*`` apart from the ``norm_dict`` scaling at ``:158-159,190``) for the
``method == "sparsevfc"`` route, where Spateo instantiates ``dynamo.vectorfield.scVectorField.SvcVectorField``
(``differential_geometry.py:24-28``).  Output shapes, including the reference's quirks, are preserved:

* Jacobian layout ``(D, D, n)`` with ``J[f, i, n] = d f_f / d x_i`` (``GPVectorField.py:143-190``).
* 3-D curl is returned as ``(n, 3, 3)``: the 3-vector is assigned into ``np.zeros((n, 3, 3))`` and so broadcast over
  the 3 rows (``GPVectorField.py:64-68``).
* torsion is the 3-vector ``(v a^T)(J a) / ||v a^T||_F^2`` broadcast into ``(n, 3, 3)`` (``GPVectorField.py:74-94``).

PARITY STATUS: every function here is pinned against the real reference twins executed in the build container with
``norm_dict`` set to the identity (``tests/golden/make_golden.py`` -> ``tests/golden/ref_twins.npz``).
"""
from __future__ import annotations

import numpy as np

from .sparsevfc_oracle import con_K, vector_field_function

__all__ = [
    "Jacobian_rkhs_gaussian",
    "compute_acceleration",
    "compute_curvature",
    "compute_curl",
    "compute_torsion",
    "compute_divergence",
    "SvcVectorField",
]


def Jacobian_rkhs_gaussian(x, vf_dict, vectorize: bool = False):
    """Analytical Jacobian ``J[i, j, n] = -2 beta sum_m K[n, m] C[m, i] (x_n - y_m)_j``.

    Twin: ``GPVectorField.py:143-190`` with ``pre_scale = 1``, ``x_norm = x``, ``Coff -> C``,
    ``inducing_variables -> X_ctrl``.  1-D input -> ``(D, D)``; 2-D input -> ``(D, D, n)``.
    ``vectorize`` selects the einsum path (``:179-188``) instead of the per-point loop (``:168-178``)."""
    x = np.asarray(x, dtype=float)
    Xc, beta, C = vf_dict["X_ctrl"], vf_dict["beta"], vf_dict["C"]
    if x.ndim == 1:
        K, D = con_K(x[None, :], Xc, beta, return_d=True)
        J = (C.T * K) @ D[0].T
    elif not vectorize:
        n, d = x.shape
        J = np.zeros((d, d, n))
        for i, xi in enumerate(x):
            K, D = con_K(xi[None, :], Xc, beta, return_d=True)
            J[:, :, i] = (C.T * K) @ D[0].T
    else:
        K, D = con_K(x, Xc, beta, return_d=True)
        if K.ndim == 1:
            K = K[None, :]
        J = np.einsum("nm,mi,njm->ijn", K, C, D)
    return -2 * beta * J


def compute_acceleration(vf, f_jac, X, Js=None, return_all=False):
    """``a_i = J_i v_i`` and ``||a_i||`` (``GPVectorField.py:12-32``)."""
    n = len(X)
    v_ = vf(X)
    J_ = f_jac(X) if Js is None else Js
    acce_mat = np.einsum("fin,ni->nf", J_, v_).reshape(n, X.shape[1])
    acce = np.linalg.norm(acce_mat, axis=1)
    if return_all:
        return v_, J_, acce, acce_mat
    return acce, acce_mat


def compute_curvature(vf, f_jac, X, Js=None, formula=2):
    """formula 1: ``||v a^T||_F / ||v||^3``; formula 2: ``kappa = (a (v.v) - v (v.a)) / ||v||^4``
    (``GPVectorField.py:35-52``)."""
    n = len(X)
    v, _, _, a = compute_acceleration(vf, f_jac, X, Js=Js, return_all=True)
    curv = np.zeros(n)
    cur_mat = np.zeros((n, X.shape[1])) if formula == 2 else None
    for i in range(n):
        ai, vi = a[i], v[i]
        if formula == 1:
            curv[i] = np.linalg.norm(np.outer(vi, ai)) / np.linalg.norm(vi) ** 3
        elif formula == 2:
            cur_mat[i] = (ai * np.dot(vi, vi) - vi * np.dot(vi, ai)) / np.linalg.norm(vi) ** 4
            curv[i] = np.linalg.norm(cur_mat[i])
    return curv, cur_mat


def compute_curl(f_jac, X):
    """2-D: ``J10 - J01`` -> (n,).  3-D: ``[J21-J12, J02-J20, J10-J01]`` broadcast into ``(n, 3, 3)``
    (``GPVectorField.py:55-71``)."""
    n = len(X)
    if X.shape[1] == 2:
        curl = np.zeros(n)
        for i in range(n):
            jac = f_jac(X[i])
            curl[i] = jac[1, 0] - jac[0, 1]
    elif X.shape[1] == 3:
        curl = np.zeros((n, 3, 3))
        for i in range(n):
            jac = f_jac(X[i])
            curl[i] = np.array([jac[2, 1] - jac[1, 2], jac[0, 2] - jac[2, 0], jac[1, 0] - jac[0, 1]])
    else:
        raise ValueError("X has incorrect dimensions.")
    return curl


def compute_torsion(vf, f_jac, X):
    """``tau = (v a^T)(J a) / ||v a^T||_F^2`` broadcast into ``(n, 3, 3)``; 3-D only (``GPVectorField.py:74-94``)."""
    if X.shape[1] != 3:
        raise Exception("torsion is only defined in 3 dimension.")
    n = len(X)
    tor = np.zeros((n, 3, 3))
    v, J, _, a = compute_acceleration(vf, f_jac, X, return_all=True)
    for i in range(n):
        va = np.outer(v[i], a[i])
        tor[i] = va.dot(J[:, :, i].dot(a[i])) / np.linalg.norm(va) ** 2
    return tor


def compute_divergence(f_jac, X, Js=None, vectorize_size=1000):
    """``div = trace(J)`` in batches of ``vectorize_size`` (``GPVectorField.py:97-121``)."""
    n = len(X)
    if vectorize_size is None:
        vectorize_size = n
    div = np.zeros(n)
    for i in range(0, n, vectorize_size):
        J = f_jac(X[i : i + vectorize_size]) if Js is None else Js[:, :, i : i + vectorize_size]
        div[i : i + vectorize_size] = np.trace(J)
    return div


class SvcVectorField:
    """Shape of ``dynamo.vectorfield.scVectorField.SvcVectorField`` as Spateo uses it
    (``differential_geometry.py:25-28,68,108-109,154-157,197-198,242-243,291-292,331-335``; in-tree twin
    ``GPVectorField.py:193-266``).  ``data["V"]`` is the raw input ``Y``."""

    def __init__(self):
        self.data = {}
        self.vf_dict = None
        self.func = None

    def from_adata(self, adata, basis=None, vf_key="VecFld"):
        if basis is not None and len(basis) > 0:
            vf_key = "%s_%s" % (vf_key, basis)
        if vf_key not in adata.uns.keys():
            raise ValueError(f"Vector field function {vf_key} is not included in the adata object!")
        vf_dict = adata.uns[vf_key]
        self.vf_dict = vf_dict
        self.func = lambda x: vector_field_function(x, vf_dict)
        self.data["X"] = vf_dict["X"]
        self.data["V"] = vf_dict["Y"]
        return self

    def get_data(self):
        return self.data["X"], self.data["V"]

    def get_Jacobian(self, method="analytical", **kwargs):
        if method != "analytical":
            raise NotImplementedError("method='numerical' (numdifftools) is out of scope")
        return lambda x: Jacobian_rkhs_gaussian(x, self.vf_dict)

    def compute_acceleration(self, X=None, method="analytical", **kwargs):
        X = self.data["X"] if X is None else X
        return compute_acceleration(self.func, self.get_Jacobian(method=method), X)

    def compute_curvature(self, X=None, method="analytical", formula=2, **kwargs):
        X = self.data["X"] if X is None else X
        return compute_curvature(self.func, self.get_Jacobian(method=method), X, formula=formula)

    def compute_curl(self, X=None, method="analytical", dim1=0, dim2=1, dim3=2, **kwargs):
        X = self.data["X"] if X is None else X
        if dim3 is None or X.shape[1] == 2:
            X = X[:, [dim1, dim2]]
        else:
            X = X[:, [dim1, dim2, dim3]]
        return compute_curl(self.get_Jacobian(method=method), X)

    def compute_torsion(self, X=None, method="analytical", **kwargs):
        X = self.data["X"] if X is None else X
        return compute_torsion(self.func, self.get_Jacobian(method=method), X)

    def compute_divergence(self, X=None, method="analytical", vectorize_size=1000, **kwargs):
        X = self.data["X"] if X is None else X
        return compute_divergence(self.get_Jacobian(method=method), X, vectorize_size=vectorize_size)
