"""The seven ``morphofield_*`` differential-geometry wrappers with the reference's signatures and AnnData slots
(``spateo/tdr/morphometrics/morphofield_dg/differential_geometry.py:42-341``), evaluated by the fused HIP kernel
behind :class:`spateo_amd.vectorfield.SvcVectorField`."""
from __future__ import annotations

from typing import Optional

import numpy as np

from ....vectorfield import GPVectorField, SvcVectorField


def _generate_vf_class(adata, vf_key: str, method: str = "gaussian_process", nonrigid_only: bool = False):
    """Dispatch on ``adata.uns[vf_key]["method"]`` (``differential_geometry.py:12-39``)."""
    if vf_key in adata.uns.keys():
        if method == "sparsevfc":
            vf = SvcVectorField()
            vf.from_adata(adata, basis=None, vf_key=vf_key)
            return vf
        elif method == "gaussian_process":
            vf = GPVectorField()
            vf.from_adata(adata, vf_key=vf_key, nonrigid_only=nonrigid_only)
            return vf
        raise Exception(
            f"The {method} is not in ``anndata.uns[{vf_key}]``."
            f"Please re-run ``st.tdr.morphofield_gp`` or ``st.tdr.morphofield_sparsevfc`` before running this function."
        )
    raise Exception(
        f"The {vf_key} that corresponds to the reconstructed vector field is not in ``anndata.uns``."
        f"Please run ``st.align.morpho_align(adata, vecfld_key_added='{vf_key}')`` before running this function."
    )


def _row_norms(a):
    """``np.array([np.linalg.norm(i) for i in a])`` (``differential_geometry.py:199,244``: 2-norm of a scalar / vector,
    Frobenius norm of a matrix entry) without the Python loop over the cells."""
    a = np.asarray(a)
    return np.abs(a) if a.ndim == 1 else np.sqrt(np.einsum("ij,ij->i", a.reshape(len(a), -1), a.reshape(len(a), -1)))


def _vf(adata, vf_key, nonrigid_only):
    return _generate_vf_class(adata=adata, vf_key=vf_key, method=adata.uns[vf_key]["method"], nonrigid_only=nonrigid_only)


def morphofield_velocity(adata, vf_key: str = "VecFld_morpho", key_added: str = "velocity",
                         nonrigid_only: bool = False, inplace: bool = True):
    """``obsm[key_added] = vf.func(uns[vf_key]["X"])`` (``:42-70``)."""
    adata = adata if inplace else adata.copy()
    vf = _vf(adata, vf_key, nonrigid_only)
    adata.obsm[key_added] = vf.func(adata.uns[vf_key]["X"])
    return None if inplace else adata


def morphofield_acceleration(adata, vf_key: str = "VecFld_morpho", key_added: str = "acceleration",
                             method: str = "analytical", nonrigid_only: bool = False, inplace: bool = True):
    """``obs[key] = ||J v||``, ``obsm[key] = J v`` (``:73-111``)."""
    adata = adata if inplace else adata.copy()
    vf = _vf(adata, vf_key, nonrigid_only)
    X, _ = vf.get_data()
    adata.obs[key_added], adata.obsm[key_added] = vf.compute_acceleration(X=X, method=method)
    return None if inplace else adata


def morphofield_curvature(adata, vf_key: str = "VecFld_morpho", key_added: str = "curvature", formula: int = 2,
                          method: str = "analytical", nonrigid_only: bool = False, inplace: bool = True):
    """``obs[key]`` = curvature, ``obsm[key]`` = curvature vectors (``:114-159``)."""
    adata = adata if inplace else adata.copy()
    vf = _vf(adata, vf_key, nonrigid_only)
    X, _ = vf.get_data()
    adata.obs[key_added], adata.obsm[key_added] = vf.compute_curvature(X=X, formula=formula, method=method)
    return None if inplace else adata


def morphofield_curl(adata, vf_key: str = "VecFld_morpho", key_added: str = "curl", method: str = "analytical",
                     nonrigid_only: bool = False, inplace: bool = True):
    """``obsm[key]`` = curl ((n, 3, 3) in 3-D, reference quirk); ``obs[key]`` = per-cell norm of that entry, i.e. the
    Frobenius norm of the 3 x 3 = sqrt(3) ||curl|| (``:162-204``)."""
    adata = adata if inplace else adata.copy()
    vf = _vf(adata, vf_key, nonrigid_only)
    X, _ = vf.get_data()
    curl = vf.compute_curl(X=X, method=method)
    adata.obs[key_added] = _row_norms(curl)
    adata.obsm[key_added] = curl
    return None if inplace else adata


def morphofield_torsion(adata, vf_key: str = "VecFld_morpho", key_added: str = "torsion", method: str = "analytical",
                        nonrigid_only: bool = False, inplace: bool = True):
    """``uns[key]`` = torsion (n, 3, 3); ``obs[key]`` = per-cell Frobenius norm (``:207-249``)."""
    adata = adata if inplace else adata.copy()
    vf = _vf(adata, vf_key, nonrigid_only)
    X, _ = vf.get_data()
    torsion_mat = vf.compute_torsion(X=X, method=method)
    adata.obs[key_added] = _row_norms(torsion_mat)
    adata.uns[key_added] = torsion_mat
    return None if inplace else adata


def morphofield_divergence(adata, vf_key: str = "VecFld_morpho", key_added: str = "divergence",
                           method: str = "analytical", vectorize_size: Optional[int] = 1000,
                           nonrigid_only: bool = False, inplace: bool = True):
    """``obs[key] = trace J`` (``:252-294``).  ``vectorize_size`` is accepted for compatibility; the kernel batches
    internally."""
    adata = adata if inplace else adata.copy()
    vf = _vf(adata, vf_key, nonrigid_only)
    X, _ = vf.get_data()
    adata.obs[key_added] = vf.compute_divergence(X=X, method=method, vectorize_size=vectorize_size)
    return None if inplace else adata


def morphofield_jacobian(adata, vf_key: str = "VecFld_morpho", key_added: str = "jacobian", method: str = "analytical",
                         nonrigid_only: bool = False, inplace: bool = True):
    """``uns[key]`` = Jacobians (d, d, n); ``obs[key]`` = det per cell (``:297-341``)."""
    adata = adata if inplace else adata.copy()
    vf = _vf(adata, vf_key, nonrigid_only)
    X, _ = vf.get_data()
    cell_idx = np.arange(adata.n_obs)
    Js, det = vf.jacobian_with_det(X[cell_idx], method=method)  # one device pass; no per-cell np.linalg.det
    adata.obs[key_added] = det
    adata.uns[key_added] = Js
    return None if inplace else adata
