"""``kernel_interpolation`` with the reference's signature (``spateo/tdr/interpolations/interpolation_sparseVFC.py:13-85``):
the second call site of the SparseVFC engine, with ``Y`` = the selected obs / gene columns (Dy = #keys) instead of a
displacement field.  The fit runs on the GPU; wide ``Y`` is processed in 3-column groups sharing one Gram matrix."""
from __future__ import annotations

from typing import Optional, Union

import numpy as np

from ...logging import logger_manager as lm
from ...vectorfield import SparseVFC


def _dense(a):
    return a.toarray() if hasattr(a, "toarray") else np.asarray(a)


def kernel_interpolation(
    source_adata,
    target_points: Optional[np.ndarray] = None,
    keys: Union[str, list] = None,
    spatial_key: str = "spatial",
    layer: str = "X",
    lambda_: float = 0.02,
    lstsq_method: str = "scipy",
    **kwargs,
):
    """Learn a continuous mapping from space to the ``keys`` (obs columns first, then genes) with SparseVFC and
    evaluate it at ``target_points``.  Returns an AnnData (``anndata.AnnData`` when installed, else ``AnnDataLite``)
    with ``obs[obs_keys]``, ``X`` / ``var_names`` = the interpolated genes and ``obsm[spatial_key] = target_points``."""
    assert keys is not None, "`keys` cannot be None."
    keys = [keys] if isinstance(keys, str) else list(keys)
    Xmat = source_adata.X if layer == "X" else source_adata.layers[layer]
    spatial = np.asarray(source_adata.obsm[spatial_key], dtype=float)
    var_names = [str(v) for v in list(source_adata.var_names)]
    obs_keys = [k for k in keys if k in source_adata.obs.keys()]
    var_keys = [k for k in keys if k in var_names]
    cols = []
    if obs_keys:
        cols.append(np.column_stack([np.asarray(source_adata.obs[k], dtype=float) for k in obs_keys]))
    if var_keys:
        idx = [var_names.index(k) for k in var_keys]
        cols.append(_dense(Xmat[:, idx]).astype(float))
    if not cols:
        raise ValueError(f"none of the keys {keys} is an obs column or a gene of source_adata")
    info_data = np.concatenate(cols, axis=1)

    res = SparseVFC(spatial, info_data, target_points, lambda_=lambda_, lstsq_method=lstsq_method, **kwargs)
    target = res["grid_V"]
    lm.main_info("Creating an adata object with the interpolated expression...")
    obs_part = target[:, : len(obs_keys)]
    x_part = target[:, len(obs_keys) :] if var_keys else None
    try:
        import pandas as pd
        from anndata import AnnData

        out = AnnData(
            X=x_part,
            obs=pd.DataFrame(obs_part, columns=obs_keys) if obs_keys else None,
            obsm={spatial_key: np.asarray(target_points)},
            var=pd.DataFrame(index=var_keys) if var_keys else None,
        )
    except ImportError:
        from ..._anndata_lite import AnnDataLite

        out = AnnDataLite(X=x_part, var_names=var_keys, obs={k: obs_part[:, i] for i, k in enumerate(obs_keys)},
                          obsm={spatial_key: np.asarray(target_points)}, n_obs=len(target))
    lm.main_finish_progress(progress_name="KernelInterpolation")
    return out
