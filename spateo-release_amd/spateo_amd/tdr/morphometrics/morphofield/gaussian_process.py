"""``morphofield_gp`` with the reference's signature (``spateo/tdr/morphometrics/morphofield/gaussian_process.py:173-233``):
evaluates the Gaussian-process field produced by ``st.align.morpho_align`` (``uns[vf_key]`` with ``norm_dict``,
``inducing_variables``, ``Coff``, ``R``, ``t``, ``beta``) at the cells and on a grid, on the GPU."""
from __future__ import annotations

from typing import List, Optional

import numpy as np

from ....logging import logger_manager as lm
from ....vectorfield import gp_velocity as _gp_velocity
from ...interpolations import get_X_Y_grid


def morphofield_gp(
    adata,
    spatial_key: str = "align_spatial",
    vf_key: str = "VecFld_morpho",
    NX: Optional[np.ndarray] = None,
    grid_num: Optional[List[int]] = None,
    nonrigid_only: bool = False,
    inplace: bool = True,
):
    """Fills ``uns[vf_key]`` with ``X``, ``V``, ``grid``, ``grid_V`` and ``method = "gaussian_process"``."""
    adata = adata if inplace else adata.copy()
    if vf_key in adata.uns.keys():
        vf_dict = adata.uns[vf_key]
        vf_dict["X"] = np.asarray(adata.obsm[spatial_key], dtype=float)
        vf_dict["V"] = _gp_velocity(vf_dict["X"], vf_dict=vf_dict, nonrigid_only=nonrigid_only)
        if NX is not None:
            predict_X = NX
        else:
            if grid_num is None:
                grid_num = [50, 50, 50]
                lm.main_warning("grid_num and NX are both None, using `grid_num = [50,50,50]`.", indent_level=1)
            _, _, predict_X, _ = get_X_Y_grid(X=vf_dict["X"].copy(), Y=vf_dict["V"].copy(), grid_num=grid_num)
        vf_dict["grid"] = predict_X
        vf_dict["grid_V"] = _gp_velocity(predict_X, vf_dict=vf_dict, nonrigid_only=nonrigid_only)
        vf_dict["method"] = "gaussian_process"
        lm.main_finish_progress(progress_name="morphofield")
    else:
        raise Exception(
            f"The {vf_key} that corresponds to the reconstructed vector field is not in ``anndata.uns``."
            f"Please run ``st.align.morpho_align(adata, vecfld_key_added='{vf_key}')`` before running this function."
        )
    return None if inplace else adata
