"""``morphopath`` with the reference's signature (``spateo/tdr/morphometrics/morphofield/trajectory.py:11-117``):
predicts cell trajectories by integrating the learned morphometric vector field.

The reference hands the field to dynamo's ``fate`` (adaptive RK45 + arc-length resampling; third-party, not in the
reference tree).  Here the whole integration runs in ONE fused HIP kernel (``mvf_integrate``: classical RK4, one lane per
trajectory) and the trajectories are sampled at ``interpolation_num`` UNIFORM time points - documented deviation,
parity unpinned.  Output slots are the reference's: ``uns[key_added]["t"][i]`` (times) and
``uns[key_added]["prediction"][i]`` ((n_t, d) states: the reference transposes dynamo ``fate``'s (d, n_t) arrays,
``trajectory.py:113``, and its consumer concatenates ``init_states[[i]]`` with it along axis 0,
``tdr/models/models_migration/morphopath_model.py:225``) per cell; ``init_cells`` = the cells' ``obs_names``."""
from __future__ import annotations

from typing import Optional, Union

import numpy as np

from ....vectorfield import integrate_field


def _obs_names(adata, n):
    names = getattr(adata, "obs_names", None)
    return [str(x) for x in names] if names is not None and len(names) == n else [str(i) for i in range(n)]


def morphopath(
    adata,
    vf_key: str = "VecFld_morpho",
    key_added: str = "fate_morpho",
    layer: str = "X",
    direction: str = "forward",
    interpolation_num: int = 250,
    t_end: Optional[Union[int, float]] = None,
    average: bool = False,
    cores: int = 1,
    nonrigid_only: bool = False,
    inplace: bool = True,
    **kwargs,
):
    adata = adata if inplace else adata.copy()
    if vf_key not in adata.uns.keys():
        raise Exception(
            f"The {vf_key} that corresponds to the reconstructed vector field is not in ``anndata.uns``."
            f"Please run ``st.tdr.morphofield_gp`` or ``st.tdr.morphofield_sparsevfc`` before fate prediction."
        )
    vf_dict = adata.uns[vf_key]
    if vf_dict["method"] not in ["gaussian_process", "sparsevfc"]:
        raise Exception(
            f"The method for vector field  reconstruction is not in avaliable."
            f"Please re-run ``st.tdr.morphofield_gp`` or ``st.tdr.morphofield_sparsevfc`` before fate prediction."
        )
    init_states = np.asarray(vf_dict["X"], dtype=float)
    t, pred = integrate_field(vf_dict, init_states, t_end=t_end, interpolation_num=interpolation_num,
                              direction=direction, average=average, nonrigid_only=nonrigid_only, **kwargs)
    n = len(pred)
    adata.uns[key_added] = {
        "init_states": init_states,
        "init_cells": _obs_names(adata, len(init_states)),
        "average": average,
        "genes": None,
        "t": {i: t[i] for i in range(n)},
        "prediction": {i: pred[i] for i in range(n)},
    }
    return None if inplace else adata
