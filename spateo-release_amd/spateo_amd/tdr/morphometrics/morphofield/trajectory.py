"""``morphopath`` with the reference's signature (``spateo/tdr/morphometrics/morphofield/trajectory.py:11-117``):
predicts cell trajectories by integrating the learned morphometric vector field.

The reference hands the field to dynamo's ``fate`` (third-party, not in the reference tree: SciPy ``solve_ivp`` RK45
with ``max_step = t_end / interpolation_num`` and a terminal event where the field is at rest, then resampling to
``interpolation_num`` points equally spaced in ARC LENGTH, evaluated on the solver's dense output).  Here the
integration runs in ONE fused HIP kernel (``mvf_integrate``: classical RK4, one lane per trajectory, 8 steps per output
interval) and the same semantics are applied to its finely sampled paths (``vectorfield.integrate_field``,
``sampling="arc_length"``; ``sampling="uniform_time"`` is available through ``**kwargs``).  dynamo's source is not
available here, so its restatement (``oracle/trajectory_oracle.py``) is parity-unpinned; agreement is to the
reference solver's own tolerance (rtol 1e-3).  Output slots are the reference's: ``uns[key_added]["t"][i]`` (times) and
``uns[key_added]["prediction"][i]`` ((n_t, d) states: the reference transposes dynamo ``fate``'s (d, n_t) arrays,
``trajectory.py:113``, and its consumer concatenates ``init_states[[i]]`` with it along axis 0,
``tdr/models/models_migration/morphopath_model.py:225``) per cell; ``init_cells`` = the cells' ``obs_names``."""
from __future__ import annotations

from typing import Optional, Union

import numpy as np

from ....vectorfield import genesis_states, integrate_field


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


def construct_genesis_states(adata, fate_key: str = "fate_morpho", n_steps: int = 100, logspace: bool = False,
                             t_end: Optional[Union[int, float]] = None, **kwargs):
    """The computation inside ``construct_genesis`` (``spateo/tdr/models/models_migration/morphopath_model.py:84-154``)
    without its PyVista model building (out of scope): the time vector is derived from the fate prediction exactly as
    the reference does (``:123-135``, including its hard-coded ``adata.uns["fate_morpho"]`` and the integer truncation
    of the times), the field is the one behind ``fate_key`` (``VecFld_<fate_key[5:]>``, ``:137-138``), and the cells are
    displaced step by step (``:140-148``).  Returns ``(stages_X, time_vec)``: the list the reference hands to
    ``construct_genesis_X`` and the step durations."""
    if fate_key not in adata.uns.keys():
        raise Exception(
            f"You need to first perform develop_trajectory prediction before animate the prediction, please run"
            f"st.tdr.develop_trajectory(adata, key_added='{fate_key}' before running this function"
        )
    t_ind = np.asarray(list(adata.uns[fate_key]["t"].keys()), dtype=int)
    t_sort_ind = np.argsort(t_ind)
    t = [list(adata.uns["fate_morpho"]["t"].values())[i] for i in t_sort_ind]
    flats = np.unique([int(item) for sublist in t for item in sublist])
    flats = np.hstack((0, flats))
    flats = np.sort(flats) if t_end is None else np.sort(flats[flats <= t_end])
    time_vec = (
        np.logspace(0, np.log10(max(flats) + 1), n_steps) - 1
        if logspace
        else flats[(np.linspace(0, len(flats) - 1, n_steps)).astype(int)]
    )
    vf_key = "VecFld_%s" % fate_key[5:]
    if vf_key not in adata.uns.keys():
        raise ValueError(f"Vector field function {vf_key} is not included in the adata object!")
    stages = genesis_states(adata.uns[vf_key], adata.uns[fate_key]["init_states"], time_vec, **kwargs)
    return stages, np.asarray(time_vec, dtype=float)
