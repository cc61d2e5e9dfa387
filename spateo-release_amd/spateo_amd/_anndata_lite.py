"""Minimal AnnData stand-in.

The reference wrappers only touch ``.obsm``, ``.obs``, ``.uns``, ``.n_obs`` and ``.copy()`` of the AnnData object
(``sparsevfc.py:313-316``, ``differential_geometry.py:62-70,334-339``).  ``anndata`` is not installable in the build
image, so the wrappers duck-type their argument; a real ``anndata.AnnData`` works unchanged, and this class lets the
path run (and be tested) where ``anndata`` is absent.
"""
from __future__ import annotations

import copy

import numpy as np


class _ObsFrame(dict):
    """dict of per-cell columns; assignment coerces to a 1-D NumPy array like ``DataFrame.__setitem__`` would."""

    def __setitem__(self, key, value):
        super().__setitem__(key, np.asarray(value))


class AnnDataLite:
    def __init__(self, obsm=None, obs=None, uns=None, n_obs=None, X=None, var_names=None, layers=None):
        self.X = None if X is None else np.asarray(X)
        self.var_names = list(var_names) if var_names is not None else (
            [str(i) for i in range(self.X.shape[1])] if self.X is not None else [])
        self.layers = dict(layers or {})
        self.obsm = dict(obsm or {})
        self.obs = _ObsFrame()
        for k, v in (obs or {}).items():
            self.obs[k] = v
        self.uns = dict(uns or {})
        if n_obs is None:
            n_obs = len(self.X) if self.X is not None else (len(next(iter(self.obsm.values()))) if self.obsm else 0)
        self.n_obs = int(n_obs)

    def copy(self):
        return copy.deepcopy(self)
