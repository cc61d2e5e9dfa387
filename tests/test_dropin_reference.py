"""Drop-in proof (runs where /root/reference exists, i.e. in the build container; skipped on the GPU box).

The REAL reference files ``spateo/tdr/morphometrics/morphofield/sparsevfc.py``,
``.../morphofield_dg/differential_geometry.py`` and ``spateo/tdr/interpolations/interpolation_sparseVFC.py`` are loaded
by path and executed with the seam they import from dynamo -
``dynamo.vectorfield.scVectorField.{SparseVFC, SvcVectorField}`` - bound to THIS repo's ``spateo_amd.vectorfield``
(INTEGRATION.md's import swap, applied for real).  Their outputs must equal the goldens of ``tests/golden/ref_twins.npz``,
which the same reference files produced with the float64 oracle at that seam.  The device is the oracle-backed CPU
double here (there is no GPU in the build container); the ``-m gpu`` suite runs the same wrappers of this repo's mirror
package on the real kernels against the same goldens.
"""
import os
import sys
import types

import numpy as np
import pandas as pd
import pytest

REF = "/root/reference/spateo"
pytestmark = pytest.mark.skipif(not os.path.isdir(REF), reason="the reference tree only exists in the build container")

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))


@pytest.fixture
def reference_with_our_engine(monkeypatch):
    """(sparsevfc module, differential_geometry module, interpolation_sparseVFC module) of the REAL reference, importing
    SparseVFC / SvcVectorField from spateo_amd.vectorfield."""
    import make_golden as mg
    from _cpu_kernels import CpuKernels
    from spateo_amd import vectorfield as vfm

    before = set(sys.modules)
    monkeypatch.setattr(vfm._rt, "_make_kernels", lambda device, dtype: CpuKernels(device, dtype))
    mg.install_stubs()
    sv = types.ModuleType("dynamo.vectorfield.scVectorField")
    sv.SparseVFC = vfm.SparseVFC          # <- the swap: this repo's engine behind dynamo's name
    sv.SvcVectorField = vfm.SvcVectorField
    sys.modules["dynamo.vectorfield.scVectorField"] = sv
    sys.modules["dynamo.vectorfield"].SvcVectorField = vfm.SvcVectorField
    interp_pkg = mg._pkg("spateo.tdr.interpolations")
    iu = mg._load("spateo.tdr.interpolations.utils", "spateo/tdr/interpolations/utils.py")
    interp_pkg.get_X_Y_grid = iu.get_X_Y_grid
    svfc = mg._load("spateo.tdr.morphometrics.morphofield.sparsevfc", "spateo/tdr/morphometrics/morphofield/sparsevfc.py")
    mg._load("spateo.tdr.morphometrics.morphofield.gaussian_process",
             "spateo/tdr/morphometrics/morphofield/gaussian_process.py")
    mg._load("spateo.tdr.morphometrics.morphofield_dg.GPVectorField",
             "spateo/tdr/morphometrics/morphofield_dg/GPVectorField.py")
    dg = mg._load("spateo.tdr.morphometrics.morphofield_dg.differential_geometry",
                  "spateo/tdr/morphometrics/morphofield_dg/differential_geometry.py")
    ki = mg._load("spateo.tdr.interpolations.interpolation_sparseVFC", "spateo/tdr/interpolations/interpolation_sparseVFC.py")
    yield svfc, dg, ki
    for name in set(sys.modules) - before:
        del sys.modules[name]
    for name in ("anndata",):
        sys.modules.pop(name, None)


def _close(a, b, tol=1e-7):
    a, b = np.asarray(a, dtype=float), np.asarray(b, dtype=float)
    assert a.shape == b.shape, (a.shape, b.shape)
    assert np.abs(a - b).max() <= tol * max(np.abs(b).max(), 1e-300), float(np.abs(a - b).max() / np.abs(b).max())


def test_real_reference_wrappers_run_on_this_engine(reference_with_our_engine, golden):
    svfc, dg, _ = reference_with_our_engine
    from spateo_amd._anndata_lite import AnnDataLite

    g = golden
    Xw, Vw = g["w_X"], g["w_V"]
    # _morphofield_sparsevfc: grid, restart loop, acceptance metric (sparsevfc.py:103-238)
    res = svfc._morphofield_sparsevfc(Xw[:300], Vw[:300], NX=None, grid_num=[5, 4, 3], M=30, lambda_=0.02,
                                      lstsq_method="scipy", min_vel_corr=0.5, restart_num=3, restart_seed=[0, 100, 200],
                                      MaxIter=30)
    assert res["method"] == "sparsevfc"
    for k in ["valid_ind", "X_ctrl", "ctrl_idx", "grid", "iteration"]:
        np.testing.assert_array_equal(np.asarray(res[k]), g[f"w1_{k}"])
    for k in ["beta", "V", "P", "sigma2", "grid_V", "tecr_traj", "E_traj"]:
        _close(res[k], g[f"w1_{k}"], 1e-6)
    res2 = svfc._morphofield_sparsevfc(Xw[300:], Vw[300:].copy(), NX=Xw[300:310], M=12, min_vel_corr=2.0, restart_num=2,
                                       restart_seed=(0, 100, 200, 300, 400), MaxIter=8)
    for k in ["X_ctrl", "V", "grid_V", "iteration", "sigma2"]:
        _close(res2[k], g[f"w2_{k}"], 1e-6)
    # AnnData wrapper + the seven morphofield_* evaluators (differential_geometry.py:42-341)
    ad = AnnDataLite(obsm={"align_spatial": g["a_X"], "V_mapping": g["a_V"]})
    svfc.morphofield_sparsevfc(ad, NX=g["a_X"][:5], M=15, MaxIter=20, restart_num=1, restart_seed=[0])
    for k in ["X_ctrl", "beta", "V", "grid_V"]:
        _close(ad.uns["VecFld_morpho"][k], g[f"a_vf_{k}"], 1e-6)
    # evaluate on the GOLDEN coefficients (C is only determined up to the solve's null space; the evaluators are linear in it)
    ad.uns["VecFld_morpho"]["C"] = g["a_vf_C"]
    for fn in (dg.morphofield_velocity, dg.morphofield_acceleration, dg.morphofield_curvature, dg.morphofield_curl,
               dg.morphofield_torsion, dg.morphofield_divergence, dg.morphofield_jacobian):
        fn(ad)
    _close(ad.obsm["velocity"], g["a_velocity"])
    _close(ad.obs["acceleration"], g["a_acceleration_obs"]), _close(ad.obsm["acceleration"], g["a_acceleration_obsm"])
    _close(ad.obs["curvature"], g["a_curvature_obs"]), _close(ad.obsm["curvature"], g["a_curvature_obsm"])
    _close(ad.obs["curl"], g["a_curl_obs"]), _close(ad.obsm["curl"], g["a_curl_obsm"])
    _close(ad.obs["torsion"], g["a_torsion_obs"], 1e-6), _close(ad.uns["torsion"], g["a_torsion_uns"], 1e-6)
    _close(ad.obs["divergence"], g["a_divergence_obs"])
    _close(ad.obs["jacobian"], g["a_jacobian_obs"]), _close(ad.uns["jacobian"], g["a_jacobian_uns"])


class _MiniAnnData:
    """Just enough of anndata.AnnData for the real kernel_interpolation (interpolation_sparseVFC.py:13-85)."""

    def __init__(self, X=None, obs=None, obsm=None, var=None, layers=None):
        self.X = None if X is None else np.asarray(X)
        self.obs = obs if obs is not None else pd.DataFrame(index=range(0 if X is None else len(X)))
        self.obsm = dict(obsm or {})
        self.var = var if var is not None else pd.DataFrame(index=[str(i) for i in range(0 if X is None else self.X.shape[1])])
        self.layers = dict(layers or {})

    @property
    def var_names(self):
        return self.var.index

    def copy(self):
        return _MiniAnnData(None if self.X is None else self.X.copy(), self.obs.copy(), {k: v.copy() for k, v in self.obsm.items()},
                            self.var.copy(), {k: v.copy() for k, v in self.layers.items()})

    def __getitem__(self, idx):
        rows, cols = idx
        assert rows == slice(None)
        j = [list(self.var.index).index(c) for c in cols]
        return _MiniAnnData(self.X[:, j], self.obs, self.obsm, self.var.iloc[j])


def test_real_kernel_interpolation_runs_on_this_engine(reference_with_our_engine):
    """The second SparseVFC call site (wide Y: obs keys + genes) with the real wrapper; this repo's engine vs the oracle
    behind the same wrapper, and vs this repo's mirror of the wrapper."""
    _, _, ki = reference_with_our_engine
    import spateo_amd as st
    from oracle import sparsevfc_oracle as svo

    rng = np.random.default_rng(7)
    n = 260
    X = rng.uniform(-1, 1, (n, 3)) * np.array([30.0, 20.0, 15.0])
    genes = np.column_stack([np.sin(X[:, 0] / 9 + j) + 0.2 * np.cos(X[:, 1] / 7 * (j + 1)) for j in range(4)])
    genes += 0.02 * rng.standard_normal(genes.shape)
    obs = pd.DataFrame({"area": np.cos(X[:, 2] / 6) + 2.0})
    ad = _MiniAnnData(X=genes, obs=obs, obsm={"spatial": X}, var=pd.DataFrame(index=[f"g{j}" for j in range(4)]))
    target = X[::13] + 0.3
    sys.modules["anndata"].AnnData = _MiniAnnData
    ki.AnnData = _MiniAnnData
    kw = dict(spatial_key="spatial", keys=["area", "g0", "g2", "g3"], target_points=target, lambda_=3.0, M=25, MaxIter=15)
    ours = ki.kernel_interpolation(ad, **kw)
    ki.SparseVFC = svo.SparseVFC  # the same real wrapper with the oracle at the seam
    ref = ki.kernel_interpolation(ad, **kw)
    _close(ours.X, ref.X, 1e-7)
    _close(ours.obs["area"].values, ref.obs["area"].values, 1e-7)
    np.testing.assert_array_equal(ours.obsm["spatial"], target)
    assert list(ours.var_names) == ["g0", "g2", "g3"]
