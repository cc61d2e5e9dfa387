"""Shared by the CPU (test-double) and GPU tests of the Gaussian-process morphofield variant: rebuilds the vf dict the
goldens were generated from (tests/golden/make_golden.py) and checks every AnnData slot against the reference output."""
import numpy as np


def gp_dict(g):
    return {
        "norm_dict": {
            "scale_fixed": float(g["gp_nd_scale_fixed"]),
            "scale_transformed": float(g["gp_nd_scale_transformed"]),
            "mean_transformed": g["gp_nd_mean_transformed"],
            "mean_fixed": g["gp_nd_mean_fixed"],
        },
        "kernel_type": "euc",
        "inducing_variables": g["gp_Xc"],
        "beta": float(g["gp_beta"]),
        "Coff": g["gp_C"],
        "R": g["gp_R"],
        "t": g["gp_t"],
    }


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300))


def run_and_check(st, g, tol):
    for tag, nro in (("full", False), ("nr", True)):
        ad = st.AnnDataLite(obsm={"align_spatial": g["gpw_X"]})
        ad.uns["VecFld_morpho"] = gp_dict(g)
        assert st.tdr.morphofield_gp(ad, NX=g["gpw_X"][:6] + 1.5, nonrigid_only=nro) is None
        vf = ad.uns["VecFld_morpho"]
        assert vf["method"] == "gaussian_process"
        assert _rel(vf["V"], g[f"gpw_{tag}_V"]) < tol
        np.testing.assert_array_equal(vf["grid"], g[f"gpw_{tag}_grid"])
        assert _rel(vf["grid_V"], g[f"gpw_{tag}_grid_V"]) < tol
        for fn in (st.tdr.morphofield_velocity, st.tdr.morphofield_acceleration, st.tdr.morphofield_curvature,
                   st.tdr.morphofield_curl, st.tdr.morphofield_torsion, st.tdr.morphofield_divergence,
                   st.tdr.morphofield_jacobian):
            assert fn(ad, nonrigid_only=nro) is None
        assert _rel(ad.obsm["velocity"], g[f"gpw_{tag}_velocity"]) < tol
        assert _rel(ad.obs["acceleration"], g[f"gpw_{tag}_acc_obs"]) < tol
        assert _rel(ad.obsm["acceleration"], g[f"gpw_{tag}_acc_obsm"]) < tol
        assert _rel(ad.obs["curvature"], g[f"gpw_{tag}_curv_obs"]) < tol
        assert _rel(ad.obsm["curvature"], g[f"gpw_{tag}_curv_obsm"]) < tol
        assert ad.obsm["curl"].shape == g[f"gpw_{tag}_curl_obsm"].shape
        assert _rel(ad.obs["curl"], g[f"gpw_{tag}_curl_obs"]) < tol
        assert _rel(ad.obsm["curl"], g[f"gpw_{tag}_curl_obsm"]) < tol
        assert _rel(ad.obs["torsion"], g[f"gpw_{tag}_tor_obs"]) < 100 * tol
        assert _rel(ad.uns["torsion"], g[f"gpw_{tag}_tor_uns"]) < 100 * tol
        assert _rel(ad.obs["divergence"], g[f"gpw_{tag}_div_obs"]) < tol
        assert _rel(ad.uns["jacobian"], g[f"gpw_{tag}_jac_uns"]) < tol
        assert _rel(ad.obs["jacobian"], g[f"gpw_{tag}_jac_obs"]) < 100 * tol


def check_axes_and_2d(st, g, tol, **kw):
    """The two GP shapes round 5 refused, against outputs of the REAL reference functions (tests/golden/make_golden_gp_axes.py ->
    ref_gp_axes.npz): per-axis ``norm_dict`` scales (velocities; the Jacobian raises, as the reference's own does) and a 2-D
    field (velocities and Jacobian)."""
    import pytest

    ax = {"norm_dict": {"scale_fixed": g["ax_sf"], "scale_transformed": g["ax_stt"], "mean_transformed": g["ax_mean_t"],
                        "mean_fixed": g["ax_mean_f"]},
          "kernel_type": "euc", "inducing_variables": g["ax_ind"], "beta": float(g["ax_beta"]), "Coff": g["ax_C"],
          "R": g["ax_R"], "t": g["ax_t"]}
    assert _rel(st.vectorfield.gp_velocity(g["ax_X"], ax, **kw), g["ax_V_full"]) < tol
    assert _rel(st.vectorfield.gp_velocity(g["ax_X"], ax, nonrigid_only=True, **kw), g["ax_V_nr"]) < tol
    assert bool(g["ax_jac_raises"])
    vf = st.GPVectorField(**kw)
    vf.vf_dict = ax
    with pytest.raises(ValueError, match="broadcast"):
        vf.get_Jacobian()(g["ax_X"])
    d2 = {"norm_dict": {"scale_fixed": float(g["d2_sf"]), "scale_transformed": float(g["d2_stt"]),
                        "mean_transformed": g["d2_mean_t"], "mean_fixed": g["d2_mean_f"]},
          "kernel_type": "euc", "inducing_variables": g["d2_ind"], "beta": float(g["d2_beta"]), "Coff": g["d2_C"],
          "R": g["d2_R"], "t": g["d2_t"]}
    v = st.vectorfield.gp_velocity(g["d2_X"], d2, **kw)
    assert v.shape == g["d2_V_full"].shape and _rel(v, g["d2_V_full"]) < tol
    assert _rel(st.vectorfield.gp_velocity(g["d2_X"], d2, nonrigid_only=True, **kw), g["d2_V_nr"]) < tol
    vf2 = st.GPVectorField(**kw)
    vf2.vf_dict = d2
    J = vf2.get_Jacobian()(g["d2_X"])
    assert J.shape == g["d2_J"].shape and _rel(J, g["d2_J"]) < tol
