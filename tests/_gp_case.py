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
