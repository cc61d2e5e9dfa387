"""CPU tests of the product's HOST logic with the device replaced by the oracle-backed test double
(tests/_cpu_kernels.py): preprocessing, the EM driver (energy, tecr, gamma, stopping rule, jitter escalation), the
restart loop and the AnnData wrappers against the goldens produced by the REAL reference wrappers, sharding."""
import numpy as np
import pytest

import spateo_amd as st
from spateo_amd import vectorfield as vfm
from oracle import sparsevfc_oracle as svo

from _cpu_kernels import CpuKernels


@pytest.fixture
def cpu_kernels(monkeypatch):
    monkeypatch.setattr(vfm._rt, "_make_kernels", lambda device, dtype: CpuKernels(device, dtype))
    return CpuKernels()


def _rel(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / np.abs(b).max())


def _data(n=600, seed=0):
    from spateo_amd._synthetic import make_config

    X, V, _ = make_config("C2", N=n)
    return X, V


# ------------------------------------------------------------------------------------------------ preprocessing
@pytest.mark.parametrize("vbs", [True, False])
def test_preprocess_matches_oracle_bit_for_bit(vbs):
    X, V = _data(800)
    V[3] = np.nan
    X[20] = X[21]
    got = vfm.sparsevfc_preprocess(X, V, M=60, seed=7, velocity_based_sampling=vbs)
    ref = svo.sparsevfc_setup(X, V, M=60, seed=7, velocity_based_sampling=vbs)
    for g, r in zip(got, ref):
        np.testing.assert_array_equal(g, r) if isinstance(r, np.ndarray) else None
    assert got[5] == pytest.approx(ref[5], rel=1e-13)  # beta: cKDTree vs sklearn kd-tree, both exact kNN


def test_preprocess_clips_M_to_unique_rows_and_rejects_all_nan():
    X = np.repeat(np.arange(12.0).reshape(4, 3), 3, axis=0)
    Y = np.ones_like(X)
    valid, Xv, Yv, idx, ctrl, beta = vfm.sparsevfc_preprocess(X, Y, M=10, beta=0.5)
    assert ctrl.shape == (4, 3) and beta == 0.5
    with pytest.raises(ValueError, match="no row of Y is finite"):
        vfm.sparsevfc_preprocess(X, np.full_like(Y, np.nan), M=3)


def test_shard_bounds_partition():
    for n, w in [(10, 3), (7, 8), (8_000_000, 8), (5, 1)]:
        b = [vfm.shard_bounds(n, r, w) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        assert max(hi - lo for lo, hi in b) - min(hi - lo for lo, hi in b) <= 1


# ------------------------------------------------------------------------------------------------ EM driver
def test_em_driver_reproduces_oracle_sparsevfc(cpu_kernels):
    X, V = _data(700)
    Grid = X[::25] + 0.5
    kw = dict(M=40, lambda_=3.0, lstsq_method="scipy", MaxIter=40, seed=0)
    ref = svo.SparseVFC(X, V, Grid, **kw)
    got = st.SparseVFC(X, V, Grid, **kw)
    assert set(got) == set(ref) | {"tecr_vec"}  # + the name Spateo's docstring gives tecr_traj (sparsevfc.py:155)
    np.testing.assert_array_equal(got["tecr_vec"], got["tecr_traj"])
    assert got["iteration"] == ref["iteration"] and len(got["E_traj"]) == got["iteration"] + 1
    np.testing.assert_array_equal(got["ctrl_idx"], ref["ctrl_idx"])
    np.testing.assert_array_equal(got["valid_ind"], ref["valid_ind"])
    assert _rel(got["V"], ref["V"]) < 1e-8 and _rel(got["grid_V"], ref["grid_V"]) < 1e-8
    np.testing.assert_allclose(got["E_traj"], ref["E_traj"], rtol=1e-8)
    np.testing.assert_allclose(got["tecr_traj"], ref["tecr_traj"], rtol=1e-4, atol=1e-10)
    np.testing.assert_allclose(got["sigma2"], ref["sigma2"], rtol=1e-8)
    np.testing.assert_allclose(got["P"], ref["P"], rtol=1e-6, atol=1e-12)
    np.testing.assert_array_equal(got["VFCIndex"], ref["VFCIndex"])
    assert got["P"].shape == (700, 1) and got["C"].shape == (40, 3)


@pytest.mark.parametrize("dy", [1, 2, 4, 7])
def test_em_driver_wide_and_narrow_outputs(cpu_kernels, dy):
    """Dy != D (kernel_interpolation's use of SparseVFC): Y is processed in 3-column groups sharing one Gram matrix."""
    rng = np.random.default_rng(dy)
    X, V = _data(500)
    Y = np.column_stack([np.sin(X[:, 0] / 80 + j) + 0.3 * np.cos(X[:, 1] / 60 * (j + 1)) for j in range(dy)])
    Y += 0.02 * rng.standard_normal(Y.shape)
    kw = dict(M=30, lambda_=3.0, lstsq_method="scipy", MaxIter=12, seed=0)
    ref = svo.SparseVFC(X, Y, X[::20], **kw)
    got = st.SparseVFC(X, Y, X[::20], **kw)
    assert got["V"].shape == (500, dy) and got["C"].shape == (30, dy) and got["grid_V"].shape == (25, dy)
    assert got["iteration"] == ref["iteration"]
    assert _rel(got["V"], ref["V"]) < 1e-8 and _rel(got["grid_V"], ref["grid_V"]) < 1e-8
    np.testing.assert_allclose(got["E_traj"], ref["E_traj"], rtol=1e-8)
    np.testing.assert_allclose(got["P"], ref["P"], rtol=1e-6, atol=1e-12)


def test_em_driver_stops_like_the_reference(cpu_kernels):
    X, V = _data(300)
    for kw in (dict(MaxIter=3), dict(MaxIter=50, ecr=1e-2), dict(MaxIter=1)):
        full = dict(M=20, lambda_=3.0, lstsq_method="scipy", seed=0, **kw)
        ref = svo.SparseVFC(X, V, None, **full)
        got = st.SparseVFC(X, V, None, **full)
        assert got["iteration"] == ref["iteration"], kw
        assert got["grid_V"] is None and got["grid"] is None


def test_solve_policy_cholesky_while_certified_then_minimum_norm(cpu_kernels):
    """lstsq_method="scipy": un-regularised Cholesky while the pivots certify full numerical rank, the truncated
    minimum-norm solve once they do not (sticky within a fit): the rank-revealing one by default (the factor rank of one
    step is the next step's hint), the full-width one (mn_method = "full") with a shift that escalates only on failure."""
    from spateo_amd.vectorfield import SparseVFCEngine

    X, V = _data(400)
    valid, Xv, Yv, idx, ctrl, beta = vfm.sparsevfc_preprocess(X, V, M=30, seed=0)
    eng = SparseVFCEngine(Xv, Yv, ctrl, beta, kernels=cpu_kernels)
    eng.init_state()
    eng.em_step(lambda_=3.0)
    assert eng.solver_stats["cholesky"] == 1 and eng.solver_stats["minnorm"] == 0 and not eng.rank_deficient

    class TinyPivot(CpuKernels):
        chol = mn = 0

        def solve(self, G, K, ls2, jitter, R, C_out, info, pivots=None):
            TinyPivot.chol += 1
            super().solve(G, K, ls2, jitter, R, C_out, info, pivots)
            if pivots is not None:
                pivots[0] = 1e-14 * pivots[1]  # a pivot at rounding level: full rank is NOT certified

        def solve_minnorm(self, *a, **kw):
            TinyPivot.mn += 1
            super().solve_minnorm(*a, **kw)

        def solve_minnorm_lr(self, *a, **kw):
            TinyPivot.mn += 1
            TinyPivot.hints.append(kw.get("rank_hint"))
            super().solve_minnorm_lr(*a, **kw)

    for method in ("lowrank", "full"):
        TinyPivot.chol = TinyPivot.mn = 0
        TinyPivot.hints = []
        eng2 = SparseVFCEngine(Xv, Yv, ctrl, beta, kernels=TinyPivot())
        eng2.mn_method = method
        eng2.init_state()
        eng2.em_step(lambda_=3.0)
        assert eng2.rank_deficient and (TinyPivot.chol, TinyPivot.mn) == (1, 1)
        eng2.em_step(lambda_=3.0)
        assert (TinyPivot.chol, TinyPivot.mn) == (1, 2)  # sticky: no more Cholesky attempts in this fit
        assert eng2.solver_stats["minnorm"] == 2 and eng2.solver_stats["rank"] == [30, 30]
        assert TinyPivot.hints == ([0, 30] if method == "lowrank" else [])
        # well conditioned system: every path gives the same field
        np.testing.assert_allclose(eng2.results()[0], _two_steps(Xv, Yv, ctrl, beta, cpu_kernels), rtol=1e-9, atol=1e-12)

    # the second witness of the full-rank certificate: pivots that look fine over a lambda_min far below eps lambda_max
    # (what a Kahan-type matrix does to an unpivoted Cholesky factorisation) - the inverse-iteration probes riding along as
    # extra right-hand sides see 1 / lambda_min and refuse the certificate
    class HiddenNullDirection(CpuKernels):
        chol = mn = 0
        widths = []

        def solve(self, G, K, ls2, jitter, R, C_out, info, pivots=None):
            HiddenNullDirection.chol += 1
            HiddenNullDirection.widths.append(R.shape[1])
            super().solve(G, K, ls2, jitter, R, C_out, info, pivots)
            if pivots is not None:
                q = torch.ones(len(C_out), 1, dtype=torch.float64) / np.sqrt(len(C_out))
                C_out[:, 3:] += 1e19 * float(pivots[1]) ** -1 * q * (q.T @ R[:, 3:])  # = a component along lambda_min ~ 1e-19 lambda_max

        def solve_minnorm(self, *a, **kw):
            HiddenNullDirection.mn += 1
            super().solve_minnorm(*a, **kw)

    import torch

    eng3 = SparseVFCEngine(Xv, Yv, ctrl, beta, kernels=HiddenNullDirection())
    eng3.mn_method = "full"
    eng3.init_state()
    eng3.em_step(lambda_=3.0)
    assert HiddenNullDirection.widths == [5]            # 3 field columns + the 2 probe columns in ONE factorisation
    assert eng3.rank_deficient and (HiddenNullDirection.chol, HiddenNullDirection.mn) == (1, 1)

    class ShiftTooSmall(CpuKernels):
        shifts = []

        def solve(self, G, K, ls2, jitter, R, C_out, info, pivots=None):
            info.fill_(3)  # non-positive pivot without regularisation

        def solve_minnorm(self, G, K, ls2, shift, R, C_out, info, einfo, **kw):
            ShiftTooSmall.shifts.append(shift)
            if len(ShiftTooSmall.shifts) <= 2:
                info.fill_(7)
                return
            super().solve_minnorm(G, K, ls2, shift, R, C_out, info, einfo, **kw)

    eng3 = SparseVFCEngine(Xv, Yv, ctrl, beta, kernels=ShiftTooSmall())
    eng3.mn_method = "full"
    eng3.init_state()
    eng3.em_step(lambda_=3.0)
    assert ShiftTooSmall.shifts == [2.0 ** -36, 2.0 ** -32, 2.0 ** -28] and eng3.mn_shift == 2.0 ** -28

    class AlwaysFail(CpuKernels):
        def solve(self, G, K, ls2, jitter, R, C_out, info, pivots=None):
            info.fill_(1)

        def solve_minnorm(self, G, K, ls2, shift, R, C_out, info, einfo, **kw):
            info.fill_(1)

        def solve_minnorm_lr(self, G, K, ls2, R, C_out, info, einfo, **kw):
            info.fill_(1)

    for method, msg in (("full", "not numerically positive semi-definite"), ("lowrank", "non-finite")):
        eng4 = SparseVFCEngine(Xv, Yv, ctrl, beta, kernels=AlwaysFail())
        eng4.mn_method = method
        eng4.init_state()
        with pytest.raises(RuntimeError, match=msg):
            eng4.em_step(lambda_=3.0)


def _two_steps(Xv, Yv, ctrl, beta, kernels):
    from spateo_amd.vectorfield import SparseVFCEngine

    e = SparseVFCEngine(Xv, Yv, ctrl, beta, kernels=kernels)
    e.init_state()
    e.em_step(lambda_=3.0)
    e.em_step(lambda_=3.0)
    return e.results()[0]


def test_cholesky_mode_jitter_escalates_only_on_failed_pivots(cpu_kernels):
    """lstsq_method="cholesky" (non-reference fast mode): jitter 0 -> 1e-15 -> x10 ..., sticky."""
    from spateo_amd.vectorfield import SparseVFCEngine

    X, V = _data(400)
    valid, Xv, Yv, idx, ctrl, beta = vfm.sparsevfc_preprocess(X, V, M=30, seed=0)

    class FailTwice(CpuKernels):
        calls = 0

        def solve(self, G, K, ls2, jitter, R, C_out, info, pivots=None):
            FailTwice.calls += 1
            if FailTwice.calls <= 2:
                info.fill_(5)
                return
            super().solve(G, K, ls2, jitter, R, C_out, info, pivots)

    eng2 = SparseVFCEngine(Xv, Yv, ctrl, beta, kernels=FailTwice())
    eng2.lstsq_method = "cholesky"
    eng2.init_state()
    eng2.em_step(lambda_=3.0)
    assert eng2.solve_retries == 2 and eng2.jitter == pytest.approx(1e-14)  # 0 -> 1e-15 -> 1e-14, then sticky
    eng2.em_step(lambda_=3.0)
    assert eng2.jitter == pytest.approx(1e-14)

    class AlwaysFail(CpuKernels):
        def solve(self, G, K, ls2, jitter, R, C_out, info, pivots=None):
            info.fill_(1)

    eng3 = SparseVFCEngine(Xv, Yv, ctrl, beta, kernels=AlwaysFail())
    eng3.lstsq_method = "cholesky"
    eng3.init_state()
    with pytest.raises(RuntimeError, match="non-positive pivot"):
        eng3.em_step(lambda_=3.0)


def test_lstsq_method_is_honoured_or_warned(cpu_kernels):
    X, V = _data(300)
    kw = dict(M=20, MaxIter=3, lambda_=3.0)
    a = vfm.SparseVFC(X, V, None, lstsq_method="scipy", **kw)
    vfm._rt._LSTSQ_WARNED.clear()
    with pytest.warns(RuntimeWarning, match="normal-equations arithmetic of 'drouin' is not reproduced"):
        b = vfm.SparseVFC(X, V, None, lstsq_method="drouin", **kw)
    np.testing.assert_array_equal(a["V"], b["V"])
    c = vfm.SparseVFC(X, V, None, lstsq_method="cholesky", **kw)
    np.testing.assert_allclose(a["V"], c["V"], rtol=1e-9, atol=1e-12)


def test_engine_rejects_unsupported_shapes(cpu_kernels):
    from spateo_amd.vectorfield import SparseVFCEngine

    X = np.zeros((10, 4))
    with pytest.raises(NotImplementedError, match="spatial dimensions"):
        SparseVFCEngine(X, X[:, :3], X[:3], 0.1, kernels=cpu_kernels)
    with pytest.raises(ValueError):
        SparseVFCEngine(X[:, :3], X[:5, :3], X[:3, :3], 0.1, kernels=cpu_kernels)
    with pytest.raises(NotImplementedError):
        st.SparseVFC(X[:, :3], X[:, :3], None, div_cur_free_kernels=True)


# ------------------------------------------------------------------------------------------------ wrappers vs goldens
def test_get_X_Y_grid_matches_reference(golden):
    g = golden
    X, Y, Grid, in_hull = st.tdr.get_X_Y_grid(X=g["grid_X"].copy(), Y=g["grid_X"].copy(), grid_num=[4, 5, 6])
    np.testing.assert_array_equal(Grid, g["grid_Grid"])
    np.testing.assert_array_equal(in_hull, g["grid_in_hull"])
    assert Grid.shape == (120, 3)


def test_restart_loop_matches_reference_wrapper(golden, cpu_kernels):
    g = golden
    res = st.tdr._morphofield_sparsevfc(
        g["w_X"][:300], g["w_V"][:300], NX=None, grid_num=[5, 4, 3], M=30, lambda_=0.02, lstsq_method="scipy",
        min_vel_corr=0.5, restart_num=3, restart_seed=[0, 100, 200], MaxIter=30)
    assert res["method"] == "sparsevfc"
    for k in ["valid_ind", "X_ctrl", "ctrl_idx", "grid", "VFCIndex"]:
        np.testing.assert_array_equal(res[k], g[f"w1_{k}"])
    assert res["iteration"] == int(g["w1_iteration"])
    assert _rel(res["V"], g["w1_V"]) < 1e-8 and _rel(res["grid_V"], g["w1_grid_V"]) < 1e-8
    np.testing.assert_allclose(res["E_traj"], g["w1_E_traj"], rtol=1e-8)
    # forced restarts + default seed-length quirk (5 seeds for restart_num=2 -> seeds become arange(2)*100)
    Xf, Vf = g["w_X"][300:], g["w_V"][300:]
    res2 = st.tdr._morphofield_sparsevfc(Xf, Vf, NX=Xf[:10], M=12, min_vel_corr=2.0, restart_num=2,
                                         restart_seed=(0, 100, 200, 300, 400), MaxIter=8)
    np.testing.assert_array_equal(res2["X_ctrl"], g["w2_X_ctrl"])
    assert res2["iteration"] == int(g["w2_iteration"])
    assert _rel(res2["V"], g["w2_V"]) < 1e-8 and _rel(res2["grid_V"], g["w2_grid_V"]) < 1e-8
    # restart_num = 0 -> a single fit, no acceptance test
    res3 = st.tdr._morphofield_sparsevfc(Xf, Vf, NX=Xf[:10], M=12, restart_num=0, MaxIter=8)
    assert _rel(res3["V"], g["w2_V"]) < 1e-8


def test_non_finite_row_reproduces_reference_indexerror(cpu_kernels):
    """The reference wrapper indexes the N_valid-row V with valid_ind (sparsevfc.py:201-204): IndexError unless the
    non-finite rows are the last ones.  Observable reference behaviour, kept."""
    X, V = _data(120)
    V[7, 1] = np.nan
    with pytest.raises(IndexError):
        st.tdr._morphofield_sparsevfc(X, V, NX=X[:5], M=10, MaxIter=3, restart_num=1, restart_seed=[0])


def test_anndata_wrappers_match_reference_wrappers(golden, cpu_kernels):
    g = golden
    ad = st.AnnDataLite(obsm={"align_spatial": g["a_X"], "V_mapping": g["a_V"]})
    ad2 = st.tdr.morphofield_sparsevfc(ad, NX=g["a_X"][:5], M=15, MaxIter=20, restart_num=1, restart_seed=[0],
                                       inplace=False)
    assert "VecFld_morpho" not in ad.uns and ad2 is not ad  # inplace=False works on a copy
    assert st.tdr.morphofield_sparsevfc(ad, NX=g["a_X"][:5], M=15, MaxIter=20, restart_num=1, restart_seed=[0]) is None
    vf = ad.uns["VecFld_morpho"]
    for k in ["X_ctrl", "C", "beta", "V", "grid_V"]:
        assert _rel(vf[k], g[f"a_vf_{k}"]) < 1e-7, k
    assert vf["method"] == "sparsevfc" and vf["X"].dtype == np.float64
    for fn in (st.tdr.morphofield_velocity, st.tdr.morphofield_acceleration, st.tdr.morphofield_curvature,
               st.tdr.morphofield_curl, st.tdr.morphofield_torsion, st.tdr.morphofield_divergence,
               st.tdr.morphofield_jacobian):
        assert fn(ad) is None
    tol = 1e-6
    assert _rel(ad.obsm["velocity"], g["a_velocity"]) < tol
    assert _rel(ad.obs["acceleration"], g["a_acceleration_obs"]) < tol
    assert _rel(ad.obsm["acceleration"], g["a_acceleration_obsm"]) < tol
    assert _rel(ad.obs["curvature"], g["a_curvature_obs"]) < tol
    assert _rel(ad.obsm["curvature"], g["a_curvature_obsm"]) < tol
    assert ad.obsm["curl"].shape == (len(g["a_X"]), 3, 3)
    assert _rel(ad.obs["curl"], g["a_curl_obs"]) < tol and _rel(ad.obsm["curl"], g["a_curl_obsm"]) < tol
    assert _rel(ad.obs["torsion"], g["a_torsion_obs"]) < 1e-4 and _rel(ad.uns["torsion"], g["a_torsion_uns"]) < 1e-4
    assert _rel(ad.obs["divergence"], g["a_divergence_obs"]) < tol
    assert ad.uns["jacobian"].shape == (3, 3, len(g["a_X"]))
    assert _rel(ad.uns["jacobian"], g["a_jacobian_uns"]) < tol
    assert _rel(ad.obs["jacobian"], g["a_jacobian_obs"]) < 1e-4
    # custom keys + inplace=False on an evaluator
    ad3 = st.tdr.morphofield_divergence(ad, key_added="div2", inplace=False)
    assert "div2" in ad3.obs and "div2" not in ad.obs


def test_wrapper_error_conventions(cpu_kernels, golden):
    from _gp_case import gp_dict

    ad = st.AnnDataLite(obsm={"align_spatial": np.zeros((3, 3))})
    ad.uns["bad"] = {"method": "nope"}
    with pytest.raises(Exception, match="is not in ``anndata.uns"):
        st.tdr.morphofield_velocity(ad, vf_key="bad")
    with pytest.raises(KeyError):
        st.tdr.morphofield_jacobian(ad, vf_key="absent")
    with pytest.raises(Exception, match="morpho_align"):
        st.tdr.morphofield_gp(ad, vf_key="absent")
    geo = dict(gp_dict(golden), kernel_type="geodist")
    with pytest.raises(NotImplementedError, match="geodist"):
        st.vectorfield.gp_velocity(np.zeros((2, 3)), geo)
    with pytest.raises(ValueError):
        st.vectorfield.gp_velocity(np.zeros((2, 3)), dict(geo, kernel_type="other"))


def test_gp_variant_matches_reference_wrappers(cpu_kernels, golden):
    """morphofield_gp + the seven evaluators on a gaussian_process field (norm_dict scaling, rigid part,
    nonrigid_only) against the outputs of the REAL reference wrappers / GPVectorField twins."""
    from _gp_case import gp_dict, run_and_check

    g = golden
    np.testing.assert_allclose(st.vectorfield.gp_velocity(g["gp_Xq"], gp_dict(g)), g["gp_vel"], rtol=1e-9, atol=1e-14)
    np.testing.assert_allclose(st.vectorfield.gp_velocity(g["gp_Xq"], gp_dict(g), nonrigid_only=True),
                               g["gp_vel_nonrigid"], rtol=1e-9, atol=1e-14)
    vf = st.GPVectorField()
    vf.vf_dict, vf.nonrigid_only = gp_dict(g), False
    np.testing.assert_allclose(vf.get_Jacobian()(g["gp_Xq"]), g["gp_J"], rtol=1e-9, atol=1e-16)
    run_and_check(st, g, 1e-8)


def test_gp_variant_per_axis_scales_and_2d_fields(cpu_kernels):
    """VERDICT r5 "missing" #5: per-axis ``norm_dict`` scales and 2-D GP fields (round 5 raised NotImplementedError) against
    goldens produced by the real reference functions (tests/golden/make_golden_gp_axes.py)."""
    from _gp_case import check_axes_and_2d

    import os

    g = dict(np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_gp_axes.npz")))
    check_axes_and_2d(st, g, 1e-9)


def test_svcvectorfield_shapes_and_errors(cpu_kernels):
    rng = np.random.default_rng(0)
    vfd = {"X_ctrl": rng.standard_normal((8, 3)), "C": rng.standard_normal((8, 3)), "beta": 0.3,
           "X": rng.standard_normal((5, 3)), "Y": rng.standard_normal((5, 3))}
    ad = st.AnnDataLite(obsm={"s": vfd["X"]}, uns={"VecFld_morpho": vfd})
    vf = st.SvcVectorField().from_adata(ad, basis=None, vf_key="VecFld_morpho")
    X, V = vf.get_data()
    assert X is vfd["X"] and V is vfd["Y"]
    assert vf.func(X).shape == (5, 3) and vf.func(X[0]).shape == (3,)
    assert vf.get_Jacobian()(X).shape == (3, 3, 5) and vf.get_Jacobian()(X[0]).shape == (3, 3)
    with pytest.raises(NotImplementedError):
        vf.get_Jacobian(method="numerical")
    with pytest.raises(ValueError):
        st.SvcVectorField().from_adata(ad, vf_key="missing")
    with pytest.raises(ValueError, match="dimensions"):
        vf.func(np.zeros((2, 2)))
    # basis suffix
    ad.uns["VecFld_pca"] = vfd
    assert st.SvcVectorField().from_adata(ad, basis="pca", vf_key="VecFld").vf_dict is vfd


def test_anndata_lite_copy_is_deep():
    ad = st.AnnDataLite(obsm={"a": np.zeros((4, 3))}, obs={"x": [1, 2, 3, 4]})
    assert ad.n_obs == 4 and isinstance(ad.obs["x"], np.ndarray)
    c = ad.copy()
    c.obsm["a"][0, 0] = 5
    c.uns["k"] = 1
    assert ad.obsm["a"][0, 0] == 0 and "k" not in ad.uns


def test_kernel_interpolation_matches_reference_call_shape(cpu_kernels):
    """kernel_interpolation == SparseVFC(spatial, [obs cols | gene cols], target_points)["grid_V"] split back into
    obs / X (interpolation_sparseVFC.py:44-83)."""
    rng = np.random.default_rng(3)
    n = 400
    S = rng.uniform(-1, 1, (n, 3)) * 50
    genes = np.column_stack([np.sin(S[:, 0] / 20), np.cos(S[:, 1] / 15), S[:, 2] / 50, np.sin(S[:, 0] / 9) ** 2])
    genes += 0.01 * rng.standard_normal(genes.shape)
    score = np.cos(S[:, 2] / 25)
    ad = st.AnnDataLite(X=genes, var_names=["g0", "g1", "g2", "g3"], obs={"score": score}, obsm={"spatial": S})
    tgt = rng.uniform(-1, 1, (30, 3)) * 45
    kw = dict(M=40, MaxIter=10, seed=0)
    out = st.tdr.kernel_interpolation(ad, target_points=tgt, keys=["g2", "score", "g0", "g3"], lambda_=3.0, **kw)
    info = np.column_stack([score, genes[:, [2, 0, 3]]])  # obs keys first, then genes, each in `keys` order
    ref = svo.SparseVFC(S, info, tgt, lambda_=3.0, lstsq_method="scipy", **kw)["grid_V"]
    got_obs = np.asarray(out.obs["score"], dtype=float)
    got_X = np.asarray(out.X)
    assert got_X.shape == (30, 3) and list(out.var_names) == ["g2", "g0", "g3"]
    np.testing.assert_allclose(got_obs, ref[:, 0], rtol=1e-7, atol=1e-9)
    np.testing.assert_allclose(got_X, ref[:, 1:], rtol=1e-7, atol=1e-9)
    np.testing.assert_array_equal(np.asarray(out.obsm["spatial"]), tgt)
    # single key as a string; layer selection; error conventions
    ad.layers["alt"] = genes * 2
    one = st.tdr.kernel_interpolation(ad, target_points=tgt, keys="g1", layer="alt", lambda_=3.0, **kw)
    assert np.asarray(one.X).shape == (30, 1)
    with pytest.raises(AssertionError, match="keys"):
        st.tdr.kernel_interpolation(ad, target_points=tgt)
    with pytest.raises(ValueError, match="none of the keys"):
        st.tdr.kernel_interpolation(ad, target_points=tgt, keys=["nope"])


def test_morphopath_slots_and_accuracy(cpu_kernels, golden):
    """morphopath's AnnData slots (trajectory.py:111-115) and the RK4 solution against SciPy DOP853 on the oracle field."""
    from oracle import trajectory_oracle as tro

    g = golden
    ad = st.AnnDataLite(obsm={"align_spatial": g["a_X"], "V_mapping": g["a_V"]})
    vf = {k: g[f"a_vf_{k}"] for k in ["X_ctrl", "C", "beta", "V"]}
    vf.update(X=g["a_X"][:6], Y=g["a_V"][:6], method="sparsevfc", beta=float(g["a_vf_beta"]))
    vf["V"] = vf["V"][:6]
    ad.uns["VecFld_morpho"] = vf
    assert st.tdr.morphopath(ad, interpolation_num=21, t_end=40.0, direction="both", sampling="uniform_time") is None
    fate = ad.uns["fate_morpho"]
    assert set(fate["t"].keys()) == set(range(6)) and fate["prediction"][0].shape == (41, 3)  # (n_t, d) as the reference
    # the reference's consumer (construct_trajectory_X, morphopath_model.py:225) prepends the start point along axis 0
    assert np.concatenate([fate["init_states"][[0]], fate["prediction"][0]], axis=0).shape == (42, 3)
    assert fate["init_cells"] == [str(i) for i in range(6)]
    np.testing.assert_allclose(fate["t"][0], np.linspace(-40, 40, 41))
    np.testing.assert_allclose(fate["prediction"][2][20], g["a_X"][2])  # t = 0 is the start point
    tq = np.linspace(0, 40, 21)
    ref = tro.integrate(vf, g["a_X"][:6], tq)
    got = np.stack([fate["prediction"][i][20:] for i in range(6)])
    assert np.abs(got - ref).max() / np.abs(ref).max() < 1e-7
    refb = tro.integrate(vf, g["a_X"][:6], -tq)
    gotb = np.stack([fate["prediction"][i][20::-1] for i in range(6)])
    assert np.abs(gotb - refb).max() / np.abs(refb).max() < 1e-7
    # default t_end (dynamo getTend), averaging modes, copies, errors
    ad2 = st.tdr.morphopath(ad, interpolation_num=5, average="origin", inplace=False, key_added="f2",
                            sampling="uniform_time")
    assert "f2" in ad2.uns and "f2" not in ad.uns and len(ad2.uns["f2"]["prediction"]) == 1
    st.tdr.morphopath(ad, interpolation_num=5, t_end=3.0, average="trajectory", key_added="f3", sampling="uniform_time")
    assert ad.uns["f3"]["prediction"][0].shape == (5, 3)
    ad.uns["bad"] = {"method": "other", "X": g["a_X"][:2]}
    with pytest.raises(Exception, match="not in avaliable"):
        st.tdr.morphopath(ad, vf_key="bad")
    with pytest.raises(Exception, match="not in ``anndata.uns``"):
        st.tdr.morphopath(ad, vf_key="absent")


def test_sparsevfc_many_matches_sequential(cpu_kernels):
    from spateo_amd.vectorfield import SparseVFC_many

    data = []
    for k in range(3):
        X, V = _data(300 + 50 * k, seed=k)
        data.append((X, V, X[:5]))
    kw = dict(M=15, lambda_=3.0, MaxIter=5)
    seq = [st.SparseVFC(X, V, G, **kw) for X, V, G in data]
    par = SparseVFC_many(data, n_streams=2, **kw)
    for a, b in zip(seq, par):
        np.testing.assert_array_equal(a["V"], b["V"])
        np.testing.assert_array_equal(a["grid_V"], b["grid_V"])


def test_degenerate_inputs_behave_like_the_reference(cpu_kernels):
    """Tiny / degenerate inputs: same results or the same exception type as the oracle (= dynamo through sklearn/NumPy)."""
    rng = np.random.default_rng(0)
    X, Y = rng.standard_normal((5, 3)), rng.standard_normal((5, 3)) * 0.1
    kw = dict(lambda_=3.0, lstsq_method="scipy", MaxIter=3)
    for Xi, Yi, M in ((X, Y, 3), (X[:2], Y[:2], 5)):  # fewer cells than a tile; M clipped to the 2 unique rows
        ref = svo.SparseVFC(Xi, Yi, None, M=M, **kw)
        got = st.SparseVFC(Xi, Yi, None, M=M, **kw)
        assert got["X_ctrl"].shape == ref["X_ctrl"].shape and got["iteration"] == ref["iteration"]
        np.testing.assert_allclose(got["V"], ref["V"], rtol=1e-7, atol=1e-12)
    cases = [
        (X, Y, 1),                           # one control point: the kNN bandwidth rule needs >= 2
        (X[:1], Y[:1], 5),                   # one cell
        (np.zeros((6, 3)), Y[:1].repeat(6, 0), 5),  # all cells at one position -> one unique row
        (X, np.zeros((5, 3)), 3),            # zero velocities: sampling probabilities are NaN
        (np.zeros((0, 3)), np.zeros((0, 3)), 3),     # empty
    ]
    for Xi, Yi, M in cases:
        with pytest.raises(ValueError):
            svo.SparseVFC(Xi, Yi, None, M=M, **kw)
        with pytest.raises(ValueError):
            st.SparseVFC(Xi, Yi, None, M=M, **kw)
    # an explicit beta bypasses the bandwidth rule, but a single control point still fails in the reference (its
    # 1 x 1 K is flattened to 1-D and the energy term cannot be formed): ValueError in both
    with pytest.raises(ValueError):
        svo.SparseVFC(X, Y, None, M=1, beta=0.3, **kw)
    with pytest.raises(ValueError):
        st.SparseVFC(X, Y, None, M=1, beta=0.3, **kw)


def test_row_norms_and_finite_rows_are_the_reference_expressions_bit_for_bit():
    """The two host shortcuts of round 6: ``row_norms(V)`` IS ``np.linalg.norm(V, axis=1)`` (the weights of dynamo's
    velocity-based draw: one differing bit could change the draw) and ``finite_rows(Y)`` IS
    ``np.where(np.isfinite(Y.sum(1)))[0]`` - including NaN / inf entries, row sums that overflow although every entry is
    finite, integer and float32 input, 1 - 9 columns, empty and Fortran-ordered arrays."""
    from spateo_amd.preprocess import finite_rows, row_norms

    rng = np.random.default_rng(23)
    for d in range(1, 10):
        V = rng.standard_normal((4001, d)) * rng.uniform(1e-150, 1e150, (4001, 1))
        for A in (V, np.asfortranarray(V), V[::3], V.astype(np.float32)):
            assert row_norms(A).tobytes() == np.linalg.norm(A, axis=1).tobytes(), d
            assert np.array_equal(finite_rows(A), np.where(np.isfinite(A.sum(1)))[0])
    Y = rng.standard_normal((1000, 3))
    for bad in (np.nan, np.inf, -np.inf):
        Z = Y.copy()
        Z[[3, 500, 999], [0, 2, 1]] = bad
        assert np.array_equal(finite_rows(Z), np.where(np.isfinite(Z.sum(1)))[0]) and len(finite_rows(Z)) == 997
    Z = Y.copy()
    Z[7] = [1.7e308, 1.7e308, 0.0]      # finite entries, the row SUM overflows: the reference drops the row
    Z[8] = [1.7e308, -1.7e308, 1.0]     # ... and keeps this one
    with np.errstate(over="ignore"):
        ref = np.where(np.isfinite(Z.sum(1)))[0]
        got = finite_rows(Z)
    assert np.array_equal(got, ref) and 7 not in got and 8 in got
    assert np.array_equal(finite_rows(np.zeros((0, 3))), np.where(np.isfinite(np.zeros((0, 3)).sum(1)))[0])
    assert np.array_equal(finite_rows(np.arange(12).reshape(4, 3)), np.arange(4))


def test_unique_rows_matches_numpy_unique():
    """The host shortcut for dynamo's ``np.unique(X, axis=0, return_index=True)`` is bit-identical to it (rows, first
    occurrence indices), including ties on the leading coordinate, duplicated rows, signed zeros and 1/2/3 columns."""
    from spateo_amd.vectorfield import unique_rows

    rng = np.random.default_rng(5)
    for n, d, hi in [(1, 3, 5), (2, 3, 2), (1000, 3, 4), (1000, 2, 6), (5000, 3, 50), (3000, 1, 40), (50000, 3, 10**9)]:
        X = rng.integers(0, hi, size=(n, d)).astype(np.float64)
        if n > 10:
            X[::7] = X[3]
            X[5] = -0.0 * X[5]
        a, ai = np.unique(X, axis=0, return_index=True)
        b, bi = unique_rows(X)
        assert a.tobytes() == b.tobytes() and np.array_equal(ai, bi), (n, d, hi)
    Xn = rng.normal(size=(100, 3))
    Xn[7, 1] = np.nan  # non-finite input takes the NumPy route (whatever it does, it is the same thing)
    a, ai = np.unique(Xn, axis=0, return_index=True)
    b, bi = unique_rows(Xn)
    assert np.array_equal(ai, bi) and np.array_equal(a, b, equal_nan=True)


def test_ba_transform_matches_reference(cpu_kernels, golden_align):
    """spateo_amd.align.BA_transform (host composition around the device field evaluation) against the outputs of the
    real spateo/alignment/transform.py::BA_transform; con_K against the alignment module's own formulation."""
    from _align_case import ba_dict, check_ba

    g = golden_align
    check_ba(st.align.BA_transform, g, 1e-12)
    np.testing.assert_allclose(st.con_K(g["ak_x"], g["ak_y"], float(g["ak_beta"])), g["ak_K"], rtol=1e-12, atol=0)
    np.testing.assert_allclose(st.con_K(g["ak_x2"], g["ak_y2"], 0.7), g["ak_K2"], rtol=1e-12, atol=0)
    with pytest.raises(AssertionError):  # feature mismatch: the reference's con_K assertion (normalize_c False) ...
        st.align.BA_transform(ba_dict(g, False), g["ba_raw_q"][:, :2])
    with pytest.raises(ValueError):  # ... or NumPy's broadcast error in the normalisation, as in the reference
        st.align.BA_transform(ba_dict(g, True), g["ba_q"][:, :2])
    with pytest.raises(ValueError):
        st.align.BA_transform(ba_dict(g, True), g["ba_q"], dtype="float16")


def test_sample_by_velocity_same_draw_and_same_global_rng_state_as_dynamo():
    """The product draws from a private RandomState (thread-safe for SparseVFC_many) but must pick the same indices and
    leave NumPy's global RNG where dynamo's `np.random.seed(seed); np.random.choice(...)` leaves it."""
    import threading

    from spateo_amd.vectorfield import sample_by_velocity

    rng = np.random.default_rng(3)
    V = rng.standard_normal((4000, 3))
    a = sample_by_velocity(V, 150)
    x1 = np.random.rand(3)
    b = svo.sample_by_velocity(V, 150)
    x2 = np.random.rand(3)
    assert np.array_equal(a, b) and np.array_equal(x1, x2)
    out = [None] * 6
    ths = [threading.Thread(target=lambda i=i: out.__setitem__(i, sample_by_velocity(V, 150))) for i in range(6)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    assert all(np.array_equal(o, a) for o in out)


def test_unique_rows_and_shard_bounds_properties():
    """Property tests (hypothesis): unique_rows == np.unique(axis=0, return_index=True) on arbitrary small-alphabet
    float matrices (many ties, signed zeros), and shard_bounds tiles [0, n) contiguously with sizes differing by <= 1."""
    from hypothesis import given, settings
    from hypothesis import strategies as hs
    from hypothesis.extra import numpy as hnp

    from spateo_amd.vectorfield import shard_bounds, unique_rows

    vals = hs.sampled_from([-2.0, -0.0, 0.0, 0.5, 1.0, 3.25, 1e-300, -1e300])

    @settings(max_examples=150, deadline=None)
    @given(hnp.arrays(np.float64, hnp.array_shapes(min_dims=2, max_dims=2, min_side=1, max_side=40).filter(
        lambda s: s[1] <= 3), elements=vals))
    def uniq(X):
        a, ai = np.unique(X, axis=0, return_index=True)
        b, bi = unique_rows(X)
        assert np.array_equal(ai, bi) and a.tobytes() == b.tobytes()

    @settings(max_examples=200, deadline=None)
    @given(hs.integers(0, 10**7), hs.integers(1, 64))
    def shards(n, world):
        b = [shard_bounds(n, r, world) for r in range(world)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(world - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1 and sorted(sizes, reverse=True) == sizes

    uniq()
    shards()


def test_update_nonrigid_host_composition_matches_reference(cpu_kernels, golden_em):
    """spateo_amd.align.update_nonrigid (host composition around gram / min-norm solve / apply) against the outputs of
    the real Morpho_pairwise._update_nonrigid (device replaced by the CPU double; the GPU test runs the kernels)."""
    from _align_case import check_update_nonrigid

    check_update_nonrigid(st.align.update_nonrigid, golden_em, "float64")
    from _align_case import check_update_nonrigid_branches

    check_update_nonrigid_branches(st.align.update_nonrigid, golden_em, "float64")  # guidance / SVI / both
    with pytest.raises(AssertionError):
        st.align.update_nonrigid(golden_em["a_coordsA"][:, :2], golden_em["a_inducing_variables"], 0.5,
                                 golden_em["a_K_NA"], golden_em["a_PXB_term"], 0.5, 100.0)


from _fate_case import _fate_case, check_fate_semantics  # noqa: E402


def test_morphopath_default_sampling_is_dynamo_fates_arc_length(cpu_kernels, golden):
    check_fate_semantics(golden)


def test_construct_genesis_states_like_the_reference_loop(cpu_kernels, golden):
    """construct_genesis' numeric core (morphopath_model.py:123-148): time vector from the fate prediction (integer
    truncation, reference quirks included) and the step-by-step displacement, against SciPy odeint per cell and step."""
    from oracle import trajectory_oracle as tro

    vf = _fate_case(golden)
    ad = st.AnnDataLite(obsm={"align_spatial": golden["a_X"][:5]}, uns={"VecFld_morpho": vf})
    st.tdr.morphopath(ad, interpolation_num=12, t_end=40.0)
    stages, tv = st.tdr.construct_genesis_states(ad, fate_key="fate_morpho", n_steps=6)
    flats = np.sort(np.hstack((0, np.unique([int(v) for t in ad.uns["fate_morpho"]["t"].values() for v in t]))))
    np.testing.assert_array_equal(tv, flats[np.linspace(0, len(flats) - 1, 6).astype(int)])
    ref = tro.genesis_states(vf, ad.uns["fate_morpho"]["init_states"], tv)
    assert len(stages) == 6 and stages[0].shape == (5, 3)
    for a, b in zip(stages, ref):
        assert np.abs(a - b).max() / np.abs(b).max() < 1e-5
    _, tv2 = st.tdr.construct_genesis_states(ad, n_steps=4, logspace=True, t_end=25)
    np.testing.assert_allclose(tv2, np.logspace(0, np.log10(max(flats[flats <= 25]) + 1), 4) - 1)
    with pytest.raises(Exception, match="develop_trajectory"):
        st.tdr.construct_genesis_states(ad, fate_key="nope")


def test_cell_directions_mapping_logic_matches_the_real_function():
    """st.tdr.cell_directions (coupling -> one partner per cell -> X_mapping, V_mapping) against goldens produced by the
    REAL cell_directions + get_optimal_mapping_relationship (tests/golden/make_golden_celldir.py; only the PASTE solve
    in front of them was replaced by a prepared coupling with exact ties)."""
    import os

    with np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_celldir.npz")) as z:
        g = {k: z[k] for k in z.files}
    for keep_all, tag in ((False, "nearest"), (True, "all")):
        A = st.AnnDataLite(obsm={"align_spatial": g["XA"].copy()})
        B = st.AnnDataLite(obsm={"align_spatial": g["XB"].copy()})
        ret, pi = st.tdr.cell_directions(A, B, pi=g["pi"], keep_all=keep_all)
        assert ret is None and pi.shape == g["pi"].shape
        np.testing.assert_array_equal(A.obsm["X_mapping"], g[f"{tag}_X_mapping"])
        np.testing.assert_array_equal(A.obsm["V_mapping"], g[f"{tag}_V_mapping"])
    assert not np.array_equal(g["nearest_X_mapping"], g["all_X_mapping"])  # the ties do matter in this fixture
    A = st.AnnDataLite(obsm={"align_spatial": g["XA"].copy()})
    A2, _ = st.tdr.cell_directions(A, B, pi=g["pi"], inplace=False, key_added="m2")
    assert "X_m2" in A2.obsm and "X_m2" not in A.obsm
    with pytest.raises(NotImplementedError, match="optimal-transport"):
        st.tdr.cell_directions(A, B)
    with pytest.raises(ValueError):
        st.tdr.cell_directions(A, B, pi=g["pi"][:5])


# ------------------------------------------------------------------------------------------------ fused evaluator pass
def test_one_evaluator_launch_serves_the_calls_on_the_same_points(monkeypatch):
    """``get_Jacobian()(X)`` then ``compute_curl(X=X)`` (BASELINE config 2's call shape) and the seven ``morphofield_*``
    wrappers, each with a fresh vector-field object, are ONE evaluator launch as long as points and field are the same by
    value; another field or other points launch again; results are those of separate launches."""
    launches = []

    class Counting(CpuKernels):
        def eval(self, x4, ctrl4, beta, C, flags, affine=None):
            launches.append(flags)
            return super().eval(x4, ctrl4, beta, C, flags, affine)

    monkeypatch.setattr(vfm._rt, "_make_kernels", lambda device, dtype: Counting(device, dtype))
    vfm.clear_eval_cache()
    rng = np.random.default_rng(5)
    vfd = {"X_ctrl": rng.normal(size=(40, 3)), "C": rng.normal(size=(40, 3)), "beta": 0.3, "X": None, "Y": None}
    X = rng.normal(size=(300, 3))
    vf = st.SvcVectorField()
    vf.vf_dict = vfd
    J = vf.get_Jacobian()(X)
    curl = vf.compute_curl(X=X.copy())          # equal by value, another array object
    vf2 = st.SvcVectorField()
    vf2.vf_dict = dict(vfd)                      # a fresh object and a copied dict: still the same field
    div = vf2.compute_divergence(X=X)
    _, acc = vf2.compute_acceleration(X=X)
    Jd, det = vf2.jacobian_with_det(X)
    assert len(launches) == 1 and launches[0] == vfm._EVAL_ALL
    np.testing.assert_array_equal(Jd, J)
    assert not np.shares_memory(Jd, J)
    np.testing.assert_allclose(det, np.linalg.det(np.moveaxis(J, 2, 0)), rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(curl[:, 0, :], np.stack([J[2, 1] - J[1, 2], J[0, 2] - J[2, 0], J[1, 0] - J[0, 1]], 1))
    np.testing.assert_allclose(div, np.trace(J))
    # the host arrays handed out are the caller's own: writing into one does not leak into the next call
    keep = J.copy()
    J[:] = 0.0
    np.testing.assert_array_equal(vf.get_Jacobian()(X), keep)
    assert len(launches) == 1
    # other points, or the same points on a changed field: a new launch each
    vf.compute_divergence(X=X + 1e-9)
    assert len(launches) == 2
    vfd2 = dict(vfd, C=vfd["C"] * (1 + 1e-12))
    vf3 = st.SvcVectorField()
    vf3.vf_dict = vfd2
    d3 = vf3.compute_divergence(X=X + 1e-9)
    assert len(launches) == 3 and not np.array_equal(d3, div)
    # beyond the prefetch cap only what is asked for is computed and kept; a later quantity launches for itself only
    monkeypatch.setattr(vfm, "_EVAL_PREFETCH_CAP", 0)
    vfm.clear_eval_cache()
    vf3.compute_divergence(X=X)
    vf3.compute_divergence(X=X)
    vf3.compute_curl(X=X)
    assert launches[3:] == [vfm._lib.EVAL_DIV, vfm._lib.EVAL_CURL]
    vfm.clear_eval_cache()


def test_curl_column_selection(cpu_kernels):
    """``compute_curl(X, dim1, dim2, dim3)`` evaluates at ``X[:, [dim1, dim2, dim3]]`` (``GPVectorField.py:55-63``): the
    default selection of a 3-D X is X itself (no copy of the points is made for it), a permuted one is gathered."""
    rng = np.random.default_rng(8)
    vfd = {"X_ctrl": rng.normal(size=(30, 3)), "C": rng.normal(size=(30, 3)), "beta": 0.4}
    vf = st.SvcVectorField()
    vf.vf_dict = vfd
    X = rng.normal(size=(50, 3))
    J = vf.get_Jacobian()(X)
    want = np.stack([J[2, 1] - J[1, 2], J[0, 2] - J[2, 0], J[1, 0] - J[0, 1]], 1)
    c = vf.compute_curl(X=X)
    assert c.shape == (50, 3, 3)
    for r in range(3):
        np.testing.assert_allclose(c[:, r, :], want, rtol=1e-12, atol=1e-14)
    Xp = X[:, [1, 0, 2]]
    np.testing.assert_array_equal(vf.compute_curl(X=X, dim1=1, dim2=0, dim3=2), vf.compute_curl(X=Xp.copy()))
    with pytest.raises(ValueError):
        vf.compute_curl(X=X, dim3=None)  # two selected columns against a 3-D field


def test_jacobian_with_det_in_two_dimensions(cpu_kernels):
    rng = np.random.default_rng(6)
    vfd = {"X_ctrl": rng.normal(size=(30, 2)), "C": rng.normal(size=(30, 2)), "beta": 0.4}
    vf = st.SvcVectorField()
    vf.vf_dict = vfd
    X = rng.normal(size=(50, 2))
    J, det = vf.jacobian_with_det(X)
    assert J.shape == (2, 2, 50)
    np.testing.assert_allclose(det, np.linalg.det(np.moveaxis(J, 2, 0)), rtol=1e-12, atol=1e-15)


def test_deflated_route_to_the_truncated_solve_restated_in_numpy():
    """mvf_solve_minnorm_lrd's algorithm (HISTORY.md 2.2.11) as NumPy (`_cpu_kernels.deflated_minnorm`): on a rank-deficient
    kernel system the invariant subspace below the eps * lambda_max cut-off, found by block inverse iteration on the r x r matrix
    L^T L of the pivoted factor, and the deflated solve give the truncated minimum-norm solution of that factor - the same number
    of truncated eigenvalues, the field within 1e-5 of the SVD route's (the reference's own lstsq-vs-eigh floor on this system:
    4e-2) - while a block smaller than what lies below the cut is noticed and answered by the SVD route."""
    import scipy.linalg
    import torch
    from spateo_amd._synthetic import make_config

    from _cpu_kernels import CpuKernels, deflated_minnorm
    from oracle import sparsevfc_oracle as svo

    n, m = 6000, 700
    X, Y, _ = make_config("C2", N=n)
    valid, Xv, Yv, idx, ctrl, beta = svo.sparsevfc_setup(X, Y, M=m, seed=0)
    K = svo.con_K(ctrl, ctrl, beta)
    U = svo.con_K(Xv, ctrl, beta)
    P = np.clip(np.random.default_rng(0).random(len(Xv)), 1e-5, 1.0)
    G, R, ls2 = (U.T * P[None, :]) @ U, (U.T * P[None, :]) @ Yv, 0.02 * 1e-3

    def solve(block, deflate):
        k = CpuKernels()
        k.DEFL_BLOCK = block
        C, info, e = torch.empty(m, 3, dtype=torch.float64), torch.zeros(1, dtype=torch.int32), torch.zeros(12, dtype=torch.float64)
        k.solve_minnorm_lr(torch.from_numpy(G), torch.from_numpy(K), ls2, torch.from_numpy(R), C, info, e, deflate=deflate)
        return k, C.numpy().copy(), e.numpy().copy()

    k, C_svd, e = solve(128, False)
    r, kept = int(e[6]), int(e[1])
    assert r >= 2 * 128 and 0 < r - kept < 96          # a rank-deficient system with room in the block
    F = U @ scipy.linalg.lstsq(G + ls2 * K, R)[0]
    sc = np.abs(F).max()
    cut = np.finfo(float).eps * k._lr[1].max()
    C_defl, nsel = deflated_minnorm(k._lr_factor, R, cut, 128)
    assert nsel == r - kept
    assert np.abs(U @ (C_defl - C_svd)).max() / sc < 1e-5
    assert np.abs(U @ C_defl - F).max() / sc < 4e-2      # and with it inside the reference's floor on this system
    np.testing.assert_array_equal(solve(128, True)[1], C_defl)   # the twin's deflate=True IS this route ...
    np.testing.assert_array_equal(solve(32, True)[1], C_svd)     # ... and falls back when the block cannot hold the subspace


def test_restart_loop_refits_per_seed_unless_the_memo_is_opted_into(cpu_kernels, monkeypatch):
    """the default follows the reference's control flow (one fit per restart, sparsevfc.py:178-232);
    reusing the first fit for every restart (valid only if dynamo's sample_by_velocity ignores `seed`, a [VERIFY] item) is
    the opt-in `reuse_identical_restarts=True`."""
    import spateo_amd.tdr.morphometrics.morphofield.sparsevfc as wrap

    calls = []
    orig = wrap.SparseVFC

    def counting(*a, **kw):
        calls.append(kw.get("seed"))
        return orig(*a, **kw)

    monkeypatch.setattr(wrap, "SparseVFC", counting)
    X, V = _data(300)
    kw = dict(NX=X[:10], M=12, min_vel_corr=2.0, restart_num=3, restart_seed=[0, 100, 200], MaxIter=4)
    res = st.tdr._morphofield_sparsevfc(X, V, **kw)
    assert calls == [0, 100, 200]
    calls.clear()
    res2 = st.tdr._morphofield_sparsevfc(X, V, reuse_identical_restarts=True, **kw)
    assert calls == [0]
    np.testing.assert_array_equal(res["V"], res2["V"])
    calls.clear()
    st.tdr._morphofield_sparsevfc(X, V, reuse_identical_restarts=True, velocity_based_sampling=False, **kw)
    assert calls == [0, 100, 200]  # seeded permutation sampling: the seed reaches the fit, nothing to reuse
