"""GPU parity at the sizes the headline numbers live at (VERDICT r1 "prove parity where the headline lives").

* BASELINE config 2 AT ITS STATED SIZE (50 k cells x 500 control points, Jacobian + curl on the 64^3 grid), float64 and
  float32, against the float64 NumPy oracle;
* M = 2000 and M = 3000 control points (the 24 x 24-tile Gram plan, the 47-panel Cholesky, the eigensolver) with
  N = 20 k cells: single EM step and a 10-step fit, lambda_ = 3 and Spateo's default 0.02;
* one BASELINE config 5 organ at its size (250 k cells x 500);
* the benchmark's own sizes against the STREAMED oracle's fixtures (round 4): one EM iteration at 8 M x 3000, five at
  1 M x 3000 (one rank's share) and at 2 M x 2000 (BASELINE config 3 at its stated size), every quantity at 1.25 x its floor;
* float32 mode vs float64 mode at 1 M x 3000 (supplement).

Tolerances (BASELINE.json north_star): field within 1e-5 relative in float64 mode, 1e-3 in float32 mode, wherever the
reference's own solve is stable.  It rarely is at these sizes: with the 20 %-nearest-neighbour bandwidth rule the
Gaussian Gram system is numerically rank deficient for EVERY lambda_ once M is in the thousands (measured: rank
1943 / 3000 at lambda_ = 3 in the very first EM step), and at M = 500 from the second or third step on.  There the
reference's own result moves under changes that leave its mathematics untouched - its LAPACK driver swapped for an
identical one (gelsd -> truncated eigh, same eps cut-off), its Gram product summed over the cells in another order
(what a different BLAS thread count does): tests/_floors.py measures both on the oracle for every case and quantity
(field on the cells, grid field inside the data hull and over the whole bounding box, sigma^2, P, energy) and the GPU
result must sit within 1.25x of the larger one or inside the mode's tolerance, whichever is larger.  No assertion is
conditional: a differing iteration count is a failure.  Float32 mode additionally carries the effect of the data type
itself: the same oracle fed with kernel values computed in float32 arithmetic ("f32kernel" floor).
"""
import functools
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dg_oracle as dgo  # noqa: E402
from oracle import sparsevfc_oracle as svo  # noqa: E402

import _floors as F  # noqa: E402

TOL = {"float64": 1e-5, "float32": 1e-3}
# sigma^2, the energy and (where the field floor allows it) P are well determined even when C is not: held to 1e-4 in both
# modes wherever the reference's own floor for them is below that
TIGHT = 1e-4


@pytest.fixture(scope="module")
def st():
    import spateo_amd

    assert torch.cuda.is_available()
    return spateo_amd


_rel = F.rel
_eigh_solver = F.eigh_solver


def _tol(dtype, floors):
    """floors = (float64-mode floor, float32-mode floor) of the field on the cells."""
    return max(F.ALLOW * floors[0 if dtype == "float64" else 1], TOL[dtype])


# The float64 oracle needs minutes of host time for the M = 2000 / 3000 cases (three 10-step fits each, every step a
# 3000 x 3000 lstsq).  Its outputs for exactly these seeded cases are committed in tests/golden/scale_oracle.npz
# (written by tests/golden/make_scale_oracle.py, which calls the very functions below); delete the file or set
# MVF_SCALE_ORACLE_LIVE=1 to recompute them inside the test run.
_ORACLE_NPZ = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "scale_oracle.npz")
_ORACLE_STORE = {}


def _oracle_cached(key, compute):
    """dict of arrays for `key`: from the committed fixture when present, else computed now (and kept for the writer)."""
    if not _ORACLE_STORE and os.path.exists(_ORACLE_NPZ) and os.environ.get("MVF_SCALE_ORACLE_LIVE") != "1":
        with np.load(_ORACLE_NPZ) as z:
            for name in z.files:
                k, field = name.split("|")
                _ORACLE_STORE.setdefault(k, {})[field] = z[name]
    if key not in _ORACLE_STORE:
        _ORACLE_STORE[key] = {f: np.asarray(v) for f, v in compute().items()}
    return _ORACLE_STORE[key]


# ------------------------------------------------------------------------------------------- BASELINE config 2
@functools.lru_cache(maxsize=None)
def _c2_case(lambda_):
    from spateo_amd._synthetic import make_config
    from spateo_amd.tdr.interpolations.utils import get_X_Y_grid

    X, V, M = make_config("C2")
    assert X.shape == (50_000, 3) and M == 500
    _, _, Grid, in_hull = get_X_Y_grid(X=X, Y=V, grid_num=[64, 64, 64])
    assert Grid.shape == (64**3, 3)
    kw = dict(M=M, lambda_=lambda_, lstsq_method="scipy", seed=0)
    ref = F.oracle_fit(X, V, Grid, **kw)
    near = F.near_mask(X, Grid)
    table = F.floor_table(X, V, Grid, ref, kw, in_hull=in_hull, near=near)
    return X, V, Grid, kw, ref, table, (in_hull, near)


def _base_tolerances(dtype, tight=TIGHT):
    """Base (floor-independent) tolerances: the north-star field tolerance of the mode; sigma^2 and the energy at 1e-4 in
    float64 mode (they are well determined even where C is not) and at the mode's 1e-3 in float32 mode (cell records,
    residuals and P are float32 there: measured 1.3e-4 / 5.8e-4 at C2, lambda_ = 3); P at 10 x the field tolerance."""
    se = tight if dtype == "float64" else TOL[dtype]
    return {"V": TOL[dtype], "grid": TOL[dtype], "grid12": TOL[dtype], "hull": TOL[dtype], "sigma2": se, "E": se,
            "P": 10 * TOL[dtype], "P999": 10 * TOL[dtype]}


# Two statistics are held to a LOOSE bound instead of the 1.25 x of every other quantity, because a maximum over a heavy tail
# cannot be held to a tight multiple of another draw of the same tail (rounds 3 - 4 asserted them with constants fitted to the
# measurement, 1.75 x / 1.85 x; round 5 only printed them - ADVICE r5: a localised regression must still
# fail): "P" = max over the cells of |dP| (set by the single worst cell at the inlier / outlier boundary) and "grid" = the
# whole bounding-box grid (corners 1.75 hull radii out, pure extrapolation) are capped at F.HARD_CAP = 3 x their own floor.
# What carries the 1.25 x in their place: "P999" = the 99.9th percentile of |dP| and "grid12" = the grid within 1.2 hull
# radii (tests/_floors.py: P_QUANTILE, NEAR_RADIUS - definitions, not fits).
LOOSE = {"P": "P999", "grid": "grid12"}


def _limits(dtype, table, dev, base):
    """max(1.25 x floor, base tolerance) for every quantity; max(3 x floor, base tolerance) for the two heavy-tailed ones whose
    tight counterpart is present."""
    lim = {}
    for k in dev:
        loose = k in LOOSE and LOOSE[k] in dev
        lim[k] = F.cap(dtype, table, k, base[k]) if loose else F.tol(dtype, table, k, base[k])
    return lim


def _report(tag, dtype, dev, fl, lim, extra=""):
    print(f"{tag} {dtype}: {extra}" + "; ".join(
        f"{k} gpu {dev[k]:.2e} / floor {fl[k]:.2e} (x{dev[k] / max(fl[k], 1e-300):.2f}, limit {lim[k]:.2e})" for k in dev))


def _check_fit(tag, dtype, got, ref, table, masks=None, tight=TIGHT):
    """Every quantity of a whole fit against the oracle, each within max(1.25 x its own reference floor, its base
    tolerance): the field (cells; grid inside the hull; grid within 1.2 hull radii) at the mode's tolerance, sigma^2 /
    energy at min(mode tolerance, ...) >= `tight`, the 99.9th percentile of |dP| at 10 x the mode's tolerance; max |dP| and
    the whole bounding-box grid within max(3 x their floor, their base tolerance).  Nothing is conditional."""
    assert got["iteration"] == ref["iteration"], (got["iteration"], ref["iteration"])
    in_hull, near = masks if masks is not None else (None, None)
    dev = F.deviations(got, ref, in_hull, near)
    base = _base_tolerances(dtype, tight)
    lim = _limits(dtype, table, dev, base)
    fl = {k: table[k][0 if dtype == "float64" else 1] for k in dev}
    _report(tag, dtype, dev, fl, lim, f"iterations {got['iteration'] + 1}; ")
    print(F.fmt(table))
    bad = {k: (dev[k], lim[k]) for k in dev if not dev[k] <= lim[k]}
    assert not bad, bad
    return dev


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_c2_full_size_fit_well_regularised(st, dtype):
    """50 k x 500, run to convergence, lambda_ = 3: the whole dict against the oracle."""
    X, V, Grid, kw, ref, table, masks = _c2_case(3.0)
    got = st.SparseVFC(X, V, Grid, dtype=dtype, device="cuda:0", **kw)
    np.testing.assert_array_equal(got["ctrl_idx"], ref["ctrl_idx"])
    _check_fit("C2 lambda 3", dtype, got, ref, table, masks)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_c2_full_size_fit_default_lambda(st, dtype):
    """Spateo's default lambda_ = 0.02 at the stated size, run to convergence: field on the cells, grid field inside the
    hull, grid field over the whole bounding box (extrapolation through the ill-determined part of C), sigma^2, P and
    the energy are each reported and asserted against their own floor."""
    X, V, Grid, kw, ref, table, masks = _c2_case(0.02)
    got = st.SparseVFC(X, V, Grid, dtype=dtype, device="cuda:0", **kw)
    _check_fit("C2 lambda 0.02", dtype, got, ref, table, masks)


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-9), ("float32", 1e-3)])
def test_c2_jacobian_and_curl_on_the_64_cubed_grid(st, dtype, tol):
    """Jacobian, curl, divergence on all 262 144 grid points against the oracle's analytical formulas (chunked), on the
    ORACLE's coefficients (evaluator parity only; float32 mode = float32 kernel values x the coefficients' large
    cancelling entries: 4e-4 measured, tolerance = the mode's 1e-3)."""
    X, V, Grid, kw, ref = _c2_case(3.0)[:5]
    vf = st.SvcVectorField(dtype=dtype, device="cuda:0")
    vf.vf_dict = ref
    J = vf.get_Jacobian()(Grid)
    curl = vf.compute_curl(X=Grid)
    div = vf.compute_divergence(X=Grid)
    assert J.shape == (3, 3, len(Grid)) and curl.shape == (len(Grid), 3, 3) and div.shape == (len(Grid),)
    jmax, worst_j, worst_c, worst_d = 0.0, 0.0, 0.0, 0.0
    for lo in range(0, len(Grid), 32768):
        sl = slice(lo, lo + 32768)
        Jr = dgo.Jacobian_rkhs_gaussian(Grid[sl], ref, vectorize=True)
        cr = np.stack([Jr[2, 1] - Jr[1, 2], Jr[0, 2] - Jr[2, 0], Jr[1, 0] - Jr[0, 1]], axis=1)
        jmax = max(jmax, np.abs(Jr).max())
        worst_j = max(worst_j, np.abs(J[:, :, sl] - Jr).max())
        worst_c = max(worst_c, np.abs(curl[sl, 0, :] - cr).max())
        worst_d = max(worst_d, np.abs(div[sl] - np.trace(Jr)).max())
    assert worst_j / jmax < tol and worst_c / jmax < tol and worst_d / jmax < tol


# ------------------------------------------------------------------------------------------- M = 2000 / 3000
_KEEP = ("V", "P", "sigma2", "E_traj", "iteration")


def _fixture_fit(key, X, V, kw, stride=1):
    """Oracle fit + floor table of a seeded case, through the committed fixture.  Stored: the reference's V / P (every
    `stride`-th cell), sigma2, E_traj, iteration and the floor table (computed on ALL cells)."""
    def compute():
        ref = F.oracle_fit(X, V, None, **kw)
        table = F.floor_table(X, V, None, ref, kw)
        out = dict(V=ref["V"][::stride], P=ref["P"][::stride], sigma2=ref["sigma2"], E_traj=ref["E_traj"],
                   iteration=ref["iteration"], vmax=np.abs(ref["V"]).max())
        for q, val in table.items():
            if q != "_variants":
                out[f"floor_{q}"] = np.array(val)
        for v, d in table["_variants"].items():
            for q, val in (d or {}).items():
                out[f"var_{v}_{q}"] = val
        return out

    c = _oracle_cached(key, compute)
    ref = dict(V=c["V"], P=c["P"], sigma2=float(c["sigma2"]), E_traj=c["E_traj"], iteration=int(c["iteration"]),
               vmax=float(c["vmax"]))
    table = {q[len("floor_"):]: (float(v[0]), float(v[1])) for q, v in c.items() if q.startswith("floor_")}
    per = {}
    for name, val in c.items():
        if name.startswith("var_"):
            _, v, q = name.split("_", 2)
            per.setdefault(v, {})[q] = float(val)
    table["_variants"] = per
    return ref, table


def _check_fixture_fit(tag, dtype, got, ref, table, stride=1, tight=TIGHT):
    """As _check_fit, against a (possibly strided) stored reference; the field error is normalised by the reference's
    max |V| over ALL cells."""
    assert got["iteration"] == ref["iteration"], (got["iteration"], ref["iteration"])
    dP = np.abs(got["P"][::stride] - ref["P"])
    dev = {"V": float(np.abs(got["V"][::stride] - ref["V"]).max() / ref["vmax"]),
           "sigma2": abs(got["sigma2"] - ref["sigma2"]) / ref["sigma2"],
           "P": float(dP.max()),
           "P999": F.p_quantile(dP),
           "E": float(np.abs((got["E_traj"] - ref["E_traj"]) / ref["E_traj"]).max())}
    assert "P999" in table, "tests/golden/scale_oracle.npz predates the P999 floor: rerun tests/golden/make_scale_oracle.py"
    base = _base_tolerances(dtype, tight)
    lim = _limits(dtype, table, dev, base)
    fl = {k: table[k][0 if dtype == "float64" else 1] for k in dev}
    _report(tag, dtype, dev, fl, lim)
    print(F.fmt(table))
    bad = {k: (dev[k], lim[k]) for k in dev if not dev[k] <= lim[k]}
    assert not bad, bad
    return dev


@functools.lru_cache(maxsize=None)
def _large_m_case(M, lambda_, n=20_000, steps=10):
    from spateo_amd._synthetic import make_config

    X, V, _ = make_config("C3", N=n)
    kw = dict(M=M, lambda_=lambda_, lstsq_method="scipy", MaxIter=steps, ecr=0.0, seed=0)
    ref, table = _fixture_fit(f"fit_M{M}_lam{lambda_}_n{n}_s{steps}", X, V, kw)
    return X, V, kw, ref, table


def _single_step_case(M, lambda_):
    """Oracle side of the single-EM-step test: (Xv, Yv, ctrl, beta, oracle outputs + floors)."""
    from spateo_amd._synthetic import make_config
    from spateo_amd.vectorfield import sparsevfc_preprocess

    X, V, _ = make_config("C3", N=20_000)
    valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=M, seed=0)
    N, D = Yv.shape
    s2 = np.sum(Yv**2) / (N * D)

    def compute():
        K = svo.con_K(ctrl, ctrl, beta)
        U = svo.con_K(Xv, ctrl, beta)
        Pr, Er, tecr_r, Cr, Vr, s2r, gr = svo.em_step(
            U, K, Yv, np.zeros_like(Yv), np.zeros((M, D)), s2, 0.9, 1, a=5, lambda_=lambda_, minP=1e-5, theta=0.75,
            lstsq_method="scipy")
        # the reference's own floors for this one step: same state, LAPACK driver swapped / Gram summed in another
        # order / float32 kernel values
        UP = U.T * np.maximum(Pr, 1e-5).T
        lhs, rhs = UP @ U + lambda_ * s2 * K, UP @ Yv
        f_eigh = _rel(U @ _eigh_solver(lhs, rhs), Vr)
        f_sum = _rel(U @ svo.lstsq_solver(F.chunked_dot(UP, U) + lambda_ * s2 * K, F.chunked_dot(UP, Yv), "scipy"), Vr)
        U32, K32 = U.astype(np.float32).astype(np.float64), K.astype(np.float32).astype(np.float64)
        UP32 = U32.T * np.maximum(Pr, 1e-5).T
        f32 = _rel(U32 @ svo.lstsq_solver(UP32 @ U32 + lambda_ * s2 * K32, UP32 @ Yv, "scipy"), Vr)
        return dict(Vr=Vr, Pr=Pr, Er=Er, s2r=s2r, f_eigh=f_eigh, f_sum=f_sum, f64=max(f_eigh, f_sum), f32=f32)

    return Xv, Yv, ctrl, beta, _oracle_cached(f"step_M{M}_lam{lambda_}", compute)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("M", [2000, 3000])
def test_large_m_single_em_step(st, dtype, M):
    """One EM iteration from the identical state (V = 0) at M = 2000 / 3000 - multi-tile Gram plan, 32 / 47-panel
    Cholesky, the eigensolver (the system is rank deficient from the first step) - for both lambdas."""
    from spateo_amd.vectorfield import SparseVFCEngine

    for lambda_ in (3.0, 0.02):
        Xv, Yv, ctrl, beta, c = _single_step_case(M, lambda_)
        Vr, Pr, Er, s2r, f64, f32 = c["Vr"], c["Pr"], float(c["Er"]), float(c["s2r"]), float(c["f64"]), float(c["f32"])
        eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype=dtype, device="cuda:0")
        eng.init_state(gamma=0.9)
        E, tecr = eng.em_step(a=5, lambda_=lambda_, minP=1e-5, theta=0.75)
        Vg, Pg, Cg = eng.results()
        tol = _tol(dtype, (f64, max(f64, f32)))
        err = _rel(Vg, Vr)
        print(f"M={M} {dtype} lambda={lambda_}: V err {err:.2e} (reference floors: eigh {float(c['f_eigh']):.2e}, "
              f"sum order {float(c['f_sum']):.2e}, f32 kernel {f32:.2e}; limit {tol:.2e}), sigma2 rel "
              f"{abs(eng.sigma2 - s2r) / s2r:.2e}, solver {eng.solver_stats}")
        assert err < tol
        np.testing.assert_allclose(eng.sigma2, s2r, rtol=TIGHT)  # measured <= 1.2e-5 in either mode
        np.testing.assert_allclose(Pg, Pr, rtol=TOL[dtype], atol=1e-9)  # P and E precede the solve: the mode's tolerance
        np.testing.assert_allclose(E, Er, rtol=TOL[dtype])
        del eng


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("M", [2000, 3000])
def test_large_m_ten_step_fit_well_regularised(st, dtype, M):
    X, V, kw, ref, table = _large_m_case(M, 3.0)
    got = st.SparseVFC(X, V, None, dtype=dtype, device="cuda:0", **kw)
    assert got["iteration"] == 9
    _check_fixture_fit(f"M={M} lambda=3", dtype, got, ref, table)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("M", [2000, 3000])
def test_large_m_ten_step_fit_default_lambda(st, dtype, M):
    """lambda_ = 0.02, M = 2000 / 3000: numerically rank deficient from the first iterations - the regime of the bench."""
    X, V, kw, ref, table = _large_m_case(M, 0.02)
    got = st.SparseVFC(X, V, None, dtype=dtype, device="cuda:0", **kw)
    assert got["iteration"] == 9
    _check_fixture_fit(f"M={M} lambda=0.02", dtype, got, ref, table)


# ------------------------------------------------------------------------------------------- the CPU baseline's size
_C4_SAMPLE = dict(n=200_000, M=3000, lambda_=0.02, steps=10, stride=8)


@functools.lru_cache(maxsize=None)
def _c4_sample_case():
    """BASELINE.md section 3's N_cpu: the bench's own generator (C4) at 200 k cells x 3000 control points, lambda_ = 0.02,
    10 EM iterations - the largest size the reference form (U + its M x N temporary) runs at on a build host."""
    from spateo_amd._synthetic import make_config

    c = _C4_SAMPLE
    X, V, _ = make_config("C4", N=c["n"])
    kw = dict(M=c["M"], lambda_=c["lambda_"], lstsq_method="scipy", MaxIter=c["steps"], ecr=0.0, seed=0)
    ref, table = _fixture_fit(f"c4_M{c['M']}_lam{c['lambda_']}_n{c['n']}_s{c['steps']}", X, V, kw, stride=c["stride"])
    return X, V, kw, ref, table


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_c4_generator_at_the_cpu_baseline_size(st, dtype):
    """200 k x 3000, lambda_ = 0.02, 10 steps on the bench's workload generator against the committed oracle fixture
    (every 8th cell of V and P stored; floors computed on all cells)."""
    X, V, kw, ref, table = _c4_sample_case()
    got = st.SparseVFC(X, V, None, dtype=dtype, device="cuda:0", **kw)
    assert got["iteration"] == _C4_SAMPLE["steps"] - 1
    _check_fixture_fit("C4 generator 200k x 3000 lambda=0.02", dtype, got, ref, table, stride=_C4_SAMPLE["stride"])


# ------------------------------------------------------------------------------------------- the benchmark's own sizes
# VERDICT r3 "missing" #1: nothing beyond 250 k cells was compared with the oracle.  tests/golden/make_stream_oracle.py
# runs the float64 oracle with U streamed over cell chunks (oracle/streamed_oracle.py: the very statements of
# sparsevfc_oracle.em_step, bit-identical to the `sumorder` form of the in-memory oracle) at BASELINE config 3's stated size,
# at one rank's share of config 4, and for ONE EM iteration of config 4 itself; every 32nd - 256th cell of V / P is stored,
# the floors (LAPACK driver swapped; the sums over cells made of another number of pieces) are evaluated on ALL cells.
_STREAM_CASES = {"c3_full": ("C3", 2_000_000, 2000), "c4_rank": ("C4", 1_000_000, 3000), "c4_step": ("C4", 8_000_000, 3000),
                 "c4_3step": ("C4", 8_000_000, 3000)}


@functools.lru_cache(maxsize=None)
def _stream_fixture(name):
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", f"stream_oracle_{name}.npz")
    if not os.path.exists(path):
        pytest.skip(f"{os.path.basename(path)} has not been generated (tests/golden/make_stream_oracle.py {name}: hours of host time)")
    with np.load(path) as z:
        fx = {k: z[k] for k in z.files}
    cfg, n, M = _STREAM_CASES[name]
    assert (int(fx["n"]), int(fx["M"])) == (n, M) and float(fx["lambda_"]) == 0.02
    ref = dict(V=fx["V"], P=fx["P"], sigma2=float(fx["sigma2"]), E_traj=fx["E_traj"], iteration=int(fx["iteration"]),
               vmax=float(fx["vmax"]))
    # no float32-kernel variant was run at these sizes: the float32 mode is held to the float64-mode floors (stricter)
    table = {q: (float(fx[f"floor_{q}"]), float(fx[f"floor_{q}"])) for q in ("V", "sigma2", "P", "E")}
    table["_variants"] = {v: {q: float(fx[f"var_{v}_{q}"]) for q in ("V", "sigma2", "P", "E")} for v in ("eigh", "sumorder")
                          if f"var_{v}_V" in fx}  # (c4_3step: the eigh variant alone)
    return fx, ref, table


def _strict_fixture_check(tag, dtype, dev, table, tight=TIGHT):
    """Every quantity - P included - within max(1.25 x its own reference floor, its base tolerance)."""
    base = _base_tolerances(dtype, tight)
    lim = {k: F.tol(dtype, table, k, base[k]) for k in dev}
    fl = {k: table[k][0] for k in dev}
    print(f"{tag} {dtype}: " + "; ".join(
        f"{k} gpu {dev[k]:.2e} / floor {fl[k]:.2e} (x{dev[k] / max(fl[k], 1e-300):.2f}, limit {lim[k]:.2e})" for k in dev))
    print(F.fmt(table))
    bad = {k: (dev[k], lim[k]) for k in dev if not dev[k] <= lim[k]}
    assert not bad, bad


@pytest.mark.parametrize("dtype", ["float64", "float32"])
@pytest.mark.parametrize("case", ["c3_full", "c4_rank", "c4_3step"])
def test_fit_against_the_streamed_oracle_at_benchmark_sizes(st, case, dtype):
    """BASELINE config 3 AT ITS STATED SIZE (2 M cells x 2000 control points) and one rank's share of config 4 (1 M x 3000):
    5 EM iterations at Spateo's lambda_ = 0.02 through the drop-in ``SparseVFC`` against the committed streamed-oracle
    fixture - field, sigma^2, P and the energy trajectory each within 1.25 x its own reference floor.  c4_3step (round 6,
    VERDICT r5 next #7): config 4 ITSELF, 8 M cells x 3000 control points, THREE EM iterations - the rank-deficient regime
    of the coefficient solve starts at the second - floor = the oracle's own lstsq -> eigh variant over the same three."""
    from spateo_amd._synthetic import make_config

    fx, ref, table = _stream_fixture(case)
    cfg, n, M = _STREAM_CASES[case]
    X, V, _ = make_config(cfg, N=n)
    got = st.SparseVFC(X, V, None, M=M, lambda_=0.02, lstsq_method="scipy", MaxIter=int(fx["steps"]), ecr=0.0, seed=0,
                       dtype=dtype, device="cuda:0")
    np.testing.assert_array_equal(got["ctrl_idx"], fx["ctrl_idx"])
    np.testing.assert_allclose(got["beta"], float(fx["beta"]), rtol=1e-12)
    assert got["iteration"] == ref["iteration"] == int(fx["steps"]) - 1
    stride = int(fx["stride"])
    dev = {"V": float(np.abs(got["V"][::stride] - ref["V"]).max() / ref["vmax"]),
           "sigma2": abs(got["sigma2"] - ref["sigma2"]) / ref["sigma2"],
           "P": float(np.abs(got["P"][::stride] - ref["P"]).max()),
           "E": float(np.abs((got["E_traj"] - ref["E_traj"]) / ref["E_traj"]).max())}
    _strict_fixture_check(f"{case} ({n} x {M}, {int(fx['steps'])} iterations)", dtype, dev, table)
    del got
    torch.cuda.empty_cache()


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_one_em_iteration_of_the_benchmark_workload_against_the_streamed_oracle(st, dtype):
    """BASELINE config 4 itself - 8 M cells x 3000 control points, the very arrays bench.py times - ONE EM iteration from
    the initial state against the streamed oracle: P and the energy (they precede the solve) at the mode's tolerance, the
    field and sigma^2 after the M-step within 1.25 x the reference's own floor for that step."""
    from spateo_amd._synthetic import make_config
    from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess

    fx, ref, table = _stream_fixture("c4_step")
    X, V, M = make_config("C4")
    valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=M, seed=0, device="cuda:0")
    del X, V
    np.testing.assert_array_equal(idx, fx["ctrl_idx"])
    np.testing.assert_allclose(beta, float(fx["beta"]), rtol=1e-12)
    eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype=dtype, device="cuda:0")
    eng.init_state(gamma=0.9)
    E, _ = eng.em_step(a=5, lambda_=0.02, minP=1e-5, theta=0.75)
    Vg, Pg, _ = eng.results()
    stride = int(fx["stride"])
    np.testing.assert_allclose(Pg[::stride], ref["P"], rtol=TOL[dtype], atol=1e-9)
    np.testing.assert_allclose(E, float(ref["E_traj"][0]), rtol=TOL[dtype])
    dev = {"V": float(np.abs(Vg[::stride] - ref["V"]).max() / ref["vmax"]),
           "sigma2": abs(eng.sigma2 - ref["sigma2"]) / ref["sigma2"]}
    print(f"8 M x 3000, one EM iteration, solver {eng.solver_stats}")
    _strict_fixture_check("c4_step (8000000 x 3000, 1 iteration)", dtype, dev,
                          {k: table[k] for k in ("V", "sigma2", "_variants")})
    eng.k.drop_ublk()
    del eng
    torch.cuda.empty_cache()


# ------------------------------------------------------------------------------------------- the headline's own solve
HEADLINE_SOLVE_LOG = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpurun_out",
                                  "headline_solve_parity.jsonl")


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_the_deflated_solve_on_the_benchmark_system_against_scipy_lstsq(st, dtype):
    """VERDICT r4 "missing" #5 / r5 next #7: the 20 timed steps of bench.py run the rank-deficient DEFLATED solve (kept rank
    ~830 of 3000) on the 8 M x 3000 system, but the one-iteration oracle fixture of that size is a full-rank step.  Here the
    systems of the SECOND to FIFTH EM iteration  A = U^T P U + lambda sigma^2 K,  R = U^T P Y  (72 MB each, P from the GPU
    fit itself) are copied to the host and solved with the reference's call, ``scipy.linalg.lstsq`` (gelsd; reached through
    sparsevfc.py:110,194,250), and with its `gelss` / truncated `eigh` variants for the floor; the fields  V = U C  on every
    256th cell (U = the GPU's own kernel values of the mode, the U that was fitted) of the GPU's solve of that iteration -
    and, for the fifth, of its Jacobi path on the same factor for the record - must sit within 1.25 x that floor."""
    import json

    import scipy.linalg
    from spateo_amd._synthetic import make_config
    from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess

    X, V, M = make_config("C4")
    valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=M, seed=0, device="cuda:0")
    del X, V
    eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype=dtype, device="cuda:0")
    eng.init_state(gamma=0.9)
    kw = dict(a=5, lambda_=0.02, minP=1e-5, theta=0.75)
    caps = []
    inner = eng._solve_all

    def capturing(ls2):
        host = inner(ls2)
        st_ = eng.solver_stats
        caps.append(dict(G=eng.G.cpu().numpy(), R=eng.R[0].cpu().numpy(), C=eng.C_new[0].cpu().numpy(), ls2=float(ls2),
                         path=("deflated" if eng.rank_deficient else "cholesky"),
                         kept=(st_["rank"][-1] if eng.rank_deficient else M),
                         frank=((st_.get("factor_rank") or [M])[-1] if eng.rank_deficient else M),
                         block=((st_.get("block") or [0])[-1] if eng.rank_deficient else 0)))
        return host

    eng.em_step(**kw)                       # iteration 1 (the streamed-oracle fixture c4_step covers it)
    eng._solve_all = capturing
    for _ in range(4):                      # iterations 2 - 5
        eng.em_step(**kw)
    assert eng.rank_deficient and eng.mn_method == "deflated"
    assert caps[-1]["block"] in (128, 256), eng.solver_stats  # the deflated block answered the fifth, not the fallback
    k = eng.k
    # the Jacobi path on the fifth system (fresh workspace, no hint), for the record
    Gd, Rd = torch.from_numpy(caps[-1]["G"]).to(k.device), torch.from_numpy(caps[-1]["R"]).to(k.device)
    Cj = torch.empty_like(Rd)
    info, einfo = k.zeros(1, dtype=torch.int32), k.zeros(12, dtype=torch.float64)
    k._lr_ws = None
    k.solve_minnorm_lr(Gd, eng.K, caps[-1]["ls2"], Rd, Cj, info, einfo)
    assert int(info.cpu()[0]) == 0
    Cjh = Cj.cpu().numpy()
    # U on every 256th cell, generated by the mode's own kernel_value (the U that was fitted)
    rows = torch.arange(0, eng.n_local, 256, device=k.device)
    Us = k.con_k(eng.x4[rows][:, :3].contiguous(), eng.ctrl4[:, :3].contiguous(), eng.beta).to(torch.float64).cpu().numpy()
    Kh = eng.K.cpu().numpy()
    eng.k.drop_ublk()
    del eng, Gd, Rd, Cj
    torch.cuda.empty_cache()
    os.makedirs(os.path.dirname(HEADLINE_SOLVE_LOG), exist_ok=True)
    bad = []
    for it, cp in enumerate(caps, start=2):
        if dtype == "float32" and it in (2, 4):
            continue  # (host time: a 3000 x 3000 gelsd + gelss is ~15 s; float64 checks all four iterations, float32 3 and 5)
        A = cp["G"] + cp["ls2"] * Kh
        C_ref = scipy.linalg.lstsq(A, cp["R"])[0]                      # the reference's call (gelsd)
        C_eigh = _eigh_solver(A, cp["R"])
        Vr = Us @ C_ref
        vmax = np.abs(Vr).max()
        dev = lambda C: float(np.abs(Us @ C - Vr).max() / vmax)  # noqa: E731
        floors = {"eigh": dev(C_eigh)}
        if it == 5:  # both witnesses of the floor for the iteration rounds 4 - 5 reported; the truncated eigh alone before it
            floors["gelss"] = dev(scipy.linalg.lstsq(A, cp["R"], lapack_driver="gelss")[0])
        floor = max(floors.values())
        got = dev(cp["C"])
        rec = {"case": f"c4_solve (8000000 x 3000, system of EM iteration {it})", "iteration": it, "dtype": dtype,
               "path": cp["path"], "block": int(cp["block"]), "kept_rank": int(cp["kept"]), "factor_rank": int(cp["frank"]),
               "V_gpu": got, "floor": floor, "floors": floors, "ratio": got / max(floor, 1e-300),
               "cells_compared": int(len(Us))}
        if it == 5:
            rec["V_jacobi"] = dev(Cjh)
            rec["ratio_jacobi"] = rec["V_jacobi"] / max(floor, 1e-300)
        print(json.dumps(rec))
        with open(HEADLINE_SOLVE_LOG, "a") as fh:
            fh.write(json.dumps(rec) + "\n")
        if not got <= max(F.ALLOW * floor, TOL[dtype]):
            bad.append(rec)
    assert not bad, bad


# ------------------------------------------------------------------------------------------- BASELINE config 5 organ
@functools.lru_cache(maxsize=None)
def _c5_case():
    from spateo_amd._synthetic import ellipsoid_cloud, _noisy

    rng = np.random.default_rng(100)
    axes = rng.uniform(100, 400, 3)
    X = ellipsoid_cloud(rng, 250_000, axes)
    V = _noisy(rng, X, 0.05, 0.05, 2.0)
    kw = dict(M=500, lambda_=3.0, lstsq_method="scipy", seed=0, MaxIter=30)
    ref = svo.SparseVFC(X, V, None, **kw)
    return X, V, kw, ref, F.floor_table(X, V, None, ref, kw)


@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_c5_one_organ_at_its_size(st, dtype):
    """One organ of BASELINE config 5 at its stated size (250 k cells, M = 500) against the oracle."""
    from spateo_amd.vectorfield import SparseVFC_many

    X, V, kw, ref, table = _c5_case()
    got = SparseVFC_many([(X, V, None)], device="cuda:0", dtype=dtype, **kw)[0]
    _check_fit("C5 organ", dtype, got, ref, table)


# ------------------------------------------------------------------------------------------- float32 vs float64 at scale
@pytest.mark.parametrize("lambda_", [3.0, 0.02])
def test_float32_mode_vs_float64_mode_at_the_per_rank_size(st, lambda_):
    """SUPPLEMENT, not parity (the oracle comparison at this size is
    test_fit_against_the_streamed_oracle_at_benchmark_sizes[c4_rank]): 1 M cells x 3000 control points, 10 EM iterations,
    the float32 mode's field against the float64 mode's inside the 1e-3 float32 tolerance, for both lambdas."""
    from spateo_amd._synthetic import make_config
    from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess

    X, V, M = make_config("C4", N=1_000_000)
    valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=M, seed=0)
    res = {}
    for dt in ("float64", "float32"):
        eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype=dt, device="cuda:0")
        eng.init_state(0.9)
        for _ in range(10):
            eng.em_step(lambda_=lambda_)
        res[dt] = (eng.results()[0], eng.sigma2, dict(eng.solver_stats))
        eng.k.drop_ublk()
        del eng
        torch.cuda.empty_cache()
    rel = _rel(res["float32"][0], res["float64"][0])
    print(f"1M x 3000 lambda={lambda_}: f32 vs f64 field {rel:.2e}; sigma2 {res['float32'][1]:.6g} vs "
          f"{res['float64'][1]:.6g}; solver f64 {res['float64'][2]} f32 {res['float32'][2]}")
    assert rel < 1e-3
    np.testing.assert_allclose(res["float32"][1], res["float64"][1], rtol=1e-3)
