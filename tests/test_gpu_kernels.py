"""GPU parity tests, kernel by kernel: every libmvf entry point (called through the C ABI) against the float64
NumPy oracle on the same seeded inputs, plus the reference twins' golden vectors.

Tolerances (written here, per BASELINE.json north_star): float64 mode 1e-5 relative or far tighter where the
quantity is well conditioned; float32 mode 1e-3 relative (kernels are well inside it)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

from oracle import dg_oracle as dgo  # noqa: E402
from oracle import sparsevfc_oracle as svo  # noqa: E402

DTYPES = [("float64", 1e-11), ("float32", 2e-5)]


@pytest.fixture(scope="module")
def st():
    import spateo_amd

    assert torch.cuda.is_available(), "GPU tests need a HIP device"
    return spateo_amd


def _k(dtype):
    from spateo_amd._kernels import HipKernels

    return HipKernels("cuda:0", dtype)


def _relmax(a, b):
    return float(np.abs(np.asarray(a) - np.asarray(b)).max() / max(np.abs(b).max(), 1e-300))


def _cloud(seed, n, m, d=3, scale=30.0):
    rng = np.random.default_rng(seed)
    X = rng.uniform(-1, 1, (n, d)) * scale
    ctrl = X[rng.choice(n, m, replace=False)].copy()
    return rng, X, ctrl


# ------------------------------------------------------------------------------------------------- con_K
@pytest.mark.parametrize("dtype,tol", DTYPES)
def test_con_k_golden_and_shapes(st, golden, dtype, tol):
    g = golden
    x, y, beta = g["conk_x"], g["conk_y"], float(g["conk_beta"])
    K = st.con_K(x, y, beta, dtype=dtype)
    assert K.shape == (7, 5) and K.dtype == np.float64
    assert _relmax(K, g["conk_K_cdist"]) < tol
    Kd, D = st.con_K(x, y, beta, return_d=True, dtype=dtype)
    assert D.shape == (7, 3, 5)
    assert _relmax(Kd, g["conk_K_diff"]) < tol
    # D is computed on centred coordinates in the cell dtype: exact in float64 up to the centring round-off
    assert np.abs(D - g["conk_D"]).max() < (1e-12 if dtype == "float64" else 1e-5)
    Krow = st.con_K(x[2], y, beta, dtype=dtype)
    assert Krow.shape == (5,)  # single row flattened like the reference
    assert _relmax(Krow, g["conk_K_row"]) < tol
    assert _relmax(st.con_K(g["conk_x2"], g["conk_y2"], 0.5, dtype=dtype), g["conk_K_2d"]) < tol


@pytest.mark.parametrize("dtype,tol", DTYPES)
@pytest.mark.parametrize("n,m,d", [(1000, 300, 3), (513, 257, 2), (130, 1030, 5), (1, 7, 3), (33, 1, 1)])
def test_con_k_vs_oracle(st, dtype, tol, n, m, d):
    rng = np.random.default_rng(n + m)
    x = rng.standard_normal((n, d)) * 5
    y = rng.standard_normal((m, d)) * 5
    beta = 0.02
    K = st.con_K(x, y, beta, dtype=dtype)
    Kr = svo.con_K(x, y, beta)
    assert K.shape == Kr.shape
    assert np.abs(K - Kr).max() < tol  # K <= 1: absolute == relative to max


def test_con_k_empty(st):
    k = _k("float32")
    x = torch.zeros((0, 3), dtype=torch.float32, device="cuda:0")
    y = torch.zeros((5, 3), dtype=torch.float32, device="cuda:0")
    assert k.con_k(x, y, 0.1).shape == (0, 5)
    assert k.con_k(y, x, 0.1).shape == (5, 0)


# ------------------------------------------------------------------------------------------------- apply
@pytest.mark.parametrize("dtype,tol", [("float64", 1e-11), ("float32", 5e-5)])
@pytest.mark.parametrize("n,m", [(2000, 700), (257, 5), (1025, 513)])
def test_apply_vs_oracle(st, dtype, tol, n, m):
    rng, X, ctrl = _cloud(n * 7 + m, n, m)
    beta = 0.004
    C = rng.standard_normal((m, 3))
    Y = rng.standard_normal((n, 3))
    P = rng.uniform(0.0, 1.0, n)
    k = _k(dtype)
    center = ctrl.mean(0)
    x4, c4, y4 = k.to_x4(X, center), k.to_x4(ctrl, center), k.to_x4(Y)
    Cd = torch.from_numpy(C).to("cuda:0")
    Pd = torch.from_numpy(P.astype(np.float32 if dtype == "float32" else np.float64)).to("cuda:0")
    stats = torch.zeros(1, dtype=torch.float64, device="cuda:0")
    V4, r = k.apply(x4, c4, beta, Cd, y4, Pd, stats)
    V = V4[:, :3].double().cpu().numpy()
    Vr = svo.con_K(X, ctrl, beta) @ C
    assert _relmax(V, Vr) < tol
    assert np.all(V4[:, 3].cpu().numpy() == 0)
    rr = np.sum((Y - Vr) ** 2, 1)
    assert _relmax(r.double().cpu().numpy(), rr) < 10 * tol
    assert abs(float(stats[0]) - float(Pd.double().cpu().numpy() @ rr)) / (P @ rr) < 10 * tol


# ------------------------------------------------------------------------------------------------- E-step
@pytest.mark.parametrize("dtype", ["float64", "float32"])
def test_estep_vs_oracle(st, dtype):
    rng = np.random.default_rng(11)
    n = 5000
    Y = rng.standard_normal((n, 3))
    V = Y + 0.2 * rng.standard_normal((n, 3))
    V[:50] += 30.0  # gross outliers
    sigma2, gamma, a, minP, theta = 0.05, 0.9, 5.0, 1e-5, 0.75
    r = np.sum((Y - V) ** 2, 1)
    npdt = np.float32 if dtype == "float32" else np.float64
    r_dev = r.astype(npdt)
    k = _k(dtype)
    rd = torch.from_numpy(r_dev).to("cuda:0")
    mins = k.estep_min(rd, sigma2).cpu().numpy()
    t1 = np.exp(-r_dev.astype(np.float64) / (2 * sigma2))
    assert mins[1] == (t1 == 0).sum()
    assert mins[1] > 0  # the outliers underflow: the min-non-zero rule is exercised
    np.testing.assert_allclose(mins[0], t1[t1 != 0].min(), rtol=1e-12)
    Pd = torch.empty(n, dtype=rd.dtype, device="cuda:0")
    stats = torch.zeros(5, dtype=torch.float64, device="cuda:0")
    k.estep_p(rd, sigma2, gamma, a, 3, minP, theta, float(mins[0]), Pd, stats)
    # mvf_estep (ABI 7): both phases in one call, the fill taken from the device, the statistics OVERWRITTEN - the same bits
    P2 = torch.empty(n, dtype=rd.dtype, device="cuda:0")
    stats2 = torch.full((5,), 123.0, dtype=torch.float64, device="cuda:0")
    k.estep(rd, sigma2, gamma, a, 3, minP, theta, P2, stats2)
    assert torch.equal(P2, Pd) and torch.equal(stats2, stats)
    assert float(stats[4]) == mins[1]  # the cells that took the fill
    # oracle on the SAME (possibly float32-rounded) residuals: V' chosen so that ||Y - V'||^2 == r_dev
    Vp = Y.copy()
    Vp[:, 0] -= np.sqrt(r_dev.astype(np.float64))
    Vp[:, 1:] = Y[:, 1:]
    Pr, _ = svo.get_P(Y, Vp, sigma2, gamma, a)
    rq = np.sum((Y - Vp) ** 2, 1)
    Pf = np.maximum(Pr[:, 0], minP)
    tol = 1e-9 if dtype == "float64" else 2e-6
    np.testing.assert_allclose(Pd.double().cpu().numpy(), Pf, rtol=tol, atol=1e-12)
    s = stats.cpu().numpy()
    np.testing.assert_allclose(s[0], Pr[:, 0] @ rq, rtol=1e-6)
    np.testing.assert_allclose(s[1], Pr.sum(), rtol=1e-9)
    np.testing.assert_allclose(s[2], Pf.sum(), rtol=1e-6)
    assert s[3] == (Pd.double().cpu().numpy() > theta).sum()


# ------------------------------------------------------------------------------------------------- Gram + rhs
@pytest.mark.parametrize("dtype,tol", [("float64", 1e-11), ("float32", 3e-6)])
@pytest.mark.parametrize("n,m", [(3000, 300), (1000, 128), (700, 130), (5000, 40)])
def test_gram_vs_oracle(st, dtype, tol, n, m):
    rng, X, ctrl = _cloud(n + 3 * m, n, m)
    beta = 0.003
    Y = rng.standard_normal((n, 3))
    P = rng.uniform(1e-5, 1.0, n)
    npdt = np.float32 if dtype == "float32" else np.float64
    k = _k(dtype)
    center = ctrl.mean(0)
    x4, c4, y4 = k.to_x4(X, center), k.to_x4(ctrl, center), k.to_x4(Y)
    Pd = torch.from_numpy(P.astype(npdt)).to("cuda:0")
    G = torch.empty(m, m, dtype=torch.float64, device="cuda:0")
    R = torch.empty(m, 3, dtype=torch.float64, device="cuda:0")
    k.gram(x4, Pd, y4, c4, beta, G, R)
    U = svo.con_K(X, ctrl, beta)
    Pq = Pd.double().cpu().numpy()
    UP = U.T * Pq[None, :]
    Gr, Rr = UP @ U, UP @ Y
    Gd, Rd = G.cpu().numpy(), R.cpu().numpy()
    assert np.array_equal(Gd, Gd.T), "G must be exactly symmetric (mirrored tiles)"
    assert _relmax(Gd, Gr) < tol  # catches row/col transposition: off-diagonal tile blocks are not symmetric
    assert _relmax(Rd, Rr) < 10 * tol
    # run-to-run determinism (fixed-order reduction)
    G2 = torch.empty_like(G)
    R2 = torch.empty_like(R)
    k.gram(x4, Pd, y4, c4, beta, G2, R2)
    assert torch.equal(G, G2) and torch.equal(R, R2)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
@pytest.mark.parametrize("n,m", [(3000, 300), (700, 130), (5000, 40), (256, 128), (2000, 700), (1500, 1080)])
def test_gram_cached_u_is_bit_identical_to_recompute(st, n, m, dtype):
    """The cached-U Gram kernel streams materialised float32 kernel values; they are the same kernel_value() bits the
    recompute kernel generates, accumulated in the same order -> identical G and R.  (m = 300, 130, 700, 1080: the last tile
    column has <= 64 live control points - the float64 kernel runs its tiles two per job, with an odd tile left over at
    m = 300 and 1080.)"""
    rng, X, ctrl = _cloud(n + m, n, m)
    beta = 0.003
    Y = rng.standard_normal((n, 3))
    P = torch.from_numpy(rng.uniform(1e-5, 1.0, n).astype(np.float32 if dtype == "float32" else np.float64)).to("cuda:0")
    k = _k(dtype)
    center = ctrl.mean(0)
    x4, c4, y4 = k.to_x4(X, center), k.to_x4(ctrl, center), k.to_x4(Y)
    G0 = torch.empty(m, m, dtype=torch.float64, device="cuda:0")
    R0 = torch.empty(m, 3, dtype=torch.float64, device="cuda:0")
    k.gram(x4, P, y4, c4, beta, G0, R0)
    k.build_ublk(x4, c4, beta)
    G1, R1 = torch.empty_like(G0), torch.empty_like(R0)
    k.gram(x4, P, y4, c4, beta, G1, R1)
    assert k._ublk is not None
    assert torch.equal(R0, R1)
    assert torch.equal(G0, G1)
    # and against the oracle
    U = svo.con_K(X, ctrl, beta)
    Gr = (U.T * P.double().cpu().numpy()[None, :]) @ U
    assert _relmax(G1.cpu().numpy(), Gr) < (3e-6 if dtype == "float32" else 1e-11)


@pytest.mark.parametrize("dtype", ["float32", "float64"])
def test_gram_partial_tiles_in_phases(st, dtype, monkeypatch):
    """When the (capped) slices need more partial tiles than the buffer holds, the tile stage runs in several launches
    that reuse the buffer, each folded into G before the next starts (float64 mode at 8 M cells x 3000 control points:
    977 slices in 2 phases).  Forced here with 256-cell slices at 1.2 M cells x 300 control points (4688 slices x 6 tile
    pairs > the 22 888 tiles of the minimum buffer): G equals the single-launch result to the rounding of the changed
    summation partition and the NumPy Gram matrix; recompute and cached kernels stay bit-identical; the split stage calls
    (tile stage first, reductions later: what bench.py does to time the kernel) give the same bits."""
    n, m = 1_200_000, 300
    rng, X, ctrl = _cloud(77, n, m)
    beta = 0.003
    Y = rng.standard_normal((n, 3))
    P = torch.from_numpy(rng.uniform(1e-5, 1.0, n).astype(np.float32 if dtype == "float32" else np.float64)).to("cuda:0")
    k = _k(dtype)
    center = ctrl.mean(0)
    x4, c4, y4 = k.to_x4(X, center), k.to_x4(ctrl, center), k.to_x4(Y)
    out = {}
    from spateo_amd import _lib

    for name, sl in (("one_launch", None), ("phases", "256")):
        _lib.debug_option("slice_len", int(sl or 0))   # developer option of the library (mvf.h), not an environment knob
        kk = _k(dtype)
        G = torch.empty(m, m, dtype=torch.float64, device="cuda:0")
        R = torch.empty(m, 3, dtype=torch.float64, device="cuda:0")
        kk.gram(x4, P, y4, c4, beta, G, R)
        kk.build_ublk(x4, c4, beta)
        Gc, Rc = torch.empty_like(G), torch.empty_like(R)
        kk.gram_events = []  # the timed path: tile stage and reductions as separate calls
        kk.gram(x4, P, y4, c4, beta, Gc, Rc)
        assert len(kk.gram_events) == 1
        assert torch.equal(G, Gc) and torch.equal(R, Rc)
        assert torch.equal(G, G.T)
        out[name] = G.cpu().numpy()
        if sl is not None:  # the buffer really is smaller than all partial tiles
            need = kk.lib.mvf_gram_workspace_bytes(n, m, kk.cdtype)
            assert need < (n // 256) * 6 * 128 * 128 * 8
    _lib.debug_option("slice_len", 0)
    assert _relmax(out["phases"], out["one_launch"]) < 1e-12
    U = svo.con_K(X[:100_000], ctrl, beta)
    # (the oracle on the first 100 k cells only bounds nothing about the rest; the full comparison is the one above)
    Gr = (U.T * P.double().cpu().numpy()[None, :100_000]) @ U
    kk = _k(dtype)
    G = torch.empty(m, m, dtype=torch.float64, device="cuda:0")
    R = torch.empty(m, 3, dtype=torch.float64, device="cuda:0")
    _lib.debug_option("slice_len", 256)
    # 100 k cells x 6 pairs at 256-cell slices = 2346 tiles: one phase; 1.2 M: two - same kernels, so a cheap oracle check
    try:
        kk.gram(x4[:100_000].contiguous(), P[:100_000].contiguous(), y4[:100_000].contiguous(), c4, beta, G, R)
    finally:
        _lib.debug_option("slice_len", 0)
    assert _relmax(G.cpu().numpy(), Gr) < (3e-6 if dtype == "float32" else 1e-11)


# ------------------------------------------------------------------------------------------------- solve
@pytest.mark.parametrize("m,nrhs", [(64, 3), (100, 3), (300, 2), (517, 1)])
def test_solve_spd_vs_numpy(st, m, nrhs):
    rng = np.random.default_rng(m)
    A = rng.standard_normal((m, 2 * m))
    G = A @ A.T / (2 * m)
    Kc = rng.standard_normal((m, m))
    K = Kc @ Kc.T / m
    R = rng.standard_normal((m, nrhs))
    ls2 = 0.37
    k = _k("float64")
    dev = "cuda:0"
    Gd, Kd, Rd = (torch.from_numpy(a).to(dev) for a in (G, K, R))
    C = torch.empty(m, nrhs, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    k.solve(Gd, Kd, ls2, 0.0, Rd, C, info)
    assert int(info.cpu()[0]) == 0
    Cr = np.linalg.solve(G + ls2 * K, R)
    assert _relmax(C.cpu().numpy(), Cr) < 1e-9


def test_solve_pivot_range(st):
    """pivots = [min L_jj^2, max L_jj^2] of the un-regularised Cholesky: the host's numerical-rank certificate."""
    m = 200
    rng = np.random.default_rng(5)
    A = rng.standard_normal((m, 3 * m))
    G = A @ A.T / m
    k = _k("float64")
    dev = "cuda:0"
    C = torch.empty(m, 3, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    piv = torch.zeros(2, dtype=torch.float64, device=dev)
    k.solve(torch.from_numpy(G).to(dev), torch.zeros(m, m, dtype=torch.float64, device=dev), 0.0, 0.0,
            torch.ones(m, 3, dtype=torch.float64, device=dev), C, info, piv)
    d = np.diag(np.linalg.cholesky(G)) ** 2
    np.testing.assert_allclose(piv.cpu().numpy(), [d.min(), d.max()], rtol=1e-10)


def _minnorm_ref(A, R, rcond=np.finfo(float).eps):
    w, q = np.linalg.eigh((A + A.T) / 2)
    keep = np.abs(w) > rcond * np.abs(w).max()
    return (q[:, keep] / w[keep]) @ (q[:, keep].T @ R), w, keep


# mvf_solve_minnorm_lr (pivoted-Cholesky factor + Jacobi on its columns) / mvf_solve_minnorm_lrd (same factor, only the
# subspace below the cut-off computed and deflated; the Jacobi path below 512 factor columns) / mvf_solve_minnorm (all m columns)
METHODS = ["lowrank", "deflated", "full"]


def _run_minnorm(k, G, K, ls2, R, shift=2.0 ** -36, reuse_R=None, rcond=None, method="full", rank_hint=0):
    dev = "cuda:0"
    m, nrhs = R.shape
    Gd, Kd, Rd = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (G, K, R))
    C = torch.empty(m, nrhs, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    einfo = torch.zeros(12, dtype=torch.float64, device=dev)
    if method in ("lowrank", "deflated"):
        def run(Rt, Ct, **kw):
            k.solve_minnorm_lr(Gd, Kd, ls2, Rt, Ct, info, einfo, rcond=rcond, rank_hint=rank_hint,
                               deflate=method == "deflated", **kw)
    else:
        def run(Rt, Ct, **kw):
            k.solve_minnorm(Gd, Kd, ls2, shift, Rt, Ct, info, einfo, rcond=rcond, **kw)
    run(Rd, C)
    out = [C.cpu().numpy(), int(info.cpu()[0]), einfo.cpu().numpy()]
    if reuse_R is not None:
        R2 = torch.from_numpy(np.ascontiguousarray(reuse_R)).to(dev)
        C2 = torch.empty_like(R2)
        run(R2, C2, reuse=True)
        out.append(C2.cpu().numpy())
    return out


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("m,nrhs", [(2, 1), (33, 3), (64, 3), (100, 6), (300, 2), (517, 8), (700, 3)])
def test_solve_minnorm_full_rank_equals_the_inverse(st, m, nrhs, method):
    """Well conditioned SPD system: nothing is truncated, the minimum-norm solve is the ordinary solve; the reported
    extreme eigenvalues are LAPACK's."""
    rng = np.random.default_rng(m)
    A = rng.standard_normal((m, 2 * m))
    G = A @ A.T / (2 * m)
    Kc = rng.standard_normal((m, m))
    K = Kc @ Kc.T / m
    R = rng.standard_normal((m, nrhs))
    ls2 = 0.37
    C, info, e, C2 = _run_minnorm(_k("float64"), G, K, ls2, R, reuse_R=R[:, :1] * 2.0, method=method)
    assert info == 0 and e[0] == np.floor(e[0]) and e[0] < 40, e  # converged (x.5 would mean the sweep cap was hit)
    w = np.linalg.eigvalsh(G + ls2 * K)
    assert int(e[1]) == m
    if method != "full":
        assert int(e[6]) == m  # the factor keeps every column of a well-conditioned matrix
    if method == "deflated" and m >= 128:
        # the deflated solve knows lambda_max from its power iteration (stopped when two successive Rayleigh quotients agree
        # to 1e-7: a few 1e-6 on this clustered Wishart spectrum, 1e-8 on kernel Gram matrices) and the smallest Ritz value
        # of its block only
        np.testing.assert_allclose(e[2], w.max(), rtol=2e-5)
        assert w.min() * (1 - 1e-9) <= e[3]
    else:
        np.testing.assert_allclose([e[2], e[3]], [w.max(), w.min()], rtol=1e-10)
    assert _relmax(C, np.linalg.solve(G + ls2 * K, R)) < 1e-9
    assert _relmax(C2, 2.0 * C[:, :1]) < 1e-12  # reuse applies the same decomposition to another right-hand side


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("m,rank", [(96, 40), (200, 1), (257, 130), (1200, 700)])
def test_solve_minnorm_exactly_rank_deficient(st, m, rank, method):
    """A = B B^T with B m x rank: with a cut-off above the rounding level of the null-space eigenvalues (rcond = 1e-12;
    at the default eps some of them land above it, for LAPACK just the same) the result is pinv(A) R."""
    rng = np.random.default_rng(rank)
    B = rng.standard_normal((m, rank))
    G = B @ B.T
    R = G @ rng.standard_normal((m, 3))  # consistent right-hand sides, as U^T P Y is for U^T P U
    C, info, e = _run_minnorm(_k("float64"), G, np.zeros((m, m)), 0.0, R, rcond=1e-12, method=method)
    assert info == 0 and int(e[1]) == rank
    if method != "full":
        assert rank <= int(e[6]) <= rank + max(8, m // 100)  # the pivoted factor stops at the rounding level of the matrix
    Cr = np.linalg.pinv(G, rcond=1e-13, hermitian=True) @ R
    assert _relmax(C, Cr) < 1e-8
    assert _relmax(G @ C, R) < 1e-10


def _kernel_system(n, m, seed=0, lambda_=0.02, s2=1e-3, unit_p=False):
    """A numerically rank-deficient SparseVFC system: Gaussian-kernel Gram of n cells on m control points."""
    from spateo_amd._synthetic import make_config

    X, Y, _ = make_config("C2", N=n)
    valid, Xv, Yv, idx, ctrl, beta = svo.sparsevfc_setup(X, Y, M=m, seed=seed)
    K = svo.con_K(ctrl, ctrl, beta)
    U = svo.con_K(Xv, ctrl, beta)
    P = np.ones(len(Xv)) if unit_p else np.clip(np.random.default_rng(seed).random(len(Xv)), 1e-5, 1.0)
    UP = U.T * P[None, :]
    return U, UP @ U, K, UP @ Yv, lambda_ * s2


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("n,m", [(4000, 300), (6000, 1000)])
def test_solve_minnorm_kernel_gram_vs_scipy_lstsq(st, n, m, method):
    """The regime Spateo's default lambda_ puts the solve in: U^T P U + lambda sigma^2 K numerically rank deficient.
    Parity quantity = the FIELD U C (C lives in the numerical null space).  The reference noise floor is the deviation
    between scipy.linalg.lstsq (gelsd) and the mathematically identical truncated eigh solve, both on the CPU."""
    import scipy.linalg

    U, G, K, R, ls2 = _kernel_system(n, m)
    A = G + ls2 * K
    C_ls = scipy.linalg.lstsq(A, R)[0]
    C_eh, w, keep = _minnorm_ref(A, R)
    F = U @ C_ls
    sc = np.abs(F).max()
    floor = np.abs(U @ C_eh - F).max() / sc
    C, info, e = _run_minnorm(_k("float64"), G, K, ls2, R, method=method)
    dev = np.abs(U @ C - F).max() / sc
    print(f"m={m} {method}: cond {w.max() / np.abs(w).min():.1e} kept {keep.sum()} gpu kept {int(e[1])} (factor rank "
          f"{int(e[6])}) sweeps {e[0]} floor(lstsq vs eigh) {floor:.2e}  gpu vs lstsq {dev:.2e}")
    assert info == 0
    assert dev < max(2.0 * floor, 1e-9)
    assert abs(int(e[1]) - int(keep.sum())) <= max(2, m // 50)  # eigenvalues at the cut-off may fall either side
    np.testing.assert_allclose(e[2], w.max(), rtol=1e-6 if method == "deflated" else 1e-9)


@pytest.mark.parametrize("n,m", [(20000, 2000), (30000, 3000)])
def test_deflated_solve_is_the_truncated_solve(st, n, m):
    """mvf_solve_minnorm_lrd at the sizes it is made for (factor rank ~ 0.3 - 0.6 m): the field against scipy.linalg.lstsq
    within twice the reference's own lstsq-vs-eigh floor, against the Jacobi path of the same factor far below that floor
    (both apply the same eps lambda_max cut-off to the same factor), the same kept rank, eight right-hand sides, reuse, and
    mvf_pinv_diag refusing the workspace (it needs the full decomposition)."""
    import scipy.linalg
    from spateo_amd import _lib

    U, G, K, R, ls2 = _kernel_system(n, m)
    R = np.concatenate([R, R[:, ::-1] * 0.5, R[:, :2] - R[:, 1:3]], 1)[:, :8]
    A = G + ls2 * K
    F = U @ scipy.linalg.lstsq(A, R)[0]
    sc = np.abs(F).max()
    floor = np.abs(U @ _minnorm_ref(A, R)[0] - F).max() / sc
    k = _k("float64")
    Cj, info_j, ej = _run_minnorm(k, G, K, ls2, R, method="lowrank")
    Cd, info_d, ed, Cd2 = _run_minnorm(k, G, K, ls2, R, method="deflated", reuse_R=R[:, 2:5] * 4.0)
    assert info_j == 0 and info_d == 0 and int(ed[6]) >= 512
    dev_d, dev_j, between = (np.abs(U @ a - b).max() / sc for a, b in ((Cd, F), (Cj, F), (Cd, U @ Cj)))
    print(f"m={m}: factor rank {int(ed[6])} kept {int(ed[1])} (Jacobi path {int(ej[1])}), block sweeps {ed[0]}; floor {floor:.2e} "
          f"deflated vs lstsq {dev_d:.2e} Jacobi vs lstsq {dev_j:.2e} deflated vs Jacobi {between:.2e}")
    assert ed[0] == np.floor(ed[0]) and abs(int(ed[1]) - int(ej[1])) <= 1
    assert dev_d < 2.0 * floor and between < 0.1 * floor
    assert _relmax(Cd2, 4.0 * Cd[:, 2:5]) < 1e-11  # (a power of two: the coefficients amplify the rounding of 3 R by 1e10)
    with pytest.raises(_lib.MVFError, match="deflated decomposition"):
        k.pinv_diag(torch.zeros(4, 4, dtype=torch.float64, device="cuda:0"), torch.zeros(m, 4, dtype=torch.float64, device="cuda:0"),
                    1.0, lowrank=True)


def test_deflated_solve_shrinks_its_block_when_few_directions_are_truncated(st):
    """Second call on a workspace (rank_hint, as the EM loop does) after a call that deflated <= 72 directions: a block of 128
    vectors with three applications instead of 256 with two - the same truncated count and the same field as the Jacobi path
    run through the same two calls (greedy, then hinted: the hinted factor is another factor of the same matrix)."""
    U, G, K, R, ls2 = _kernel_system(24000, 1200, s2=1e-5, unit_p=True)
    kd, kj = _k("float64"), _k("float64")
    C0, info0, e0 = _run_minnorm(kd, G, K, ls2, R, method="deflated")
    C1, info1, e1 = _run_minnorm(kd, G, K, ls2, R, method="deflated", rank_hint=int(e0[6]))
    J0, _, f0 = _run_minnorm(kj, G, K, ls2, R, method="lowrank")
    J1, _, f1 = _run_minnorm(kj, G, K, ls2, R, method="lowrank", rank_hint=int(f0[6]))
    sc = np.abs(U @ J0).max()
    d0, d1 = np.abs(U @ (C0 - J0)).max() / sc, np.abs(U @ (C1 - J1)).max() / sc
    print(f"factor rank {int(e0[6])} / hinted {int(e1[6])}, truncated {int(e0[6]) - int(e0[1])} / {int(e1[6]) - int(e1[1])}; blocks "
          f"{int(e0[7])} then {int(e1[7])}; field vs the Jacobi path on the same factor {d0:.2e} / {d1:.2e}")
    assert info0 == 0 and info1 == 0 and int(e0[7]) == 256 and int(e1[7]) == 128
    assert 0 < int(e0[6]) - int(e0[1]) <= 72 and int(e0[6]) == int(f0[6]) and int(e1[6]) == int(f1[6])
    assert int(e0[1]) == int(f0[1]) and int(e1[1]) == int(f1[1])
    assert d0 < 1e-4 and d1 < 1e-4


@pytest.mark.parametrize("n,m", [(50000, 500), (20000, 300), (16000, 400)])
def test_deflated_solve_small_factor_uses_a_64_vector_block(st, n, m):
    """Factors of 128 .. 511 columns (M = 128 .. 640 control points: BASELINE configs 2 and 5) take a 64-vector block whose
    Rayleigh-Ritz problem is ONE 64 x 64 Jacobi tile, diagonalised inside a single launch (round 5).  Two calls on one
    workspace, as the EM loop makes them (greedy factor, then the hinted factor with the power iteration warm-started from the
    kept dominant eigenvector): the field against scipy.linalg.lstsq within twice the reference's own lstsq-vs-eigh floor,
    against the Jacobi path run through the same two calls far below it, the same kept rank, lambda_max to 1e-6."""
    import scipy.linalg

    U, G, K, R, ls2 = _kernel_system(n, m, s2=2.4e-3)
    A = G + ls2 * K
    F = U @ scipy.linalg.lstsq(A, R)[0]
    sc = np.abs(F).max()
    C_eh, w, keep = _minnorm_ref(A, R)
    floor = np.abs(U @ C_eh - F).max() / sc
    kd, kj = _k("float64"), _k("float64")
    C0, info0, e0 = _run_minnorm(kd, G, K, ls2, R, method="deflated")
    C1, info1, e1 = _run_minnorm(kd, G, K, ls2, R, method="deflated", rank_hint=int(e0[6]))
    J0, _, f0 = _run_minnorm(kj, G, K, ls2, R, method="lowrank")
    J1, _, f1 = _run_minnorm(kj, G, K, ls2, R, method="lowrank", rank_hint=int(f0[6]))
    d0, d1 = np.abs(U @ (C0 - J0)).max() / sc, np.abs(U @ (C1 - J1)).max() / sc
    print(f"m={m}: factor rank {int(e0[6])} / hinted {int(e1[6])}, kept {int(e0[1])} / {int(e1[1])} (eigh {int(keep.sum())}), blocks "
          f"{int(e0[7])} / {int(e1[7])}, block sweeps {e0[0]} / {e1[0]}; floor {floor:.2e}, deflated vs lstsq "
          f"{np.abs(U @ C1 - F).max() / sc:.2e}, vs the Jacobi path {d0:.2e} / {d1:.2e}; lambda_max rel err cold "
          f"{abs(e0[2] / w.max() - 1):.1e} warm {abs(e1[2] / w.max() - 1):.1e}")
    assert info0 == 0 and info1 == 0
    assert 128 <= int(e0[6]) < 512 and int(e0[7]) == 64 and int(e1[7]) == 64
    assert e0[0] == 1.0 and e1[0] == 1.0                      # the whole Rayleigh-Ritz diagonalisation in one launch
    # the hinted call took the DIRECT form when the first factor kept all m columns (one Cholesky of the permuted matrix with
    # its inverse riding along instead of pivoted factor + L^T L + its Cholesky): same answer as the factor form of the same
    # two calls (developer option lr_no_direct), reuse on another right-hand side included
    from spateo_amd import _lib

    kf = _k("float64")
    old = _lib.debug_option("lr_no_direct", 1)
    try:
        F0, _, g0 = _run_minnorm(kf, G, K, ls2, R, method="deflated")
        F1, _, g1, F1b = _run_minnorm(kf, G, K, ls2, R, method="deflated", rank_hint=int(g0[6]), reuse_R=R[:, :2] * 2.0)
    finally:
        _lib.debug_option("lr_no_direct", old)
    C1r, _, e1r, C1b = _run_minnorm(kd, G, K, ls2, R, method="deflated", rank_hint=int(e1[6]), reuse_R=R[:, :2] * 2.0)
    dd = np.abs(U @ (C1r - F1)).max() / sc
    print(f"    direct form ({'taken' if int(e0[6]) == m else 'not applicable: factor rank < m'}) vs factor form on the hinted "
          f"call: {dd:.2e}; kept {int(e1r[1])} vs {int(g1[1])}")
    # (absolute term 3e-7: on the m = 300 system - nothing truncated, the reference's own lstsq-vs-eigh floor 1.3e-7 - the two
    # forms land 0.9 - 2.0e-7 apart depending on the summation order inside the Cholesky kernels: three builds of round 6)
    assert int(e1r[1]) == int(g1[1]) and int(e1r[6]) == int(g1[6]) and dd < max(0.1 * floor, 3e-7)
    assert _relmax(C1b, 2.0 * C1r[:, :2]) < 1e-11 and _relmax(F1b, 2.0 * F1[:, :2]) < 1e-11
    assert np.abs(U @ (C1r - C1)).max() / sc < max(0.1 * floor, 1e-7)   # third call (same order, warm state) = second call
    assert int(e0[1]) == int(f0[1]) and int(e1[1]) == int(f1[1]) and abs(int(e1[1]) - int(keep.sum())) <= max(2, m // 50)
    assert np.abs(U @ C1 - F).max() / sc < max(2.0 * floor, 1e-9) and np.abs(U @ C0 - F).max() / sc < max(2.0 * floor, 1e-9)
    assert d0 < max(0.1 * floor, 1e-7) and d1 < max(0.1 * floor, 3e-7)   # (d1: the direct form where it applies, see above)
    np.testing.assert_allclose(e0[2], w.max(), rtol=1e-6)
    np.testing.assert_allclose(e1[2], w.max(), rtol=1e-6)
    np.testing.assert_allclose(f1[2], w.max(), rtol=1e-6)     # the Jacobi path's hinted call is warm-started too


def test_direct_form_falls_back_to_the_factor_form_and_cools_down(st):
    """A direct-form attempt that is not accepted (forced here: developer option direct_accept = 1 accepts no deflated
    direction, and this system truncates one) must re-run the call in the factor form - the answer is then the factor form's,
    bit for bit - and leave the next four calls on the workspace to the factor form without another attempt; the fifth tries
    again (and, with the option cleared, succeeds)."""
    import time

    from spateo_amd import _lib

    U, G, K, R, ls2 = _kernel_system(50000, 500, s2=2.4e-3)
    sc = np.abs(U @ np.linalg.lstsq(G + ls2 * K, R, rcond=None)[0]).max()
    kf, kd = _k("float64"), _k("float64")
    old = _lib.debug_option("lr_no_direct", 1)
    try:
        F, _, f0 = _run_minnorm(kf, G, K, ls2, R, method="deflated")
        ref = []
        for _ in range(7):
            F, _, f0 = _run_minnorm(kf, G, K, ls2, R, method="deflated", rank_hint=int(f0[6]))
            ref.append(F)
    finally:
        _lib.debug_option("lr_no_direct", old)
    assert int(f0[6]) == 500 and int(f0[6]) - int(f0[1]) >= 1     # all columns kept, at least one direction truncated
    C, _, e = _run_minnorm(kd, G, K, ls2, R, method="deflated")
    old = _lib.debug_option("direct_accept", 1)
    try:
        got = []
        for i in range(5):                                        # call 1: attempt + fall-back; calls 2 - 5: cool-down
            C, _, e = _run_minnorm(kd, G, K, ls2, R, method="deflated", rank_hint=int(e[6]))
            got.append(C)
    finally:
        _lib.debug_option("direct_accept", old)
    for i in range(5):
        np.testing.assert_array_equal(got[i], ref[i], err_msg=f"call {i + 1} after the failed attempt")
    C6, _, e6 = _run_minnorm(kd, G, K, ls2, R, method="deflated", rank_hint=int(e[6]))     # the sixth tries the direct form again
    assert not np.array_equal(C6, ref[5]) and np.abs(U @ (C6 - ref[5])).max() / sc < 1e-3
    assert int(e6[1]) == int(f0[1]) and int(e6[6]) == 500 and int(e6[7]) == 64 and e6[0] == 1.0


def test_direct_form_without_host_synchronisation(st):
    """mvf_solve_minnorm_lrd_async (ABI 6): the direct form with the acceptance test on the device and no status read inside
    the call.  On a workspace whose previous call kept all m columns it gives the synchronous direct form's answer (same
    kernels in the same order: bit for bit), reports form 2 / repeat flag 0 and leaves a state the next call continues; on a
    workspace that does not hold what the caller claims, and when the acceptance test fails, it sets the repeat flag, and the
    repeat through mvf_solve_minnorm_lrd is the factor form's answer."""
    from spateo_amd import _lib

    U, G, K, R, ls2 = _kernel_system(50000, 500, s2=2.4e-3)
    dev = "cuda:0"
    Gd, Kd, Rd = (torch.from_numpy(np.ascontiguousarray(a)).to(dev) for a in (G, K, R))

    def async_call(k, form):
        C = torch.empty(500, R.shape[1], dtype=torch.float64, device=dev)
        info = torch.zeros(1, dtype=torch.int32, device=dev)
        einfo = torch.zeros(12, dtype=torch.float64, device=dev)
        k.solve_minnorm_lrd_async(Gd, Kd, ls2, Rd, C, info, einfo, form)
        return C.cpu().numpy(), int(info.cpu()[0]), einfo.cpu().numpy()

    ks, ka = _k("float64"), _k("float64")
    # reference: three synchronous calls (factor form, direct form fresh, direct form continued)
    S0, _, s0 = _run_minnorm(ks, G, K, ls2, R, method="deflated")
    S1, _, s1 = _run_minnorm(ks, G, K, ls2, R, method="deflated", rank_hint=int(s0[6]))
    S2, _, s2 = _run_minnorm(ks, G, K, ls2, R, method="deflated", rank_hint=int(s1[6]))
    assert int(s0[6]) == 500 and int(s0[8]) == 1 and int(s1[8]) == 2 and int(s2[8]) == 2 and s2[9] == 0.0
    # the same sequence with the second and third call asynchronous
    A0, _, a0 = _run_minnorm(ka, G, K, ls2, R, method="deflated")
    A1, i1, a1 = async_call(ka, int(a0[8]))
    A2, i2, a2 = async_call(ka, int(a1[8]))
    assert i1 == 0 and i2 == 0 and a1[9] == 0.0 and a2[9] == 0.0 and int(a1[8]) == 2 and int(a2[8]) == 2
    np.testing.assert_array_equal(A1, S1)
    np.testing.assert_array_equal(A2, S2)
    np.testing.assert_array_equal(a2[:8], s2[:8])
    # form_hint + 4 ("the matrix moved": thirteen warm power steps instead of five) on an unchanged matrix: lambda_max was
    # settled already, the same decisions, the same field to rounding
    km = _k("float64")
    M0, _, m0 = _run_minnorm(km, G, K, ls2, R, method="deflated")
    M1, j1, m1 = async_call(km, int(m0[8]) | 4)
    M2, j2, m2 = async_call(km, int(m1[8]) | 4)
    assert j1 == 0 and j2 == 0 and m1[9] == 0.0 and m2[9] == 0.0 and int(m2[8]) == 2 and int(m2[1]) == int(s2[1])
    assert np.abs(U @ (M2 - S2)).max() / np.abs(U @ S2).max() < 1e-9 and abs(m2[2] / s2[2] - 1) < 1e-9
    A3, _, a3 = _run_minnorm(ka, G, K, ls2, R, method="deflated", rank_hint=500)   # a synchronous call continues from it
    S3, _, s3 = _run_minnorm(ks, G, K, ls2, R, method="deflated", rank_hint=500)
    np.testing.assert_array_equal(A3, S3)
    # a claim the workspace does not back (its last call was a FACTOR form, the caller says "direct"): not accepted
    kb = _k("float64")
    B0, _, b0 = _run_minnorm(kb, G, K, ls2, R, method="deflated")
    _, ib, b1 = async_call(kb, 2)
    assert ib == 0 and b1[9] == 1.0
    Bf, _, bf = _run_minnorm(kb, G, K, ls2, R, method="deflated", rank_hint=500)   # the repeat: factor form (cool-down mark)
    assert int(bf[8]) == 1 and int(bf[6]) == 500
    # an acceptance test that fails (developer option: accept no deflated direction; this system truncates one)
    kc = _k("float64")
    C0, _, c0 = _run_minnorm(kc, G, K, ls2, R, method="deflated")
    old = _lib.debug_option("direct_accept", 1)
    try:
        _, ic, c1 = async_call(kc, 1)
        assert ic == 0 and c1[9] == 1.0
        Cf, _, cf = _run_minnorm(kc, G, K, ls2, R, method="deflated", rank_hint=500)
    finally:
        _lib.debug_option("direct_accept", old)
    old = _lib.debug_option("lr_no_direct", 1)
    try:
        kr = _k("float64")
        R0, _, r0 = _run_minnorm(kr, G, K, ls2, R, method="deflated")
        R1, _, r1 = _run_minnorm(kr, G, K, ls2, R, method="deflated", rank_hint=500)
    finally:
        _lib.debug_option("lr_no_direct", old)
    assert int(cf[8]) == 1
    np.testing.assert_array_equal(Cf, R1)


def test_deflated_solve_falls_back_when_the_block_is_too_small(st):
    """With a cut-off far above eps more eigenvalues of the factor fall below it than the 256-vector block holds: the call
    must notice and return the Jacobi path's result."""
    U, G, K, R, ls2 = _kernel_system(20000, 2000)
    k = _k("float64")
    Cj, info_j, ej = _run_minnorm(k, G, K, ls2, R, method="lowrank", rcond=1e-9)
    Cd, info_d, ed = _run_minnorm(k, G, K, ls2, R, method="deflated", rcond=1e-9)
    assert info_d == 0 and int(ed[6]) - int(ed[1]) > 224 and int(ed[7]) == 0, ed
    assert int(ed[1]) == int(ej[1]) and _relmax(U @ Cd, U @ Cj) < 1e-9


def test_solve_minnorm_lr_zero_matrix_and_non_finite_input(st):
    """The zero matrix has the minimum-norm solution 0 (factor rank 0, nothing to rotate); a non-finite entry anywhere
    is reported through info and never turned into a silent solution."""
    k = _k("float64")
    m = 130
    Z = np.zeros((m, m))
    R = np.random.default_rng(0).standard_normal((m, 3))
    C, info, e = _run_minnorm(k, Z, Z, 0.0, R, method="lowrank")
    assert info == 0 and int(e[6]) == 0 and np.all(C == 0.0)
    B = np.random.default_rng(1).standard_normal((m, 2 * m))
    G = B @ B.T
    for bad in (np.nan, np.inf):
        Gb = G.copy()
        Gb[7, 90] = Gb[90, 7] = bad
        _, info, _ = _run_minnorm(k, Gb, Z, 0.0, R, method="lowrank")
        assert info != 0
    C, info, e = _run_minnorm(k, G, Z, 0.0, R, method="lowrank")  # the workspace is fine afterwards
    assert info == 0 and _relmax(C, np.linalg.solve(G, R)) < 1e-9


@pytest.mark.parametrize("m,lowrank", [(1100, True), (300, False)])
def test_pinv_diag_is_the_leverage_of_the_kernel_matrix(st, m, lowrank):
    """mvf_pinv_diag: diag(U pinv(A) U^T) from the decomposition the solve left in its workspace.  With A = U^T U these
    are the leverage scores of U's rows: in [0, 1], summing to the kept rank; checked against NumPy's eigh with the same
    cut-off (1e-10 relative: clear of the rounding level, so the truncated sum is well determined)."""
    n = 3000
    rng, X, ctrl = _cloud(m, n, m)
    beta = 0.004
    U = svo.con_K(X, ctrl, beta)
    G = U.T @ U
    R = rng.standard_normal((m, 3))
    rcond = 1e-10
    k = _k("float64")
    C, info, e = _run_minnorm(k, G, np.zeros((m, m)), 0.0, R, method="lowrank" if lowrank else "full", rcond=rcond)
    assert info == 0
    center = ctrl.mean(0)
    x4, c4 = k.to_x4(X, center), k.to_x4(ctrl, center)
    d = k.pinv_diag(x4, c4, beta, rcond=rcond, lowrank=lowrank).cpu().numpy()
    w, q = np.linalg.eigh(G)
    keep = w > rcond * w.max()
    Z = U @ q[:, keep]
    dr = (Z * Z / w[keep]).sum(1)
    print(f"m={m} lowrank={lowrank}: kept {int(e[1])} (numpy {keep.sum()}), leverage sum {d.sum():.3f}, max rel dev "
          f"{np.abs(d - dr).max() / dr.max():.2e}")
    assert abs(int(e[1]) - int(keep.sum())) <= 2
    assert np.abs(d - dr).max() < 1e-4 * dr.max() + 2.0 * abs(int(e[1]) - int(keep.sum()))
    assert d.min() > -1e-9 and d.max() < 1.0 + 1e-6
    assert abs(d.sum() - int(e[1])) < 1e-3 * int(e[1])


def _greedy_pivoted_cholesky(A, tol):
    """Diagonally pivoted Cholesky, largest remaining diagonal first (ties: lowest index), stopped when it is <= tol: the order
    and the pivot values mvf_solve_minnorm_lr's cold factorisation must take (NumPy float64, right-looking)."""
    A = A.copy()
    m = len(A)
    d = np.diag(A).copy()
    order, piv, rows = [], [], []
    alive = np.ones(m, dtype=bool)
    for _ in range(m):
        dm = np.where(alive, d, -np.inf)
        p = int(np.argmax(dm))
        if not dm[p] > tol:
            break
        row = A[p].copy()
        for y in rows:
            row -= y * y[p]
        y = np.where(alive, row / np.sqrt(row[p]), 0.0)
        order.append(p)
        piv.append(row[p])
        rows.append(y)
        d = d - y * y
        alive[p] = False
    return np.asarray(order), np.asarray(piv)


@pytest.mark.parametrize("n,m", [(4000, 300), (6000, 500), (5000, 512)])
def test_cold_pivoted_factorisation_up_to_512_columns_takes_the_greedy_pivots(st, n, m):
    """Up to 512 columns the cold factorisation (no pivot order to follow: a fit's first EM iteration) runs its pivot steps 32
    per launch in ONE workgroup with the sub-block's rows in LDS (pchol_steps_kernel, round 6).  Its pivots must be the greedy
    ones: the same order as a NumPy diagonally pivoted Cholesky wherever the choice is not a rounding-level near-tie, the same
    pivot values, the same rank; and the solve on top of it the reference's solution."""
    import scipy.linalg

    k = _k("float64")
    U, G, K, R, ls2 = _kernel_system(n, m)
    A = G + ls2 * K
    C, info, e = _run_minnorm(k, G, K, ls2, R, method="lowrank")
    assert info == 0
    order, vals, tol = k.lr_pivot_order(m, with_values=True)
    ref_order, ref_vals = _greedy_pivoted_cholesky(A, tol)
    r = len(order)
    assert abs(r - len(ref_order)) <= 2, (r, len(ref_order))
    q = min(r, len(ref_order))
    same = order[:q] == ref_order[:q]
    first_diff = int(np.argmin(same)) if not same.all() else q
    # the early pivots are separated by far more than rounding: identical; later ones may swap between near-equal diagonals
    assert first_diff >= min(q, 64), (first_diff, order[:8], ref_order[:8])
    assert len(set(order.tolist())) == r  # every column at most once
    # (pivot values: relative to the largest one - the last pivots are differences of numbers 1e10 times their size)
    np.testing.assert_allclose(vals[:first_diff], ref_vals[:first_diff], rtol=1e-6, atol=1e-12 * vals[0])
    F_ref = U @ scipy.linalg.lstsq(A, R)[0]
    dev = np.abs(U @ C - F_ref).max() / np.abs(F_ref).max()
    print(f"m={m}: rank {r} (numpy greedy {len(ref_order)}), identical pivots up to step {first_diff}, field vs lstsq {dev:.2e}")
    assert dev < 0.1  # (sanity only: where the system is rank deficient the solve's parity is the floor tests' business)


def test_solve_minnorm_lr_follows_the_previous_pivot_order(st):
    """rank_hint > 0: the factorisation follows the pivot order the previous call left in the workspace, 64 columns per
    three launches, accepting each pivot only while it is not small against the remaining diagonal.  Same matrix: same
    solution (to the level the truncated solve is determined); a nearby matrix (other weights P): the hint still covers
    almost every column and the result is that matrix's own solution; a hint from an unrelated matrix, or a hint for a
    workspace that holds none, is rejected or ignored - never wrong."""
    import scipy.linalg

    k = _k("float64")
    U, G, K, R, ls2 = _kernel_system(6000, 1000)
    F_ref = U @ scipy.linalg.lstsq(G + ls2 * K, R)[0]
    sc = np.abs(F_ref).max()
    C0, info, e0 = _run_minnorm(k, G, K, ls2, R, method="lowrank")                      # greedy
    r0 = int(e0[6])
    C1, info1, e1 = _run_minnorm(k, G, K, ls2, R, method="lowrank", rank_hint=r0)      # same matrix, hinted
    assert info == 0 and info1 == 0
    d0, d1 = np.abs(U @ C0 - F_ref).max() / sc, np.abs(U @ C1 - F_ref).max() / sc
    print(f"greedy: rows {r0} kept {int(e0[1])} vs lstsq {d0:.2e}; hinted: rows {int(e1[6])} kept {int(e1[1])} vs lstsq {d1:.2e}")
    assert abs(int(e1[6]) - r0) <= 64 and abs(int(e1[1]) - int(e0[1])) <= 8
    assert d1 < max(2.0 * d0, 1e-9)
    # a nearby matrix: other weights
    U2, G2, K2, R2, _ = _kernel_system(6000, 1000, seed=0, lambda_=0.02, s2=0.9e-3)
    P2 = np.clip(np.random.default_rng(5).random(len(U2)) ** 2, 1e-5, 1.0)
    G2 = (U2.T * P2[None, :]) @ U2
    F2 = U2 @ scipy.linalg.lstsq(G2 + ls2 * K2, R2)[0]
    C2g, _, e2g = _run_minnorm(_k("float64"), G2, K2, ls2, R2, method="lowrank")        # greedy on a fresh workspace
    C2h, info2, e2h = _run_minnorm(k, G2, K2, ls2, R2, method="lowrank", rank_hint=int(e1[6]))
    dg_, dh_ = (np.abs(U2 @ C - F2).max() / np.abs(F2).max() for C in (C2g, C2h))
    print(f"nearby matrix: greedy rows {int(e2g[6])} vs lstsq {dg_:.2e}; hinted rows {int(e2h[6])} vs lstsq {dh_:.2e}")
    assert info2 == 0 and dh_ < max(2.0 * dg_, 1e-9)
    # an unrelated, well-conditioned matrix with a stale hint in the workspace, and a hint without any history
    rng = np.random.default_rng(3)
    B = rng.standard_normal((1000, 2000))
    Gw = B @ B.T / 2000
    Rw = rng.standard_normal((1000, 3))
    for kk, hint in ((k, int(e2h[6])), (_k("float64"), 500)):
        Cw, infow, ew = _run_minnorm(kk, Gw, np.zeros_like(Gw), 0.0, Rw, method="lowrank", rank_hint=hint)
        assert infow == 0 and int(ew[1]) == 1000
        assert _relmax(Cw, np.linalg.solve(Gw, Rw)) < 1e-9


@pytest.mark.parametrize("method", METHODS)
@pytest.mark.parametrize("m", [2000, 3000])
def test_solve_large_rank_deficient_vs_scipy_lstsq(st, m, method):
    """the coefficient solve at the headline size against scipy.linalg.lstsq on a genuinely
    rank-deficient U^T P U + lambda sigma^2 K, asserting on the field U C; the jitter-Cholesky mode is measured
    beside it (deviation as a function of the jitter it needed)."""
    import time

    import scipy.linalg

    U, G, K, R, ls2 = _kernel_system(12000, m)
    A = G + ls2 * K
    C_ls = scipy.linalg.lstsq(A, R)[0]
    C_eh, w, keep = _minnorm_ref(A, R)
    F = U @ C_ls
    sc = np.abs(F).max()
    floor = np.abs(U @ C_eh - F).max() / sc
    k = _k("float64")
    _, _, e0 = _run_minnorm(k, G, K, ls2, R, method=method)  # warm-up (workspace allocation)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    C, info, e = _run_minnorm(k, G, K, ls2, R, method=method, rank_hint=int(e0[6]))
    torch.cuda.synchronize()
    ms = 1e3 * (time.perf_counter() - t0)
    dev = np.abs(U @ C - F).max() / sc
    # jitter-escalated Cholesky (lstsq_method="cholesky") on the same system
    d = "cuda:0"
    Gd, Kd, Rd = (torch.from_numpy(a).to(d) for a in (G, K, R))
    Cc = torch.empty(m, 3, dtype=torch.float64, device=d)
    inf = torch.zeros(1, dtype=torch.int32, device=d)
    jit, devs = 0.0, []
    while True:
        k.solve(Gd, Kd, ls2, jit, Rd, Cc, inf)
        if int(inf.cpu()[0]) == 0:
            devs.append((jit, np.abs(U @ Cc.cpu().numpy() - F).max() / sc))
            if len(devs) == 3:
                break
        jit = max(jit * 10, 1e-15)
    print(f"m={m} {method}: kept {keep.sum()}/{m} (gpu {int(e[1])}, factor rank {int(e[6])}), sweeps {e[0]}, {ms:.1f} ms "
          f"incl. H2D; floor {floor:.2e}; "
          f"min-norm vs lstsq {dev:.2e}; jitter-Cholesky vs lstsq: " + ", ".join(f"{j:g}: {v:.2e}" for j, v in devs))
    assert info == 0 and e[0] == np.floor(e[0])
    assert dev < max(2.0 * floor, 1e-9)


def test_solve_reports_non_psd(st):
    m = 96
    G = -np.eye(m)
    K = np.zeros((m, m))
    k = _k("float64")
    dev = "cuda:0"
    C = torch.empty(m, 3, dtype=torch.float64, device=dev)
    info = torch.zeros(1, dtype=torch.int32, device=dev)
    k.solve(torch.from_numpy(G).to(dev), torch.from_numpy(K).to(dev), 0.0, 0.0,
            torch.ones(m, 3, dtype=torch.float64, device=dev), C, info)
    assert int(info.cpu()[0]) == 1  # first pivot fails -> 1 + index 0


def test_quadform(st):
    rng = np.random.default_rng(2)
    m = 333
    K = rng.standard_normal((m, m))
    C = rng.standard_normal((m, 3))
    k = _k("float64")
    out = torch.zeros(1, dtype=torch.float64, device="cuda:0")
    k.quadform(torch.from_numpy(K).to("cuda:0"), torch.from_numpy(C).to("cuda:0"), out)
    np.testing.assert_allclose(float(out.cpu()[0]), np.trace(C.T @ K @ C), rtol=1e-11)


# ------------------------------------------------------------------------------------------------- evaluators
@pytest.mark.parametrize("dtype,tol", [("float64", 1e-10), ("float32", 1e-4)])
def test_evaluators_golden(st, golden, dtype, tol):
    g = golden
    vfd = {"X_ctrl": g["dg_Xc"], "C": g["dg_C"], "beta": float(g["dg_beta"])}
    Xq = g["dg_Xq"]
    vf = st.SvcVectorField(dtype=dtype, device="cuda:0")
    vf.vf_dict = vfd
    vf.func = lambda x: st.vector_field_function(x, vfd, dtype=dtype)
    assert _relmax(vf.func(Xq), g["dg_v"]) < tol
    J = vf.get_Jacobian()(Xq)
    assert J.shape == (3, 3, len(Xq))
    assert _relmax(J, g["dg_J_loop"]) < tol
    J1 = vf.get_Jacobian()(Xq[3])
    assert J1.shape == (3, 3) and _relmax(J1, g["dg_J_1d"]) < tol
    acc, acc_mat = vf.compute_acceleration(Xq)
    assert _relmax(acc, g["dg_acc"]) < tol and _relmax(acc_mat, g["dg_acc_mat"]) < tol
    c2, c2m = vf.compute_curvature(Xq, formula=2)
    assert _relmax(c2, g["dg_curv2"]) < tol and _relmax(c2m, g["dg_curv2_mat"]) < tol
    c1, c1m = vf.compute_curvature(Xq, formula=1)
    assert c1m is None and _relmax(c1, g["dg_curv1"]) < tol
    curl = vf.compute_curl(Xq)
    assert curl.shape == (len(Xq), 3, 3) and _relmax(curl, g["dg_curl"]) < tol
    tor = vf.compute_torsion(Xq)
    assert tor.shape == (len(Xq), 3, 3) and _relmax(tor, g["dg_tor"]) < 10 * tol
    assert _relmax(vf.compute_divergence(Xq, vectorize_size=4), g["dg_div"]) < tol
    # 2-D
    vfd2 = {"X_ctrl": g["dg_Xc"][:, :2].copy(), "C": g["dg_C"][:, :2].copy(), "beta": vfd["beta"]}
    vf2 = st.SvcVectorField(dtype=dtype, device="cuda:0")
    vf2.vf_dict = vfd2
    curl2 = vf2.compute_curl(Xq[:, :2].copy())
    assert curl2.shape == (len(Xq),) and _relmax(curl2, g["dg_curl2d"]) < tol
    with pytest.raises(Exception, match="torsion is only defined in 3 dimension"):
        vf2.compute_torsion(Xq[:, :2].copy())


def test_evaluators_large_vs_oracle(st):
    rng, X, ctrl = _cloud(5, 3000, 700)
    vfd = {"X_ctrl": ctrl, "C": rng.standard_normal((700, 3)), "beta": 0.004}
    vf = st.SvcVectorField(dtype="float64", device="cuda:0")
    vf.vf_dict = vfd
    J = vf.get_Jacobian()(X)
    Jr = dgo.Jacobian_rkhs_gaussian(X, vfd, vectorize=True)
    assert _relmax(J, Jr) < 1e-10
    div = vf.compute_divergence(X)
    assert _relmax(div, np.trace(Jr)) < 1e-10


def test_evaluators_matrix_form_far_from_the_centroid_with_cancelling_coefficients(st):
    """``eval_mfma_kernel`` forms ``sum_m K_m C_mf (p - c_m)_i`` as ``p_i v_f - W_fi`` from ONE matrix product
    ``[v | W] = K [C | C (x) c]``.  Stress of that difference: queries up to ~40 kernel widths from the control points'
    centroid and coefficients that cancel to 1e-6 of their size (what a fitted C looks like at lambda_ = 0.02) - the Jacobian,
    divergence and acceleration still agree with the float64 oracle's pair-by-pair sums to 1e-9 of their own scale."""
    rng = np.random.default_rng(17)
    m, n = 600, 5000
    ctrl = rng.uniform(-1, 1, (m, 3)) * np.array([500.0, 300.0, 200.0]) + np.array([4000.0, -2500.0, 900.0])
    X = rng.uniform(-1, 1, (n, 3)) * np.array([520.0, 310.0, 210.0]) + np.array([4000.0, -2500.0, 900.0])
    big = 1e6 * rng.standard_normal((m // 2, 3))
    C = np.concatenate([big, -big + rng.standard_normal((m // 2, 3))])      # neighbours in the list, not in space
    near = np.argsort(np.linalg.norm(ctrl[:, None] - ctrl[None], axis=2) + 1e9 * np.eye(m), axis=1)[:, 0]
    C[near[: m // 2]] = -C[: m // 2] + rng.standard_normal((m // 2, 3))   # and spatial neighbours that nearly cancel
    vfd = {"X_ctrl": ctrl, "C": C, "beta": 0.004}
    vf = st.SvcVectorField(dtype="float64", device="cuda:0")
    vf.vf_dict = vfd
    J = vf.get_Jacobian()(X)
    Jr = dgo.Jacobian_rkhs_gaussian(X, vfd, vectorize=True)
    assert _relmax(J, Jr) < 1e-9
    assert _relmax(vf.compute_divergence(X), np.trace(Jr)) < 1e-9
    vf.func = lambda x: st.vector_field_function(x, vfd, dtype="float64")
    v = svo.con_K(X, ctrl, 0.004) @ C
    acc_norm, acc = vf.compute_acceleration(X)
    ref = np.einsum("fin,ni->nf", Jr, v)
    assert _relmax(acc, ref) < 1e-9 and _relmax(acc_norm, np.linalg.norm(ref, axis=1)) < 1e-9


@pytest.mark.parametrize("dtype,tol", [("float64", 1e-10), ("float32", 2e-4)])
def test_jacobian_determinant_on_the_device_and_one_fused_pass(st, dtype, tol):
    """MVF_EVAL_JDET (the `obs` slot of ``morphofield_jacobian``, ``differential_geometry.py:336``) against
    ``np.linalg.det`` of the oracle's Jacobians; Jacobian, determinant, curl and divergence of the same points come out of
    ONE ``eval_mfma_kernel`` launch (the later calls are device -> host copies), 70 k points so that the pinned path is taken."""
    from spateo_amd import preprocess as _pre, vectorfield as vfm

    rng, X, ctrl = _cloud(6, 70_000, 400)
    vfd = {"X_ctrl": ctrl, "C": rng.standard_normal((400, 3)), "beta": 0.004}
    vfm.clear_eval_cache()
    vf = st.SvcVectorField(dtype=dtype, device="cuda:0")
    vf.vf_dict = vfd
    k = vfm._shared_kernels("cuda:0", dtype)
    calls, real = [], k.eval
    k.eval = lambda *a, **kw: (calls.append(a[4]), real(*a, **kw))[1]
    try:
        J, det = vf.jacobian_with_det(X)
        curl = vf.compute_curl(X=X)
        div = st.SvcVectorField(dtype=dtype, device="cuda:0")
        div.vf_dict = dict(vfd)
        div = div.compute_divergence(X=X.copy())
    finally:
        k.eval = real
        vfm.clear_eval_cache()
    assert len(calls) == 1
    Jr = np.concatenate([dgo.Jacobian_rkhs_gaussian(X[lo:lo + 10_000], vfd, vectorize=True)
                         for lo in range(0, len(X), 10_000)], axis=2)
    jmax = np.abs(Jr).max()
    assert np.abs(J - Jr).max() / jmax < tol
    assert np.abs(det - np.linalg.det(np.moveaxis(Jr, 2, 0))).max() / jmax**3 < tol
    assert np.abs(det - np.linalg.det(np.moveaxis(J, 2, 0))).max() / jmax**3 < 1e-14   # cofactors vs LU on the SAME J
    assert np.abs(curl[:, 1, :] - np.stack([Jr[2, 1] - Jr[1, 2], Jr[0, 2] - Jr[2, 0], Jr[1, 0] - Jr[0, 1]], 1)).max() / jmax < tol
    assert np.abs(div - np.trace(Jr)).max() / jmax < tol


def test_host_transfers_beyond_the_pinned_limit(st):
    """to_host's chunked path (two 32 MB page-locked staging buffers into pageable arrays) returns the same bytes."""
    k = _k("float64")
    g = torch.Generator(device="cuda:0").manual_seed(3)
    a = torch.randn(9_000_001, 2, dtype=torch.float64, device="cuda:0", generator=g)   # 144 MB > PINNED_MAX_BYTES
    b = torch.arange(1_000_003, dtype=torch.int64, device="cuda:0")
    ha, hb = k.to_host([a, b])
    assert ha.shape == (9_000_001, 2) and np.array_equal(ha, a.cpu().numpy()) and np.array_equal(hb, b.cpu().numpy())
    small = k.to_host([a[:1000]])[0]
    assert np.array_equal(small, a[:1000].cpu().numpy())


def test_con_k_return_d_beyond_65535_rows(st):
    """return_d=True for more rows than one mvf_con_k_d launch takes (the reference signature has no such limit)."""
    rng = np.random.default_rng(9)
    x, y = rng.standard_normal((70001, 3)), rng.standard_normal((3, 3))
    K, D = st.con_K(x, y, 0.2, return_d=True, dtype="float64")
    Kr, Dr = svo.con_K(x, y, 0.2, return_d=True)
    assert K.shape == (70001, 3) and D.shape == (70001, 3, 3)
    assert _relmax(K, Kr) < 1e-11 and np.abs(D - Dr).max() < 1e-12


@pytest.mark.parametrize("n,d", [(1, 3), (2, 3), (1000, 3), (300_000, 3), (250_000, 2), (5000, 1)])
def test_unique_rows_on_the_device_is_numpy_unique(st, n, d):
    """mvf_unique_rows == np.unique(X, axis=0, return_index=True) bit for bit: lexicographic order, FIRST occurrence of
    every duplicated row, -0.0 == 0.0, ties in leading columns."""
    rng = np.random.default_rng(n + d)
    X = rng.standard_normal((n, d))
    if n > 10:
        X[:, 0] = np.round(X[:, 0], 1)                 # many ties in the primary column
        dup = rng.integers(0, n, n // 5)
        X[rng.integers(0, n, n // 5)] = X[dup]          # exact duplicate rows
        X[3] = 0.0
        X[7] = -0.0                                     # equal to row 3 for NumPy
    S, idx = _k("float64").unique_rows(X)
    Sr, ir = np.unique(X, axis=0, return_index=True)
    assert S.shape == Sr.shape
    np.testing.assert_array_equal(idx, ir)
    np.testing.assert_array_equal(S, X[ir])


def test_preprocess_uses_the_device_unique_and_matches_the_host(st):
    from spateo_amd import preprocess as _pre, vectorfield as vfm
    from spateo_amd._synthetic import make_config

    X, V, _ = make_config("C3", N=400_000)
    X[1000] = X[5]
    a = vfm.sparsevfc_preprocess(X, V, M=300, seed=0, device="cuda:0")
    old = _pre._DEVICE_UNIQUE_MIN_ROWS
    _pre._DEVICE_UNIQUE_MIN_ROWS = 10**12  # force the host route
    try:
        b = vfm.sparsevfc_preprocess(X, V, M=300, seed=0)
    finally:
        _pre._DEVICE_UNIQUE_MIN_ROWS = old
    for u, v in zip(a, b):
        np.testing.assert_array_equal(u, v)


@pytest.mark.parametrize("m,d", [(256, 3), (500, 3), (1024, 3), (1500, 3), (3000, 3), (2000, 2), (8192, 3)])
def test_knn_bandwidth_on_the_device_is_the_host_bandwidth(st, m, d):
    """dynamo's bandwidth_selector with the neighbour search on the device (all squared distances of a point in LDS,
    bitonic sort, sum of the k - 1 smallest non-self distances) against the kd-tree route: the same distances summed in
    another order."""
    from spateo_amd import preprocess as _pre, vectorfield as vfm

    rng = np.random.default_rng(m + d)
    X = rng.standard_normal((m, d)) * np.array([300.0, 200.0, 150.0])[:d]
    hd = vfm.bandwidth_selector(X, device="cuda:0")
    old = _pre._DEVICE_KNN_MIN_POINTS
    _pre._DEVICE_KNN_MIN_POINTS = 10**9  # force the host route
    try:
        hh = vfm.bandwidth_selector(X)
    finally:
        _pre._DEVICE_KNN_MIN_POINTS = old
    assert abs(hd - hh) <= 1e-12 * hh, (hd, hh)
    # and the preprocessing picks it up: beta from the device route
    if m == 1500:
        from spateo_amd._synthetic import make_config

        Xc, Vc, _ = make_config("C3", N=20_000)
        a = vfm.sparsevfc_preprocess(Xc, Vc, M=m, seed=0, device="cuda:0")
        _pre._DEVICE_KNN_MIN_POINTS = 10**9
        try:
            b = vfm.sparsevfc_preprocess(Xc, Vc, M=m, seed=0)
        finally:
            _pre._DEVICE_KNN_MIN_POINTS = old
        np.testing.assert_array_equal(a[4], b[4])
        assert abs(a[5] - b[5]) <= 1e-12 * b[5]


# ------------------------------------------------------------------------------------------- convex-hull mask (f4)
def test_hull_mask_matches_delaunay_find_simplex():
    """mvf_hull_mask (max over facet half-spaces) against the reference's own formulation, Delaunay(hull vertices)
    .find_simplex(p) >= 0 (spateo/tools/utils.py:205-221), on the 64^3 grid of BASELINE config 2 and on random points
    hugging the hull; the two may differ only for points within 1e-9 x extent of a facet."""
    from scipy.spatial import ConvexHull, Delaunay
    from spateo_amd._kernels import HipKernels
    from spateo_amd._synthetic import make_config
    from spateo_amd.tdr.interpolations.utils import get_X_Y_grid

    X, V, _ = make_config("C2")
    _, _, Grid, mask = get_X_Y_grid(X=X, Y=V, grid_num=[64, 64, 64])  # device path (a GPU is present)
    hull = ConvexHull(X)
    rng = np.random.default_rng(0)
    near = hull.points[hull.vertices][rng.integers(0, len(hull.vertices), 20000)] * rng.uniform(0.97, 1.03, (20000, 1))
    k = HipKernels("cuda:0", "float64")
    extent = float(np.max(hull.max_bound - hull.min_bound))
    tol = 100.0 * np.finfo(np.float64).eps * extent
    tri = Delaunay(hull.points[hull.vertices, :])
    for pts, got in ((Grid, mask), (near, k.hull_mask(near, hull.equations, tol))):
        want = tri.find_simplex(pts) >= 0
        diff = np.flatnonzero(got != want)
        margin = (pts[diff] @ hull.equations[:, :3].T + hull.equations[:, 3]).max(1) if len(diff) else np.zeros(0)
        print(f"hull mask: {len(pts)} points, {int(want.sum())} inside, {len(diff)} differ (max |margin| "
              f"{np.abs(margin).max() if len(diff) else 0:.2e})")
        assert got.dtype == bool and got.shape == want.shape
        assert 0.2 < want.mean() < 0.8
        assert (np.abs(margin) < 1e-9 * extent).all()
    # non-finite points are outside, as find_simplex's -1 (fmax() would silently drop the NaN half-space values)
    bad = np.array([[np.nan, 0, 0], [0, np.inf, 0], [0, 0, -np.inf], [0.0, 0.0, 0.0]])
    np.testing.assert_array_equal(k.hull_mask(bad, hull.equations, tol), [False, False, False, True])


@pytest.mark.parametrize("dtype,n,m", [("float32", 5003, 2000), ("float64", 3001, 1000), ("float32", 701, 3000),
                                       ("float32", 1003, 1004), ("float64", 333, 3000)])
def test_con_k_store_patterns_are_bit_identical(dtype, n, m):
    """The three materialised con_K kernels (row-contiguous spans of 1 / 2 / 4 rows, flat 16 KB chunks with the control
    points in LDS, 2-D row blocks; developer option conk_form = 1 rows | 2 flat | 3 2d) agree bit for bit - ragged last chunk / pass / span
    included - and with the oracle."""
    import os

    from spateo_amd._kernels import HipKernels

    rng = np.random.default_rng(5)
    npdt = np.float32 if dtype == "float32" else np.float64
    x = (rng.standard_normal((n, 3)) * 2).astype(npdt)
    y = (rng.standard_normal((m, 3)) * 2).astype(npdt)
    k = HipKernels("cuda:0", dtype)
    xd, yd = torch.from_numpy(x).to("cuda:0"), torch.from_numpy(y).to("cuda:0")
    out = {}
    from spateo_amd import _lib

    try:
        for code, form in ((1, "rows"), (2, "flat"), (3, "2d")):
            _lib.debug_option("conk_form", code)
            out[form] = k.con_k(xd, yd, 0.37).cpu().numpy()
    finally:
        _lib.debug_option("conk_form", 0)
    default = k.con_k(xd, yd, 0.37).cpu().numpy()
    np.testing.assert_array_equal(out["rows"], out["2d"])
    np.testing.assert_array_equal(out["flat"], out["2d"])
    np.testing.assert_array_equal(default, out["2d"])
    ref = svo.con_K(x.astype(np.float64), y.astype(np.float64), 0.37)
    assert np.abs(default - ref).max() < (2e-6 if dtype == "float32" else 1e-14)


# ------------------------------------------------------------------------------------------- m <= 128: one-launch solve
@pytest.mark.parametrize("m,nrhs,jitter", [(100, 3, 0.0), (128, 6, 0.0), (37, 1, 0.0), (100, 8, 1e-9), (2, 3, 0.0)])
def test_solve_small_single_workgroup_path(st, m, nrhs, jitter):
    """m <= 128 (Spateo's stock M = 100): mvf_solve runs as ONE single-workgroup launch (Cholesky in registers, factor in
    LDS, substitutions).  Against LAPACK, against the blocked multi-launch path (developer option solve_small_off), pivots and the
    non-positive-pivot report included."""
    import os

    rng = np.random.default_rng(100 * m + nrhs)
    A = rng.standard_normal((m, 2 * m + 3))
    G = A @ A.T / (2 * m)
    Kc = rng.standard_normal((m, m))
    K = Kc @ Kc.T / m
    R = rng.standard_normal((m, nrhs))
    ls2 = 0.37
    k = _k("float64")
    dev = "cuda:0"
    Gd, Kd, Rd = (torch.from_numpy(a).to(dev) for a in (G, K, R))

    def run():
        C = torch.full((m, nrhs), np.nan, dtype=torch.float64, device=dev)
        info = torch.full((1,), 7, dtype=torch.int32, device=dev)
        piv = torch.zeros(2, dtype=torch.float64, device=dev)
        k.solve(Gd, Kd, ls2, jitter, Rd, C, info, piv)
        return C.cpu().numpy(), int(info.cpu()[0]), piv.cpu().numpy()

    C1, info1, piv1 = run()
    from spateo_amd import _lib

    _lib.debug_option("solve_small_off", 1)
    try:
        C0, info0, piv0 = run()
    finally:
        _lib.debug_option("solve_small_off", 0)
    Aref = G + ls2 * K
    Aref = Aref + jitter * np.trace(Aref) / m * np.eye(m)
    Cr = np.linalg.solve(Aref, R)
    assert info1 == 0 and info0 == 0
    assert _relmax(C1, Cr) < 1e-9 and _relmax(C0, Cr) < 1e-9
    d = np.diag(np.linalg.cholesky(Aref)) ** 2
    np.testing.assert_allclose(piv1, [d.min(), d.max()], rtol=1e-10)
    np.testing.assert_allclose(piv0, piv1, rtol=1e-10)
    # an indefinite matrix: the first non-positive pivot is reported as 1 + its index, like the blocked path
    Gbad = G.copy()
    Gbad[5 % m, 5 % m] = -10.0
    Cb = torch.empty(m, nrhs, dtype=torch.float64, device=dev)
    infob = torch.zeros(1, dtype=torch.int32, device=dev)
    k.solve(torch.from_numpy(Gbad).to(dev), Kd, 0.0, 0.0, Rd, Cb, infob)
    assert int(infob.cpu()[0]) == 1 + 5 % m
