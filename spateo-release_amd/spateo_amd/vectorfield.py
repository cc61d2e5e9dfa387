"""MI355X-native replacement of the ``dynamo.vectorfield`` seam that Spateo's morphofield path calls.

Seam being replaced: ``SparseVFC(X, Y, Grid, M=, lstsq_method=, lambda_=, seed=, **kw) -> dict``
(call sites ``spateo/tdr/morphometrics/morphofield/sparsevfc.py:189-198,234``,
``spateo/tdr/interpolations/interpolation_sparseVFC.py:63``) and the ``SvcVectorField`` class
(``spateo/tdr/morphometrics/morphofield_dg/differential_geometry.py:25-28`` and the call shapes at
``:68,108-109,154-157,197-198,242-243,291-292,331-335``).  Same names, argument meaning, dict keys, NumPy float64
outputs and exceptions; the arithmetic runs in hand-written HIP kernels behind ``include/mvf.h``.

Host (NumPy, bit-identical across implementations on purpose): finite-row filter, ``np.unique`` rows, control-point
sampling, bandwidth -> beta.  Device: con_K, E-step, weighted Gram + rhs (MFMA), Cholesky solve, field application,
sigma^2 / gamma statistics, evaluators.  Cells are block-sharded across ranks (one process per GPU); one all-reduce
of ``[G | R | scalars]`` per EM step.
"""
from __future__ import annotations

import math
import os
import threading

import numpy as np
import torch

from . import _lib
from ._kernels import HipKernels

__all__ = [
    "clear_eval_cache",
    "SparseVFC_many",
    "integrate_field",
    "genesis_states",
    "GPVectorField",
    "gp_velocity",
    "con_K",
    "vector_field_function",
    "SparseVFC",
    "SparseVFCEngine",
    "SvcVectorField",
    "sparsevfc_preprocess",
    "bandwidth_selector",
    "sample_by_velocity",
]

_DEFAULT_DTYPE = "float64"
DEFLATED_MIN_M = 256  # control points from which the rank-revealing (deflated) solve is the default (640 until round 4)


def _make_kernels(device, dtype):
    """The one place that binds the HIP kernels (tests/ monkeypatch this seam to exercise the host logic on CPU)."""
    return HipKernels(device, dtype)


_TLS = threading.local()


def _shared_kernels(device, dtype):
    """The kernels object of the STATELESS entry points (con_K, evaluators, preprocessing, hull mask): one per (thread,
    device, dtype), built on first use.  A fit owns its own object (``SparseVFCEngine`` keeps workspaces, the kernel-value
    cache and the solver's pivot-order hint in it); the stateless calls used to build a fresh one per call."""
    if device is None and torch.cuda.is_available():
        device = f"cuda:{torch.cuda.current_device()}"
    key = (_make_kernels, str(device), dtype)
    cache = _TLS.__dict__.setdefault("kernels", {})
    k = cache.get(key)
    if k is None:
        k = cache[key] = _make_kernels(device, dtype)
    return k


def _to_host(k, tensors):
    """Device tensors -> host NumPy arrays (pinned staging + one synchronisation on the GPU path)."""
    if hasattr(k, "to_host"):
        return k.to_host(tensors)
    return [t.cpu().numpy().copy() for t in tensors]


_LSTSQ_WARNED = set()

# Whole-call profile (bench.py's `whole_fit`, tools/): with PROFILE_FITS = True every SparseVFC call synchronises the device
# at its phase boundaries and leaves {phase: seconds} in this thread's `last_fit_profile()`.  Off (the default) nothing is
# synchronised or recorded.
PROFILE_FITS = False


def last_fit_profile():
    """{phase: seconds} of this thread's last SparseVFC call made while ``PROFILE_FITS`` was True (else None)."""
    return _TLS.__dict__.get("fit_profile")


class _Phases:
    def __init__(self, device):
        self.on = bool(PROFILE_FITS)
        self.device, self.t, self.out = device, None, {}
        if self.on:
            import time

            self.clock = time.perf_counter
            self.t = self.clock()

    def mark(self, name):
        if not self.on:
            return
        if torch.cuda.is_available():
            torch.cuda.synchronize(self.device)
        now = self.clock()
        self.out[name] = self.out.get(name, 0.0) + now - self.t
        self.t = now

    def done(self):
        if self.on:
            self.out["total_s"] = sum(self.out.values())
            _TLS.fit_profile = self.out


def _check_lstsq_method(method):
    """"scipy": the reference's call (Spateo always passes it, sparsevfc.py:110,194,250) - minimum-norm solve with
    gelsd's eps * s_max cut-off.  "drouin" (dynamo's default: np.linalg.solve on the normal equations lhs^T lhs) has
    the same exact-arithmetic solution; its squared-condition-number arithmetic is not reproduced - the "scipy" solve is
    used and a warning says so once.  "cholesky" (extension): jitter-escalated Cholesky, no truncation."""
    if method in ("scipy", "cholesky"):
        return method
    if method not in _LSTSQ_WARNED:
        _LSTSQ_WARNED.add(method)
        import warnings

        what = "the normal-equations arithmetic of 'drouin' is not reproduced" if method == "drouin" else \
            f"unknown lstsq_method {method!r} (dynamo falls back to 'drouin' with a warning)"
        warnings.warn(f"spateo_amd.SparseVFC: {what}; solving with the 'scipy' (minimum-norm, gelsd cut-off) "
                      f"semantics on the device.", RuntimeWarning, stacklevel=3)
    return "scipy"


def set_default_dtype(dtype: str):
    """Cell dtype used when a call does not pass ``dtype=``: "float64" (parity mode) or "float32" (fast mode)."""
    global _DEFAULT_DTYPE
    if dtype not in ("float32", "float64"):
        raise ValueError("dtype must be 'float32' or 'float64'")
    _DEFAULT_DTYPE = dtype


# =====================================================================================================================
# host-side preprocessing (dynamo SparseVFC steps 1-3) - NumPy on purpose
# =====================================================================================================================
_DEVICE_KNN_MIN_POINTS = 1024


def bandwidth_selector(X: np.ndarray, device=None) -> float:
    """dynamo ``bandwidth_selector``: exact kNN, k = max(2, int(0.2 n)) incl. self; h = sqrt(2) mean(d[:, 1:]) / 1.5.
    From 1024 points on, with a GPU: the neighbour search runs on the device (``mvf_knn_rowsum``: all squared distances of
    a point in LDS, bitonic sort; 0.33 s of kd-tree time at 3000 control points -> about a millisecond); same distances,
    summed in another order: h agrees with the host path to ~1e-15 relative."""
    n = X.shape[0]
    k = max(2, int(0.2 * n))
    if k > n:  # same condition and exception type as the sklearn kNN the reference goes through (a single control point)
        raise ValueError(f"Expected n_neighbors <= n_samples_fit, but n_neighbors = {k}, n_samples_fit = {n}")
    X = np.asarray(X, dtype=np.float64)
    if (_DEVICE_KNN_MIN_POINTS <= n <= 8192 and X.ndim == 2 and X.shape[1] <= 8 and np.isfinite(X).all()
            and torch.cuda.is_available()):
        kern = _shared_kernels(device, "float64")
        if hasattr(kern, "knn_mean_distance"):
            return float(np.sqrt(2) * kern.knn_mean_distance(X, k) / 1.5)
    from scipy.spatial import cKDTree

    distances, _ = cKDTree(X).query(X, k=k)
    d = np.mean(distances[:, 1:]) / 1.5
    return float(np.sqrt(2) * d)


_RNG_LOCK = __import__("threading").Lock()


def sample_by_velocity(V: np.ndarray, n: int, seed: int = 19491001) -> np.ndarray:
    """dynamo ``sample_by_velocity`` (in-tree copy: ``spateo/alignment/methods/sampling.py:225-241``; pinned against outputs of
    that real function, tests/golden/ref_sampling.npz): |V|-weighted sampling without replacement.  dynamo re-seeds NumPy's GLOBAL RNG
    (``np.random.seed(seed)``) and draws from it; here the draw comes from a private ``RandomState(seed)`` - the same
    MT19937 stream, so the same indices - and the global RNG is then left in the state dynamo would leave it in, so
    concurrent fits (``SparseVFC_many``) cannot interleave their draws."""
    return _sample_by_norms(np.linalg.norm(V, axis=1), n, seed)


def _sample_by_norms(tmp_V: np.ndarray, n: int, seed: int = 19491001) -> np.ndarray:
    """The draw of ``sample_by_velocity`` from the row norms themselves (same values in the same order: same indices)."""
    rs = np.random.RandomState(seed)
    p = tmp_V / np.sum(tmp_V)
    idx = rs.choice(np.arange(len(tmp_V)), size=n, p=p, replace=False)
    with _RNG_LOCK:
        np.random.set_state(rs.get_state())
    return idx


_DEVICE_UNIQUE_MIN_ROWS = 200_000


def unique_rows(X: np.ndarray, device=None):
    """``np.unique(X, axis=0, return_index=True)`` (lexicographically sorted unique rows + index of the FIRST
    occurrence of each) without NumPy's structured-view sort, which is the slowest host step at millions of cells
    (12 s at 8 M).  From 200 k rows on, with a GPU: ``mvf_unique_rows`` (stable LSD radix sort over the columns +
    compaction on the device, ~0.1 s at 8 M).  Otherwise on the host: stable argsort on the first coordinate, then a
    stable lexsort only inside runs of equal first coordinates (2-5 s at 8 M).  Both are bit-identical to np.unique for
    finite input; anything else takes the NumPy route."""
    X = np.ascontiguousarray(X)
    n, d = X.shape if X.ndim == 2 else (0, 0)
    if n < 2 or d < 1 or X.dtype.kind != "f" or not np.isfinite(X).all():
        return np.unique(X, axis=0, return_index=True)
    if n >= _DEVICE_UNIQUE_MIN_ROWS and X.dtype == np.float64 and d <= 16 and torch.cuda.is_available():
        k = _shared_kernels(device, "float64")
        if hasattr(k, "unique_rows"):
            return k.unique_rows(X)
    order = np.argsort(X[:, 0], kind="stable")
    x0 = X[order, 0]
    eq = x0[1:] == x0[:-1]
    if d > 1 and eq.any():
        tied = np.zeros(n, dtype=bool)  # positions (in sorted order) that belong to a run of equal first coordinates
        tied[1:] |= eq
        tied[:-1] |= eq
        pos = np.flatnonzero(tied)
        sub = order[pos]
        # stable lexsort (last key is the primary one); the first coordinate keeps each run in its own slots
        keys = tuple(X[sub, c] for c in range(d - 1, 0, -1)) + (X[sub, 0],)
        order[pos] = sub[np.lexsort(keys)]
    S = X[order]
    keep = np.ones(n, dtype=bool)
    keep[1:] = np.any(S[1:] != S[:-1], axis=1)
    return S[keep], order[keep]


def sparsevfc_preprocess(X, Y, M=100, beta=None, velocity_based_sampling=True, seed=0, device=None):
    """valid rows, unique rows, control points and beta exactly as dynamo's SparseVFC picks them."""
    return _sparsevfc_preprocess(X, Y, M, beta, velocity_based_sampling, seed, device)


def _sparsevfc_preprocess(X, Y, M, beta, velocity_based_sampling, seed, device=None):
    valid_ind = np.where(np.isfinite(Y.sum(1)))[0]
    # (all rows finite - the usual case: no gather copies; callers treat Xv / Yv as read-only)
    Xv, Yv = (X, Y) if len(valid_ind) == len(X) else (X[valid_ind], Y[valid_ind])
    if len(Xv) == 0:
        raise ValueError("SparseVFC: no row of Y is finite - nothing to fit.")
    tmp_X, uid = unique_rows(Xv, device)
    M = min(M, tmp_X.shape[0])
    if velocity_based_sampling:
        # (dynamo seeds the global RNG with `seed` here and sample_by_velocity immediately re-seeds it with its own
        # default, so `seed` has no effect on this branch - SURVEY App. A [VERIFY]; kept as is)
        # (= sample_by_velocity(Yv[uid], M): the norms are taken row by row BEFORE the gather into sorted-unique order, so
        # the 8 M-row random gather moves one double per row instead of a whole row - 0.3 of the 0.86 s at 8 M cells)
        idx = _sample_by_norms(np.linalg.norm(Yv, axis=1)[uid], M)
    else:
        idx = np.random.RandomState(seed=seed).permutation(tmp_X.shape[0])
        idx = idx[range(M)]
    ctrl_pts = tmp_X[idx, :]
    if beta is None:
        h = bandwidth_selector(ctrl_pts, device)
        beta = 1 / h**2
    return valid_ind, Xv, Yv, idx, ctrl_pts, float(beta)


# =====================================================================================================================
# distributed helpers (one process per GPU; RCCL = torch.distributed "nccl" on ROCm; "gloo" in the CPU tests)
# =====================================================================================================================
def _dist_info(distributed, group):
    if not distributed:
        return 0, 1
    import torch.distributed as dist

    if not (dist.is_available() and dist.is_initialized()):
        raise RuntimeError("distributed=True but torch.distributed is not initialised (launch with torchrun).")
    return dist.get_rank(group), dist.get_world_size(group)


def shard_bounds(n: int, rank: int, world: int):
    """Contiguous block shard [lo, hi) of n cells for `rank` (the first n % world ranks get one extra cell)."""
    base, rem = divmod(n, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _current_device(device):
    """Context that makes `device` the thread's current GPU: torch's OBJECT collectives move their pickles through the
    current device under the RCCL backend, whatever device the caller's tensors live on."""
    import contextlib

    if device is None or not torch.cuda.is_available():
        return contextlib.nullcontext()
    dev = torch.device(device)
    return torch.cuda.device(dev) if dev.type == "cuda" else contextlib.nullcontext()


def _collective_device(group, device):
    """Where a tensor must live for a collective on `group`: the GPU under RCCL ("nccl"), the host under gloo."""
    import torch.distributed as dist

    if "nccl" in str(dist.get_backend(group)) and torch.cuda.is_available():
        return torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
    return torch.device("cpu")


def _gather_rows_np(arr, sizes, rank, world, group, device, to_all):
    """Concatenate the ranks' row blocks of a host (n_r, c) array (sizes[r] rows on rank r) with ONE padded tensor
    collective (all_gather if to_all, else gather to rank 0; other ranks get None)."""
    import torch.distributed as dist

    dev = _collective_device(group, device)
    mx = max(max(sizes), 1)
    pad = torch.zeros((mx, arr.shape[1]), dtype=torch.from_numpy(arr[:0]).dtype, device=dev)
    pad[: len(arr)] = torch.from_numpy(np.ascontiguousarray(arr)).to(dev)
    if to_all:
        outs = [torch.empty_like(pad) for _ in range(world)]
        dist.all_gather(outs, pad, group=group)
    else:
        outs = [torch.empty_like(pad) for _ in range(world)] if rank == 0 else None
        dst = dist.get_global_rank(group, 0) if group is not None else 0
        dist.gather(pad, outs, dst=dst, group=group)
        if rank != 0:
            return None
    return torch.cat([o[:sz] for o, sz in zip(outs, sizes)], dim=0).cpu().numpy()


def _consistent_K(k, ctrl, center, beta):
    """con_K(ctrl, ctrl) as float64, its values generated by the kernels' own kernel_value<cell dtype>."""
    cc = np.zeros((len(ctrl), 3), dtype=np.float32 if k.tdtype == torch.float32 else np.float64)
    cc[:, : ctrl.shape[1]] = ctrl - np.asarray(center)[None, : ctrl.shape[1]]
    cd = torch.from_numpy(cc).to(k.device)
    return k.con_k(cd, cd, beta, dtype=k.dtype_name).to(torch.float64)


# =====================================================================================================================
# the engine
# =====================================================================================================================
class SparseVFCEngine:
    """Device-resident SparseVFC EM loop over this rank's block of cells.

    X, Y: this rank's cells (n_local x D, n_local x Dy, host float64, finite rows only).  ctrl (M x D) and beta are
    identical on every rank.  ``n_total`` = global number of cells (for gamma and the initial sigma^2).
    """

    # which truncated minimum-norm solver answers once the system is rank deficient: None = by M ("deflated" from
    # DEFLATED_MIN_M control points on, else "full"), or "deflated" | "lowrank" | "full" for every engine built afterwards
    minnorm_method = None
    wide_y = True         # Dy > 3 on a cached U: the MFMA kernels of mvf_wide.hip (False: one VALU pass per three columns)
    async_direct = True   # M <= 640 steady state: mvf_solve_minnorm_lrd_async + speculative field update (False: round 5's calls)

    def __init__(self, X, Y, ctrl, beta, *, dtype=None, device=None, distributed=False, group=None, n_total=None,
                 kernels=None, cache_u="auto", shard_sizes=None, gram_mode="full", force_collectives=False,
                 collective="torch"):
        dtype = dtype or _DEFAULT_DTYPE
        X = np.asarray(X, dtype=np.float64)
        Y = np.asarray(Y, dtype=np.float64)
        ctrl = np.asarray(ctrl, dtype=np.float64)
        if X.ndim != 2 or Y.ndim != 2 or len(X) != len(Y):
            raise ValueError("X and Y must be 2-D with the same number of rows")
        self.D, self.Dy = X.shape[1], Y.shape[1]
        if not (1 <= self.D <= 3):
            raise NotImplementedError(f"the HIP path supports 1-3 spatial dimensions, got {self.D}")
        if self.Dy < 1:
            raise ValueError("Y must have at least one column")
        # The kernels are 3 columns wide; a wider Y (kernel_interpolation: Dy = #keys) is processed as column groups
        # that share ONE Gram matrix per EM step (G does not depend on Y) and get their own rhs / solve / apply.
        self.ng = (self.Dy + 2) // 3
        clear_eval_cache()  # evaluator results of an earlier call must not shrink the HBM this fit plans with
        self.k = kernels if kernels is not None else _make_kernels(device, dtype)
        self.distributed = bool(distributed)
        self.group = group
        self.rank, self.world = _dist_info(distributed, group)
        # `multi`: the step runs the multi-rank protocol (split Gram stages, packed-triangle all-reduce, [R | stats] and the
        # closing 14-double collective).  force_collectives=True takes it with ONE rank too: every collective then executes
        # on the real backend (RCCL on a one-GPU box) and, a single rank's sum being the identity, the fit stays bit-equal
        # to the plain single-process one - how the exchange is tested and timed without a second GPU.
        # collective: "torch" = torch.distributed all_reduce (RCCL under the "nccl" backend, gloo in the CPU tests);
        # "mvf" = mvf_allreduce_stats through the C ABI on this engine's own RCCL communicator (_comm.MvfComm)
        if collective not in ("torch", "mvf"):
            raise ValueError("collective must be 'torch' or 'mvf'")
        if force_collectives and collective == "torch" and not self.distributed:
            raise ValueError("force_collectives with collective='torch' needs distributed=True and an initialised process "
                             "group (world size 1 is fine); collective='mvf' creates its own single-rank communicator")
        self.force_collectives = bool(force_collectives)
        self.multi = self.world > 1 or self.force_collectives
        self.collective = collective
        self.comm = None
        self._comm_stream = None
        self.n_local = len(X)
        self.n_total = int(n_total) if n_total is not None else self.n_local
        # rows per rank (block shards by default; the caller states them when it brings its own uneven shards)
        self.shard_sizes = [int(v) for v in shard_sizes] if shard_sizes is not None else \
            [hi - lo for lo, hi in (shard_bounds(self.n_total, r, self.world) for r in range(self.world))]
        if len(self.shard_sizes) != self.world or self.shard_sizes[self.rank] != self.n_local or \
                sum(self.shard_sizes) != self.n_total:
            raise ValueError("shard sizes do not match the rows this rank holds / the global cell count")
        self.M = len(ctrl)
        self.beta = float(beta)
        self.ctrl = ctrl
        # the kernel is translation invariant: centre on the control points so float32 keeps its bits for geometry
        self.center = ctrl.mean(0) if self.M else np.zeros(self.D)

        k = self.k
        if self.multi and collective == "mvf":
            from ._comm import MvfComm

            self.comm = MvfComm(k.device, self.rank, self.world, group)
            self._comm_stream = torch.cuda.Stream(device=k.device)
        self.x4 = k.to_x4(X, self.center)
        self.y4 = [k.to_x4(Y[:, 3 * g : 3 * g + 3]) for g in range(self.ng)]
        f64 = torch.float64
        ng = self.ng
        # (rounds 4 - 5 had gram_mode="pivot" here: the rest of a fit on the control points the pivoted factorisation selected.
        # It was a different truncation of the ill-posed system - sigma^2 at 1.4 - 8.5 x the reference's own floor - and failed
        # this repository's 1.25 x criterion on three fixtures: removed in round 6, HISTORY.md section "pivot mode")
        if gram_mode != "full":
            raise ValueError("gram_mode: only 'full' (the reference's M-step on all M control points) exists")
        self.gram_mode = gram_mode
        self.comm_events = None  # bench.py: list of (start, end) events around the collectives
        self._cache_u_wanted = cache_u
        self._setup_control_points(ctrl)
        M = self.M
        self.quad = k.zeros(ng, dtype=f64)
        # the step's LAST collective: [sum P r | failed | solver signature (6) | its squares (6)], summed over the ranks
        self.fin = k.zeros(14, dtype=f64)
        self.spr = self.fin[:1]
        self.info = k.zeros(1, dtype=torch.int32)
        self.P = torch.ones(self.n_local, dtype=k.tdtype, device=k.device)
        self.r = None
        # U = con_K(X, ctrl) is constant across EM iterations: cache its values (cell dtype) for the Gram kernel when HBM
        # has room ("auto": sizeof(dtype) n M bytes plus headroom), else the Gram kernel regenerates them every iteration
        self._build_u_cache()
        # Wide Y (Dy > 3: kernel_interpolation's genes) on a cached U: R = U^T P Y and V = U C as MFMA products that stream
        # the cache ONCE for all columns (mvf_rhs_cached / mvf_apply_cached) instead of one regenerating VALU pass per
        # group of three columns.  Dense row-major Y / V (padded to 16 columns, Y also to the cache's padded cell count).
        self.wide = bool(self.wide_y and self.cached_u and ng >= 2 and hasattr(k, "rhs_wide") and self.n_local and M)
        if self.wide:
            n_pad, m_pad = k.wide_pads(self.n_local, M)
            Dp = -(-self.Dy // 16) * 16
            self.Yd = k.zeros(n_pad, Dp)
            self.Yd[: self.n_local, : self.Dy] = torch.from_numpy(np.ascontiguousarray(Y, dtype=np.float64)).to(k.device).to(k.tdtype)
            self.Vd = k.zeros(self.n_local, Dp)
            self.Rd = k.zeros(M, self.Dy, dtype=f64)
            self.Cd = k.zeros(m_pad, Dp, dtype=f64)
            self._rbuf = k.zeros(self.n_local)
            self.V4 = None
        else:
            self.V4 = [k.zeros(self.n_local, 4) for _ in range(ng)]
        # Coefficient solve (lstsq_method "scipy" = the reference's gelsd semantics): Cholesky with NO regularisation
        # while the pivots certify full numerical rank (then nothing is truncated and it IS the gelsd solution), else
        # the truncated minimum-norm solve (mvf_solve_minnorm).  Rank deficiency is sticky within a fit: sigma^2 only
        # shrinks, so lambda sigma^2 K never comes back.
        self.lstsq_method = "scipy"
        self.rank_deficient = False
        self.basis_valid = False
        self.mn_shift = 2.0 ** -36       # Cholesky shift of the eigensolver (relative to mean(diag); subtracted again)
        self.pivot_ratio = 2.0 ** -40    # full rank is certified when min L_jj^2 > pivot_ratio * max L_jj^2
        self.solver_stats = {"cholesky": 0, "minnorm": 0, "sweeps": [], "rank": []}
        self.pivots = k.zeros(2, dtype=f64)
        self.einfo = k.zeros(12, dtype=f64)
        self.basis, self.basis_valid, self.warm_start = None, False, True
        # "deflated": pivoted-Cholesky factor, then only the invariant subspace below the cut-off (block inverse iteration on
        # 256 vectors) computed and projected out (mvf_solve_minnorm_lrd; 9 ms where the next one takes 23);
        # "lowrank": the same factor + Jacobi on all its r columns (mvf_solve_minnorm_lr); "full": Jacobi on all M
        # columns of the shifted factor, warm-started (mvf_solve_minnorm)
        # measured per solve in the EM's steady state (ms, lowrank / full): M = 500: 8.7 / 3.6, 1000: 17.3 / 18.4,
        # 1500: 20.9 / 33, 2000: 21.6 / 54, 3000: 23.6 / 113 - the full-width warm start wins while the factor keeps
        # nearly every column
        # round 4, deflated / full (ms): M = 640: 4.6 / 6.5, 768: 5.0 / 10.3, 896: 7.5 / 14.5, 1000: 7.5 / 17.5
        # round 5: factors of 128 .. 511 columns take a 64-vector block, and when the previous iteration's factor kept ALL M
        # columns (M <= 640: BASELINE configs 2 and 5) the call takes its direct form - one Cholesky of the permuted matrix with
        # the inverse factor riding along, block inverse iteration, a one-launch warm-started 64 x 64 Rayleigh-Ritz:
        # M = 500: 1.5 / 3.6 ms (profiles/r05_small_m_probe.json); below 256 control points the full-width solve stays
        self.mn_method = self.minnorm_method or ("deflated" if self.M >= DEFLATED_MIN_M else "full")
        self.rank_hint = 0
        self._lrd_form = 0
        # lstsq_method="cholesky" (extension, not a reference mode): jitter-escalated Cholesky, the round-1 solver
        self.jitter = 0.0
        self.jitter_first = 1e-15
        self.jitter_max = 1e-3
        self.solve_retries = 0
        self.E = 1.0
        self.tecr = 1.0
        self.iteration = 0

    def _setup_control_points(self, ctrl):
        """Everything whose shape follows the control points: ctrl4, K, G, the all-reduce buffer, the coefficients."""
        k, f64, ng = self.k, torch.float64, self.ng
        self.M = M = len(ctrl)
        self.ctrl = ctrl
        self.ctrl4 = k.to_x4(ctrl, self.center)
        # K = con_K(ctrl, ctrl) (regulariser + energy), stored float64 but GENERATED IN THE CELL DTYPE: K must be the
        # same function of the control points as U is of the cells.  With float32 kernel values in U and an exact float64
        # K the null spaces of U^T P U and of lambda sigma^2 K no longer line up and the field moves by 1e-3 (M = 2000,
        # lambda = 3); generated consistently it moves by 1.4e-5, the float32 rounding level (measured on the oracle).
        self.K = _consistent_K(k, ctrl, self.center, self.beta)
        # one contiguous float64 buffer for the all-reduces of an EM step: [packed upper triangle of G (M (M + 1) / 2;
        # only when there is more than one rank) | R_g (M * 3) per column group | stats (5)]
        self.G = k.zeros(M, M, dtype=f64)
        ntri = M * (M + 1) // 2 if self.multi else 0
        self.red = k.zeros(ntri + 3 * M * ng + 5, dtype=f64)
        self.tri = self.red[:ntri]
        self.R = [self.red[ntri + 3 * M * g : ntri + 3 * M * (g + 1)].view(M, 3) for g in range(ng)]
        self.st = self.red[ntri + 3 * M * ng :]
        self.C = [k.zeros(M, 3, dtype=f64) for _ in range(ng)]
        self.C_new = [k.zeros(M, 3, dtype=f64) for _ in range(ng)]
        self._probes = None

    def _build_u_cache(self):
        """U = con_K(X, ctrl) is constant across EM iterations: cache its values (cell dtype) for the Gram kernel when HBM
        has room ("auto": sizeof(dtype) n M bytes plus headroom), else the Gram kernel regenerates them every iteration."""
        k, cache_u = self.k, self._cache_u_wanted
        self.cached_u = False
        if cache_u and hasattr(k, "build_ublk") and self.n_local and self.M:
            if hasattr(k, "drop_ublk"):
                k.drop_ublk()
            need = k.ublk_bytes(self.n_local, self.M)
            free = torch.cuda.mem_get_info(k.device)[0] if cache_u == "auto" else None
            # "auto": the cache must fit with 8 GB of headroom.  Measured at the largest case (float64 cells, 8 M x 3000
            # on ONE GPU = 197 GB): streaming the cache runs the Gram kernel at 47 TF, regenerating the operands
            # (software float64 exp) at 38 TF; the float32 cache (98 GB) and the per-rank caches run at 60 / 53 TF.
            if free is None or need + (8 << 30) < free:
                k.build_ublk(self.x4, self.ctrl4, self.beta)
                self.cached_u = True

    # ------------------------------------------------------------------ collectives
    def _all_reduce(self, t, op="sum", wait=True):
        """All-reduce `t` in place over the ranks.  wait=False: returns a handle for `_wait` - the collective runs on the
        backend's own stream (RCCL) / thread (gloo) while this rank keeps enqueuing kernels that do not touch `t`."""
        if not self.multi:
            return None
        ev = None
        if self.comm_events is not None:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        nbytes = t.numel() * t.element_size()
        if self.comm is not None:
            # C-ABI path: ncclAllReduce enqueued by mvf_allreduce_stats.  Synchronous form: on the compute stream itself
            # (stream order IS the dependency).  Asynchronous form: on the engine's communication stream, fenced by events
            # on both sides, so the rhs / quadform kernels enqueued next run beside it.
            cur = torch.cuda.current_stream(self.k.device)
            if wait:
                self.comm.all_reduce(t, op, cur)
                self._wait((None, ev, nbytes))
                return None
            ready = torch.cuda.Event()
            ready.record(cur)
            self._comm_stream.wait_event(ready)
            self.comm.all_reduce(t, op, self._comm_stream)
            done = torch.cuda.Event()
            done.record(self._comm_stream)
            return (done, ev, nbytes)
        import torch.distributed as dist

        work = dist.all_reduce(t, op=dist.ReduceOp.SUM if op == "sum" else dist.ReduceOp.MIN, group=self.group,
                               async_op=not wait)
        handle = (work, ev, nbytes)
        if wait:
            self._wait((None,) + handle[1:])
            return None
        return handle

    def _wait(self, handle):
        if handle is None:
            return
        work, ev, nbytes = handle
        if isinstance(work, torch.cuda.Event):
            torch.cuda.current_stream(self.k.device).wait_event(work)  # C-ABI path: the compute stream waits for the comm stream
        elif work is not None:
            work.wait()  # RCCL: the current stream waits for the collective; gloo: the host does
        if ev is not None:
            ev[1].record()
            self.comm_events.append(ev + (nbytes,))

    # ------------------------------------------------------------------ EM
    def init_state(self, gamma=0.9):
        """V = 0, C = 0, sigma^2 = sum ||Y||^2 / (N Dy)  (Appendix A step 4)."""
        k = self.k
        self.spr.zero_()
        self.P.fill_(1.0)  # sigma^2_0 = sum ||Y||^2 / (N Dy): unit weights (a previous fit of this engine left its posterior)
        empty_ctrl = self.ctrl4[:0]
        for g in range(self.ng):
            self.C[g].zero_()
        self._apply_all(empty_ctrl)
        self._all_reduce(self.spr)
        s2 = float(self.spr.cpu()[0]) / (self.n_total * self.Dy)
        self.sigma2 = 1e-7 if s2 < 1e-8 else s2
        self.gamma = float(gamma)
        self.E, self.tecr, self.iteration = 1.0, 1.0, 0
        self.rank_deficient = False
        self.basis_valid = False
        self.rank_hint = 0
        self._lrd_form = 0
        self._lr_ran, self._spr_spec = False, None

    def _apply_all(self, ctrl4, C=None):
        """V_g = U C_g for every column group; r = sum_g ||Y_g - V_g||^2; spr += sum P r."""
        k = self.k
        C = self.C if C is None else C
        if self.wide:
            # one pass over the cached U for all Dy columns (an empty ctrl4 - init_state's V = 0 - is zero coefficients)
            if ctrl4.shape[0] == 0:
                self.Cd.zero_()
            else:
                self.Cd[: self.M, : 3 * self.ng] = torch.cat(list(C), dim=1)
            k.apply_wide(self.Cd, self.Dy, self.M, self.Yd, self.P, self.Vd, self._rbuf, self.spr)
            self.r = self._rbuf
            return
        for g in range(self.ng):
            self.V4[g], rg = k.apply(self.x4, ctrl4, self.beta, C[g], self.y4[g], self.P, self.spr)
            if g == 0:
                self.r = rg
            else:
                self.r += rg

    def em_step(self, *, a=5.0, lambda_=3.0, minP=1e-5, theta=0.75):
        """One EM iteration (Appendix A step 5 a-e).  Returns (E, tecr).

        Collectives per step (multi-rank), four: the 8-byte MIN of the E-step's global min-non-zero rule; THE all-reduce of
        the packed upper triangle of G, issued the moment this rank's G is final and overlapped with the rhs / quadform
        kernels; the small [R | stats] all-reduce behind them; and one 14-double SUM at the end of the step that carries
        sum P r together with every rank's failure flag and solver signature (`_finish_step`).  Host round trips: one after
        the solve (its status / pivots, the statistics, the energy - every control-flow decision is taken from
        all-reduced or replicated deterministic values, so all ranks decide alike) and one for sigma^2 + the agreement
        check; the minimum-norm solve adds its own (one per Jacobi sweep)."""
        k = self.k
        # ---- E-step: dynamo's `t1[t1 == 0] = min(t1[t1 != 0])` needs the GLOBAL min-non-zero t1; phase 1 leaves it in
        # device memory, phase 2 reads it from there
        mins = k.estep_min(self.r, self.sigma2)
        fill = mins[:1]
        self._all_reduce(fill, "min")
        self.st.zero_()
        k.estep_p(self.r, self.sigma2, self.gamma, a, self.Dy, minP, theta, fill, self.P, self.st)
        # ---- M-step assembly (MFMA Gram + rhs), energy regulariser with the OLD coefficients, the collectives
        if self.wide and not self.multi:
            k.gram(self.x4, self.P, None, self.ctrl4, self.beta, self.G, None, tiles_only=True)
        elif not self.multi:
            k.gram(self.x4, self.P, self.y4[0], self.ctrl4, self.beta, self.G, self.R[0])
        else:
            # G first: THE all-reduce of the step (packed upper triangle, 36 MB at M = 3000) starts the moment this rank's
            # G is final and runs while the rhs / quadform kernels below execute; their small results [R | stats] follow
            k.gram(self.x4, self.P, None, self.ctrl4, self.beta, self.G, None, tiles_only=True)
            k.sym_pack(self.G, self.tri)
            big = self._all_reduce(self.tri, wait=False)
            if not self.wide:
                k.gram(self.x4, self.P, self.y4[0], self.ctrl4, self.beta, self.G, self.R[0], rhs_only=True)
        if self.wide:
            k.rhs_wide(self.P, self.Yd, self.Dy, self.M, self.Rd)
            for g in range(self.ng):  # (the solver and the all-reduce buffer keep their three-column groups)
                w = min(3, self.Dy - 3 * g)
                self.R[g][:, :w].copy_(self.Rd[:, 3 * g : 3 * g + w])
        for g in range(1, self.ng if not self.wide else 0):
            k.gram(self.x4, self.P, self.y4[g], self.ctrl4, self.beta, self.G, self.R[g], rhs_only=True)
        for g in range(self.ng):
            k.quadform(self.K, self.C[g], self.quad[g : g + 1])
        if self.multi:
            self._all_reduce(self.red[self.tri.numel():])
            self._wait(big)
            k.sym_unpack(self.tri, self.G)
        host = self._solve_all(lambda_ * self.sigma2)
        spec = self._spr_spec   # not None: the Cholesky branch already applied the new coefficients (see _solve_all_local)
        if spec is None:
            self.fin.zero_()
        if host is not None:
            s_pr, s_p, s_pf, s_cnt = (float(host[i]) for i in range(4))
            quad = float(sum(host[5:]))
            E_old = self.E
            E = s_pr / (2 * self.sigma2) + s_p * math.log(self.sigma2) * self.Dy / 2 + lambda_ / 2 * quad
            self.tecr = abs((E - E_old) / E)
            self.E = E
            self.C, self.C_new = self.C_new, self.C
            # ---- field + sigma^2 + gamma
            if spec is None:
                self._apply_all(self.ctrl4)
        spr = spec if spec is not None else self._finish_step()
        self.sigma2 = spr / (s_pf * self.Dy)
        g = s_cnt / self.n_total
        self.gamma = 0.95 if g > 0.95 else (0.05 if g < 0.05 else g)
        self.iteration += 1
        return self.E, self.tecr

    def _rhs_batches(self):
        """Column groups are solved two at a time (mvf_solve / mvf_solve_minnorm take up to 8 right-hand sides)."""
        return [list(range(g0, min(g0 + 2, self.ng))) for g0 in range(0, self.ng, 2)]

    def _solve_batch(self, gs, fn):
        """fn(R, C_out) for the concatenated right-hand sides of the column groups `gs`; scatters C back."""
        if len(gs) == 1:
            fn(self.R[gs[0]], self.C_new[gs[0]])
            return
        Rcat = torch.cat([self.R[g] for g in gs], dim=1).contiguous()
        Ccat = torch.empty_like(Rcat)
        fn(Rcat, Ccat)
        for j, g in enumerate(gs):
            self.C_new[g].copy_(Ccat[:, 3 * j : 3 * j + 3])

    def _probe_vectors(self):
        """Two fixed +-1 vectors (M x 2, float64, identical on every rank): the right-hand sides of the inverse-iteration
        witness of the full-rank certificate."""
        if getattr(self, "_probes", None) is None:
            j = np.arange(self.M, dtype=np.uint64)
            bits = [((j * np.uint64(2654435761) + np.uint64(s_)) >> np.uint64(15)) & np.uint64(1) for s_ in (12345, 987654321)]
            b = np.stack([1.0 - 2.0 * x.astype(np.float64) for x in bits], axis=1)
            self._probes = torch.from_numpy(np.ascontiguousarray(b)).to(self.k.device)
        return self._probes

    @staticmethod
    def _check_converged(sweeps):
        """mvf_solve_minnorm(_lr) report a sweep count of x.5 when the Jacobi iteration hit its sweep cap before a clean
        sweep: the factor is then not orthogonal and the truncated back-solve is not an eigen-solve - fail loudly."""
        if float(sweeps) % 1.0 != 0.0:
            raise _lib.MVFError(f"coefficient solve failed: the Jacobi eigensolver did not converge in {int(sweeps)} "
                                f"sweeps (non-finite or wildly scaled Gram system?)")

    def _host_stats(self, *extra):
        """ONE device -> host copy: [extra ... | stats (5) | quad per column group] as float64."""
        h = torch.cat([t.to(torch.float64).reshape(-1) for t in extra] + [self.st, self.quad]).cpu()
        return h

    def _solve_all(self, ls2):
        """C_new = lstsq(G + ls2 K, R) for every column group, with the semantics `self.lstsq_method` names.
        Returns the host copy of [stats (5) | quad per group] (read in the same round trip as the solve's status).

        Multi-rank: every rank solves the same all-reduced system redundantly and must take the same branch (Cholesky /
        rank-revealing / full-width, retries, sweeps) - the kernels are deterministic, so they do.  That is VERIFIED every
        step (`_finish_step`): the signature of this rank's solver decisions, or its failure, travels in the step's last
        collective; here a failure is only recorded (returns None) so that this rank still takes part in it."""
        self._step_error, self._solver_signature, self._lr_ran, self._spr_spec = None, (0.0,) * 6, False, None
        if not self.multi:
            return self._solve_all_local(ls2)
        try:
            return self._solve_all_local(ls2)
        except Exception as exc:  # noqa: BLE001 - ANY failure must reach the collective, or the other ranks hang in it
            self._step_error = exc
            return None

    def _finish_step(self):
        """sum P r over all ranks (host float).  Multi-rank: ONE 14-double SUM all-reduce carries it together with every
        rank's failure flag and solver signature `sig = [branch, status, sweeps, kept rank, factor rank, retries]` as
        (sig, sig^2): all ranks hold the same signature iff  world * sum(sig^2) == sum(sig)^2  for every entry (the
        entries are small integers or halves: the sums are exact).  A failure on any rank, or a disagreement, raises on
        EVERY rank in this very step instead of leaving the others hanging in the next collective."""
        if not self.multi:
            return float(self.spr.cpu()[0])
        sig = [float(x) for x in self._solver_signature]
        err = self._step_error
        tail = [1.0 if err is not None else 0.0] + sig + [x * x for x in sig]
        self.fin[1:].copy_(torch.tensor(tail, dtype=torch.float64))
        self._all_reduce(self.fin)
        h = self.fin.cpu().numpy()
        if err is not None:
            raise err
        if h[1] != 0.0:
            raise _lib.MVFError(f"SparseVFC (rank {self.rank}): the coefficient solve failed on another rank")
        s1, s2 = h[2:8], h[8:14]
        if not np.array_equal(self.world * s2, s1 * s1):
            raise _lib.MVFError(
                f"SparseVFC (rank {self.rank}): ranks disagree on the coefficient solve's decisions "
                f"[branch, status, sweeps, kept rank, factor rank, retries]: mine {sig}, mean over the ranks "
                f"{(s1 / self.world).tolist()} - the all-reduced Gram systems are not identical")
        return float(h[0])

    def _solve_all_local(self, ls2):
        k = self.k
        self._solver_signature = (0.0, 0.0, 0.0, 0.0, 0.0, 0.0)
        batches = self._rhs_batches()
        if self.lstsq_method == "cholesky":
            while True:
                for gs in batches:
                    self._solve_batch(gs, lambda R, C: k.solve(self.G, self.K, ls2, self.jitter, R, C, self.info))
                h = self._host_stats(self.info)
                fail = int(h[0])
                if fail == 0:
                    self.solver_stats["cholesky"] += 1
                    self._solver_signature = (1.0, 0.0, 0.0, float(self.M), 0.0, float(self.solve_retries))
                    return h[1:]
                self.solve_retries += 1
                self.jitter = max(self.jitter * 10.0, self.jitter_first)
                if self.jitter > self.jitter_max:
                    raise _lib.MVFError(
                        f"coefficient solve failed: non-positive pivot at {fail - 1} even with jitter "
                        f"{self.jitter:g}; the system is not numerically PSD (NaN/Inf in the inputs?)")
        if not self.rank_deficient:
            # Un-regularised Cholesky for every column group.  Full numerical rank (nothing for gelsd to truncate) is
            # certified by TWO witnesses: (1) the pivot ratio min L_jj^2 > 2^-40 max L_jj^2 - but min L_jj^2 only bounds
            # lambda_min from ABOVE, a Kahan-type matrix keeps large pivots over a tiny lambda_min; so (2) one step of inverse
            # iteration from two fixed +-1 vectors, free of charge as two more right-hand sides of the first batch's
            # factorisation: ||b|| / ||A^-1 b|| lies in [lambda_min, ~sqrt(M) lambda_min] and must clear
            # 8 sqrt(M) eps x (M max L_jj^2 >= trace-scale bound of lambda_max).  Either witness failing sends this and every
            # later step of the fit to the truncated solve (where nothing is truncated the two solves coincide).
            probes = self._probe_vectors()
            Z = None
            for i, gs in enumerate(batches):
                if i == 0:
                    Rcat = torch.cat([self.R[g] for g in gs] + [probes], dim=1).contiguous()
                    Ccat = torch.empty_like(Rcat)
                    k.solve(self.G, self.K, ls2, 0.0, Rcat, Ccat, self.info, self.pivots)
                    for j, g in enumerate(gs):
                        self.C_new[g].copy_(Ccat[:, 3 * j : 3 * j + 3])
                    Z = Ccat[:, 3 * len(gs):]
                else:
                    self._solve_batch(gs, lambda R, C: k.solve(self.G, self.K, ls2, 0.0, R, C, self.info, None))
            # Single rank: the field update with the NEW coefficients is enqueued speculatively, so that its sum P r comes back
            # in the same device -> host copy as the certificate - one host round trip per EM iteration instead of two in the
            # regime of Spateo's stock call (M = 100: full rank certified in every iteration).  If the certificate fails the
            # truncated solve below recomputes C_new and em_step applies it again.
            spec = not self.multi
            if spec:
                self.fin.zero_()
                self._apply_all(self.ctrl4, self.C_new)
            h = self._host_stats(self.info, self.pivots, Z, *([self.spr] if spec else []))
            nz = Z.numel()
            z = h[3 : 3 + nz].numpy().reshape(self.M, -1)
            with np.errstate(all="ignore"):
                lam_hat = float(np.sqrt(self.M) / np.sqrt((z * z).sum(0)).max())   # min over the probes of ||b|| / ||z||
            certified = (int(h[0]) == 0 and float(h[1]) > self.pivot_ratio * float(h[2]) and np.isfinite(lam_hat) and
                         lam_hat > 8.0 * np.sqrt(self.M) * np.finfo(np.float64).eps * self.M * float(h[2]))
            if certified:
                self.solver_stats["cholesky"] += 1
                self._solver_signature = (2.0, 0.0, 0.0, float(self.M), 0.0, 0.0)
                if spec:
                    self._spr_spec = float(h[3 + nz])
                    return h[3 + nz + 1:]
                return h[3 + nz:]
            self.rank_deficient = True
        # truncated minimum-norm solve (gelsd cut-off eps * max|lambda|)
        if self.mn_method in ("lowrank", "deflated") and hasattr(k, "solve_minnorm_lr"):
            dfl = {"deflate": True} if self.mn_method == "deflated" else {}
            # M <= 640 in its steady state (the previous iteration kept all M columns): the direct form WITHOUT a host round
            # trip inside the solve (mvf_solve_minnorm_lrd_async; the acceptance test runs on the device), and - single rank -
            # the field update with the new coefficients enqueued behind it speculatively, so that status, statistics and
            # sum P r come back in ONE device -> host copy per EM iteration (round 5: three reads inside the solve, one after
            # it, one for sum P r).  Not accepted (einfo[9]): the synchronous call below answers through the factor form.
            # Only behind an ACCEPTED direct-form call (einfo[8] == 2): the first attempt of a fit, and the first after a
            # rejected one, go through the synchronous entry point, which knows how to help the power iteration along while
            # the matrix still moves and keeps the cool-down after a failure.
            if (dfl and self.async_direct and len(batches) == 1 and self._lrd_form == 2 and self.rank_hint == self.M
                    and 128 <= self.M <= 640 and hasattr(k, "solve_minnorm_lrd_async")):
                self._solve_batch(batches[0], lambda R, C: k.solve_minnorm_lrd_async(self.G, self.K, ls2, R, C, self.info,
                                                                                     self.einfo, self._lrd_form))
                spec = not self.multi
                if spec:
                    self.fin.zero_()
                    self._apply_all(self.ctrl4, self.C_new)
                h = self._host_stats(self.info, self.einfo, *([self.spr] if spec else []))
                if int(h[0]) != 0:
                    raise _lib.MVFError("coefficient solve failed: G + lambda sigma^2 K has non-finite entries")
                if float(h[1 + 9]) == 0.0:
                    self._lrd_form = 2
                    self.solver_stats["minnorm"] += 1
                    self.solver_stats["async"] = self.solver_stats.get("async", 0) + 1
                    self.solver_stats["sweeps"].append(float(h[1]))
                    self.solver_stats["rank"].append(int(h[2]))
                    self.solver_stats.setdefault("factor_rank", []).append(self.rank_hint)
                    self.solver_stats.setdefault("block", []).append(int(h[1 + 7]))
                    self._lr_ran = True
                    self._solver_signature = (3.0, 0.0, float(h[1]), float(h[2]), float(self.rank_hint), float(h[1 + 7]))
                    if spec:
                        self._spr_spec = float(h[13])
                        return h[14:]
                    return h[13:]
                self._lrd_form = 0  # (the repeat below re-derives it)
            # rank-revealing factor (pivoted Cholesky) + Jacobi on the kept columns only; the previous iteration's factor
            # rank tells how many pivot steps to enqueue before the first status read
            self._solve_batch(batches[0], lambda R, C: k.solve_minnorm_lr(self.G, self.K, ls2, R, C, self.info, self.einfo,
                                                                          rank_hint=self.rank_hint, **dfl))
            h = self._host_stats(self.info, self.einfo)
            if int(h[0]) != 0:
                raise _lib.MVFError("coefficient solve failed: G + lambda sigma^2 K has non-finite entries")
            self._check_converged(h[1])
            self.rank_hint = int(h[1 + 6])
            self._lrd_form = int(h[1 + 8]) if dfl else 0   # 1 factor form / 2 direct form: what the next call may continue
            for gs in batches[1:]:
                self._solve_batch(gs, lambda R, C: k.solve_minnorm_lr(self.G, self.K, ls2, R, C, self.info, self.einfo,
                                                                      reuse=True, **dfl))
            self.solver_stats["minnorm"] += 1
            self.solver_stats["sweeps"].append(float(h[1]))
            self.solver_stats["rank"].append(int(h[2]))
            self.solver_stats.setdefault("factor_rank", []).append(self.rank_hint)
            if dfl:
                self.solver_stats.setdefault("block", []).append(int(h[1 + 7]))  # 256 / 128; 0 = the Jacobi form answered
            self._lr_ran = True
            # (the block size - 0 when the Jacobi path answered - is part of what the ranks must agree on: ADVICE r5)
            self._solver_signature = (3.0, 0.0, float(h[1]), float(h[2]), float(self.rank_hint), float(h[1 + 7]) if dfl else 0.0)
            return h[1 + 12:]
        # mn_method = "full": Jacobi on all M columns of the shifted Cholesky factor; the shift only has to make the
        # factorisation inside the eigensolver exist, it is subtracted from the eigenvalues again.
        # warm start: the previous EM iteration's eigenvectors pre-diagonalise this iteration's matrix
        if self.basis is None and hasattr(k, "minnorm_basis"):
            self.basis = k.minnorm_basis(self.M)
        while True:
            self._solve_batch(batches[0], lambda R, C: k.solve_minnorm(self.G, self.K, ls2, self.mn_shift, R, C,
                                                                       self.info, self.einfo, basis=self.basis,
                                                                       warm=self.basis_valid))
            h = self._host_stats(self.info, self.einfo)
            if int(h[0]) == 0:
                self._check_converged(h[1])
                self.basis_valid = self.basis is not None and self.warm_start
                break
            self.basis_valid = False
            self.mn_shift *= 16.0
            if self.mn_shift > 2.0 ** -12:
                raise _lib.MVFError("coefficient solve failed: G + lambda sigma^2 K is not numerically positive "
                                    "semi-definite (NaN/Inf in the inputs?)")
        for gs in batches[1:]:
            self._solve_batch(gs, lambda R, C: k.solve_minnorm(self.G, self.K, ls2, self.mn_shift, R, C, self.info,
                                                               self.einfo, reuse=True))
        self.solver_stats["minnorm"] += 1
        self.solver_stats["sweeps"].append(float(h[1]))
        self.solver_stats["rank"].append(int(h[2]))
        self._solver_signature = (4.0, 0.0, float(h[1]), float(h[2]), 0.0, float(np.log2(self.mn_shift)))
        return h[1 + 12:]

    def fit(self, *, a=5, gamma=0.9, lambda_=3, minP=1e-5, MaxIter=500, theta=0.75, ecr=1e-5, lstsq_method="scipy"):
        self.lstsq_method = _check_lstsq_method(lstsq_method)
        self.init_state(gamma)
        tecr_vec, E_vec = [], []
        while self.iteration < MaxIter and self.tecr > ecr and self.sigma2 > 1e-8:
            E, tecr = self.em_step(a=a, lambda_=lambda_, minP=minP, theta=theta)
            E_vec.append(E)
            tecr_vec.append(tecr)
        return np.asarray(tecr_vec), np.asarray(E_vec)

    # ------------------------------------------------------------------ outputs
    def predict(self, pts):
        """v(pts) = con_K(pts, ctrl, beta) @ C on the device -> host float64 (n, Dy)."""
        pts = np.asarray(pts, dtype=np.float64)
        p4 = self.k.to_x4(pts, self.center)
        cols = [self.k.apply(p4, self.ctrl4, self.beta, self.C[g])[0][:, :3] for g in range(self.ng)]
        return torch.cat(cols, dim=1)[:, : self.Dy].to(torch.float64).cpu().numpy()

    def _gather_rows(self, t, root_only):
        """Concatenate the per-rank row blocks: on every rank, or (root_only) on rank 0 - the others keep their own."""
        if self.world == 1 and not (self.force_collectives and self.distributed):
            return t
        import torch.distributed as dist

        mx = max(self.shard_sizes)
        pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=t.device)
        pad[: t.shape[0]] = t
        if root_only:
            outs = [torch.empty_like(pad) for _ in range(self.world)] if self.rank == 0 else None
            dst = dist.get_global_rank(self.group, 0) if self.group is not None else 0
            dist.gather(pad, outs, dst=dst, group=self.group)
            if self.rank != 0:
                return t
        else:
            outs = [torch.empty_like(pad) for _ in range(self.world)]
            dist.all_gather(outs, pad, group=self.group)
        return torch.cat([o[:sz] for o, sz in zip(outs, self.shard_sizes)], dim=0)

    def results(self, gather="root"):
        """(V (N, Dy), P (N, 1), C (M, Dy)) as host float64.  Multi-rank: ``gather="root"`` (default) collects the
        cell rows on rank 0 only - the other ranks get their own rows back; ``"all"`` gives every rank all rows."""
        if gather not in ("root", "all"):
            raise ValueError("gather must be 'root' or 'all'")
        Vloc = (self.Vd[:, : self.Dy] if self.wide else torch.cat([v[:, :3] for v in self.V4], dim=1)[:, : self.Dy]).contiguous()
        V = self._gather_rows(Vloc, gather == "root").to(torch.float64).cpu().numpy()
        P = self._gather_rows(self.P[:, None].contiguous(), gather == "root").to(torch.float64).cpu().numpy()
        C = torch.cat(self.C, dim=1)[:, : self.Dy].cpu().numpy().copy()
        return V, P, C


# =====================================================================================================================
# drop-in functions
# =====================================================================================================================
def con_K(x, y, beta: float = 0.1, method: str = "cdist", return_d: bool = False, *, dtype=None, device=None):
    """GPU ``con_K`` with the reference's signature and shape rules (``gaussian_process.py:16-36``): 1-D ``x`` is
    promoted to one row, a single-row result is flattened to 1-D (cdist path), ``return_d`` also returns
    ``D[n, :, m] = x_n - y_m``.  ``method`` is accepted for compatibility (both paths give the same K)."""
    dtype = dtype or _DEFAULT_DTYPE
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if x.ndim == 1:
        x = x[None, :]
    k = _shared_kernels(device, dtype)
    npdt = np.float32 if dtype == "float32" else np.float64
    # translation invariance: centre before a possible cast to float32
    c = y.mean(0) if len(y) else np.zeros(x.shape[1])
    xd = torch.from_numpy(np.ascontiguousarray((x - c).astype(npdt))).to(k.device)
    yd = torch.from_numpy(np.ascontiguousarray((y - c).astype(npdt))).to(k.device)
    if return_d or method != "cdist":
        K, D = k.con_k(xd, yd, beta, return_d=True)
        K = K.to(torch.float64).cpu().numpy()
        K = np.squeeze(K)
        if return_d:
            return K, D.to(torch.float64).cpu().numpy()
        return K
    K = k.con_k(xd, yd, beta).to(torch.float64).cpu().numpy()
    if len(K) == 1:
        K = K.flatten()
    return K


_EVAL_ALL = (_lib.EVAL_V | _lib.EVAL_JAC | _lib.EVAL_DIV | _lib.EVAL_CURL | _lib.EVAL_ACC | _lib.EVAL_CURV |
             _lib.EVAL_TORS | _lib.EVAL_JDET)
_EVAL_BYTES_PER_POINT = 8 * (3 + 9 + 1 + 3 + 3 + 3 + 3 + 1)
# every quantity is computed by the first call and kept on the device for the next ones while ALL of them fit in this many
# bytes (256 MB = 1.2 M query points; a 64^3 grid takes 55 MB); beyond that only what a call asks for is computed and kept
_EVAL_PREFETCH_CAP = 256 << 20


class _FusedEval:
    """ONE evaluator launch behind several API calls.  The kernel accumulates v and the 3 x 3 Jacobian of a query point
    in registers whatever is asked for; every further quantity is a few register operations and an HBM store.  The
    reference's call shape, however, is one call per quantity on the same points (``get_Jacobian()(X)`` then
    ``compute_curl(X=X)``; the seven ``morphofield_*`` wrappers, each with a fresh vector-field object).  So the first
    call on (points, field) launches once for ALL quantities and keeps them on the device (thread-local, one entry);
    later calls on the same points and the same field - compared by value - only copy their quantity to the host."""

    def __init__(self, X, sig):
        # the points are remembered by a 128-bit digest of their bytes where xxhash is installed (0.37 ms for the 64^3 grid:
        # a copy at the first call and a value comparison at the second cost 0.5 + 0.6 ms), else by a copy
        self.shape, self.digest = X.shape, _digest(X)
        self.X = X.copy() if self.digest is None else None
        self.sig, self.flags, self.dev, self.k = sig, 0, {}, None

    def matches(self, X, sig):
        if X.shape != self.shape or len(sig) != len(self.sig):
            return False
        for a, b in zip(sig, self.sig):
            if isinstance(a, np.ndarray) or isinstance(b, np.ndarray):
                if not (isinstance(a, np.ndarray) and isinstance(b, np.ndarray) and a.shape == b.shape
                        and np.array_equal(a, b)):
                    return False
            elif a != b:
                return False
        return np.array_equal(X, self.X) if self.digest is None else _digest(X) == self.digest


def _digest(X):
    """128-bit xxh3 digest of a float64 array's values (None when the xxhash module is missing)."""
    try:
        import xxhash
    except ImportError:  # pragma: no cover - the image has it
        return None
    return xxhash.xxh3_128_intdigest(np.ascontiguousarray(X))


def clear_eval_cache():
    """Drop the evaluator results kept on the device by the last ``SvcVectorField`` / ``GPVectorField`` /
    ``vector_field_function`` call of this thread (one entry: the quantities of the last (points, field) pair, at most
    ``_EVAL_PREFETCH_CAP`` bytes when prefetched).  A new fit (``SparseVFCEngine``) drops it by itself before it sizes its
    kernel-value cache against the free HBM."""
    _TLS.__dict__.pop("fused", None)


def _fused_eval(X, sig, flags, k, launch, rows3=False):
    """Host arrays {flag: ndarray} of the requested quantities; ``launch(flags) -> {flag: device tensor}``.
    rows3: every (n, 3) quantity comes back as (n, 3, 3) with its row repeated - the reference's ``zeros((n, 3, 3))`` quirk of
    curl and torsion - expanded on the DEVICE and copied once into page-locked memory (np.repeat on the host wrote the same
    19 MB of the 64^3 grid at 16 GB/s: 1.2 ms of a 3.9 ms call pair)."""
    sig = tuple(np.array(a, dtype=np.float64) if isinstance(a, (np.ndarray, list, tuple)) else a for a in sig)
    ent = _TLS.__dict__.get("fused")
    if ent is None or ent.k is not k or not ent.matches(X, sig):
        ent = _TLS.fused = _FusedEval(X, sig)
        ent.k = k
    missing = flags & ~ent.flags
    if missing:
        want = _EVAL_ALL if len(X) * _EVAL_BYTES_PER_POINT <= _EVAL_PREFETCH_CAP else missing
        want &= ~ent.flags
        ent.dev.update(launch(want))
        ent.flags |= want
    fl = [f for f in ent.dev if flags & f]
    dev = [ent.dev[f] for f in fl]
    if rows3:
        dev = [t[:, None, :].expand(t.shape[0], 3, t.shape[1]).contiguous() if t.dim() == 2 and t.shape[1] == 3 else t
               for t in dev]
    return dict(zip(fl, _to_host(k, dev)))


def _field_on_device(x, vf_dict, flags, dtype=None, device=None, rows3=False):
    """Run the fused evaluator for points x (n, d) against vf_dict's control points / coefficients."""
    dtype = dtype or _DEFAULT_DTYPE
    Xc = np.asarray(vf_dict["X_ctrl"], dtype=np.float64)
    Cc = np.asarray(vf_dict["C"], dtype=np.float64)
    d = Xc.shape[1]
    if x.shape[1] != d:
        raise ValueError(f"query points have {x.shape[1]} dimensions, the vector field has {d}")
    if d > 3 or Cc.shape[1] > 3:
        raise NotImplementedError("the HIP evaluators support up to 3 dimensions")
    k = _shared_kernels(device, dtype)
    beta = float(vf_dict["beta"])

    def launch(fl):
        center = Xc.mean(0)
        x4 = k.to_x4(x, center)
        c4 = k.to_x4(Xc, center)
        C3 = np.zeros((len(Xc), 3))
        C3[:, : Cc.shape[1]] = Cc
        Cd = torch.from_numpy(C3).to(k.device)
        return k.eval(x4, c4, beta, Cd, fl)

    return _fused_eval(x, ("svc", Xc, Cc, beta), flags, k, launch, rows3)


def vector_field_function(x, vf_dict, dim=None, *, dtype=None, device=None):
    """``v(x) = con_K(x, X_ctrl, beta) @ C`` (dynamo ``vector_field_function``; call site
    ``differential_geometry.py:67-68``).  1-D ``x`` -> 1-D output, like the reference's flattened single-row K."""
    x = np.array(x, dtype=np.float64)
    one = x.ndim == 1
    if one:
        x = x[None, :]
    dy = np.asarray(vf_dict["C"]).shape[1]
    v = _field_on_device(x, vf_dict, _lib.EVAL_V, dtype, device)[_lib.EVAL_V][:, :dy]
    if dim is not None:
        v = v[:, :dim] if np.isscalar(dim) else v[:, dim]
    return v[0] if one else v


def SparseVFC(
    X,
    Y,
    Grid,
    M=100,
    a=5,
    beta=None,
    ecr=1e-5,
    gamma=0.9,
    lambda_=3,
    minP=1e-5,
    MaxIter=500,
    theta=0.75,
    div_cur_free_kernels=False,
    velocity_based_sampling=True,
    sigma=0.8,
    eta=0.5,
    seed=0,
    lstsq_method="drouin",
    verbose=1,
    *,
    dtype=None,
    device=None,
    distributed=False,
    group=None,
    sharded_input=False,
    gather="root",
    gram_mode="full",
    force_collectives=False,
    collective="torch",
) -> dict:
    """Drop-in for ``dynamo.vectorfield.scVectorField.SparseVFC`` (defaults identical).

    Extra keyword-only arguments: ``dtype`` ("float64" parity mode | "float32" fast mode), ``device``,
    ``distributed``/``group`` (one process per GPU; cells are sharded across ranks, rank 0 alone does the host
    preprocessing and broadcasts the control points), ``sharded_input`` (False: every rank passes the same full X, Y and
    takes a block of it; True: every rank passes only ITS rows - ``Grid`` is still the same everywhere) and ``gather``
    ("root": the per-cell outputs ``V``, ``P``, ``VFCIndex`` are complete on rank 0 only, other ranks keep their own rows;
    "all": complete on every rank).  Multi-rank results carry ``row_range`` = (lo, hi): the positions within
    ``valid_ind`` that this rank's ``V`` / ``P`` rows correspond to (``VFCIndex`` counts from ``lo``).
    ``lstsq_method``: "scipy" (what Spateo passes) = minimum-norm solve with gelsd's eps * s_max cut-off on the
    device (Cholesky while the pivots certify full numerical rank, else the hand-written symmetric eigensolver);
    "drouin" maps to the same solve with a warning; "cholesky" is a non-reference fast mode.  ``gram_mode``: "full" (the
    reference's M-step on all M control points; the only mode since round 6).  ``collective``: "torch" (torch.distributed: RCCL under the
    "nccl" backend) | "mvf" (``mvf_allreduce_stats`` of the C ABI on the engine's own RCCL communicator).
    ``force_collectives=True`` with ``distributed=True`` runs the whole multi-rank protocol (rank-0 preprocessing +
    broadcast, the step's collectives, the output gather) on a process group of ONE rank as well - same result bit for
    bit, every collective executed on the real backend.  The coefficients ``C`` are
    NOT a parity quantity (the reference's own solver only fixes them up to the numerical null space of the Gram
    system) - the field ``V`` / ``grid_V``, ``sigma2`` and ``P`` are.
    Returns the reference's dict with host NumPy float64 arrays.
    """
    if div_cur_free_kernels:
        raise NotImplementedError("div_cur_free_kernels=True is out of scope (SURVEY.md Appendix A)")
    X = np.asarray(X, dtype=float)
    Y = np.asarray(Y, dtype=float)
    if X.ndim != 2 or Y.ndim != 2 or len(X) != len(Y):
        raise ValueError("X and Y must be 2-D arrays with the same number of rows")
    if gather not in ("root", "all"):
        raise ValueError("gather must be 'root' or 'all'")
    ph = _Phases(device)
    X_ori, Y_ori = X.copy(), Y.copy()
    rank, world = _dist_info(distributed, group)
    shard_sizes = None
    multi = world > 1 or (bool(force_collectives) and bool(distributed))
    if not multi:
        valid_ind, Xv, Yv, idx, ctrl_pts, beta = sparsevfc_preprocess(
            X, Y, M=M, beta=beta, velocity_based_sampling=velocity_based_sampling, seed=seed, device=device
        )
        N, lo, hi = len(Xv), 0, len(Xv)
    else:
        # Multi-rank: the O(N log N) host preprocessing (unique rows, control-point sampling, kNN bandwidth) runs on
        # rank 0 ONLY and its small result (ctrl_idx, control points, beta) is broadcast.  sharded_input=False: every
        # rank passed the same full X, Y and takes its block of the finite rows.  sharded_input=True: every rank passed
        # ITS OWN rows (any sizes); the finite rows are gathered on rank 0 for the control-point selection only.
        import torch.distributed as dist

        root = dist.get_global_rank(group, 0) if group is not None else 0
        valid_loc = np.where(np.isfinite(Y.sum(1)))[0]
        if sharded_input:
            lens = [None] * world
            with _current_device(device):
                dist.all_gather_object(lens, (len(X), len(valid_loc)), group=group)  # two integers per rank
            offset = sum(n_in for n_in, _ in lens[:rank])
            shard_sizes = [n_v for _, n_v in lens]
            # global row numbers of the finite rows (every rank) and the finite rows themselves (rank 0 only, for the
            # control-point selection): padded TENSOR collectives, not pickles of whole arrays
            valid_ind = _gather_rows_np((valid_loc + offset)[:, None].astype(np.int64), shard_sizes, rank, world, group,
                                        device, to_all=True)[:, 0]
            Xloc, Yloc = X[valid_loc], Y[valid_loc]
            XY = _gather_rows_np(np.concatenate([Xloc, Yloc], axis=1), shard_sizes, rank, world, group, device,
                                 to_all=False)
            if rank == 0:
                Xall, Yall = np.ascontiguousarray(XY[:, : X.shape[1]]), np.ascontiguousarray(XY[:, X.shape[1]:])
            del XY
            N = sum(shard_sizes)
        else:
            valid_ind = valid_loc
            Xall, Yall = X[valid_loc], Y[valid_loc]
            N = len(valid_loc)
            lo, hi = shard_bounds(N, rank, world)
            Xloc, Yloc = Xall[lo:hi], Yall[lo:hi]
        if N == 0:
            raise ValueError("SparseVFC: no row of Y is finite - nothing to fit.")
        box = [None]
        if rank == 0:
            try:
                _, _, _, idx0, ctrl0, beta0 = sparsevfc_preprocess(Xall, Yall, M=M, beta=beta,
                                                                   velocity_based_sampling=velocity_based_sampling, seed=seed,
                                                                   device=device)
                box = [(idx0, ctrl0, beta0)]
            except Exception as exc:  # every rank must leave the collective: ship the error
                box = [exc]
        with _current_device(device):
            dist.broadcast_object_list(box, src=root, group=group)  # (ctrl_idx, M control points, beta): small
        if isinstance(box[0], Exception):
            raise box[0]
        idx, ctrl_pts, beta = box[0]
        Xv, Yv, lo, hi = Xloc, Yloc, 0, len(Xloc)
    if len(ctrl_pts) < 2:
        # reference behaviour: con_K(ctrl, ctrl) of a single control point is flattened to 1-D (gaussian_process.py:23-24)
        # and the energy term C.T.dot(K).dot(C) then fails with a ValueError
        raise ValueError("SparseVFC needs at least 2 control points (shapes (3,) and (1,3) not aligned in the reference)")
    ph.mark("preprocess_s")
    eng = SparseVFCEngine(Xv[lo:hi], Yv[lo:hi], ctrl_pts, beta, dtype=dtype, device=device, distributed=distributed,
                          group=group, n_total=N, shard_sizes=shard_sizes, gram_mode=gram_mode,
                          force_collectives=force_collectives, collective=collective)
    ph.mark("upload_and_u_cache_s")
    tecr_vec, E_vec = eng.fit(a=a, gamma=gamma, lambda_=lambda_, minP=minP, MaxIter=MaxIter, theta=theta, ecr=ecr,
                              lstsq_method=lstsq_method)
    ph.mark("em_s")
    V, P, C = eng.results(gather=gather)
    grid_V = eng.predict(Grid) if Grid is not None else None
    i = eng.iteration
    if eng.comm is not None:
        eng.comm.close()
    extra = {}
    if multi:
        # which rows of the finite-row sequence (positions in `valid_ind`) the per-cell outputs V / P / VFCIndex of THIS
        # rank cover: all of them on rank 0 and with gather="all", this rank's block otherwise (VFCIndex is relative to it)
        first = sum(eng.shard_sizes[:rank])
        extra["row_range"] = (0, N) if (gather == "all" or rank == 0) else (first, first + eng.shard_sizes[rank])
    vfc_index = np.where(P > theta)[0]
    ph.mark("download_s")
    ph.done()
    ph.out["em_iterations"] = int(i)
    return {
        **extra,
        "X": X_ori,
        "valid_ind": valid_ind,
        "X_ctrl": ctrl_pts,
        "ctrl_idx": idx,
        "Y": Y_ori,
        "beta": beta,
        "V": V,
        "C": C,
        "P": P,
        "VFCIndex": vfc_index,
        "sigma2": eng.sigma2,
        "grid": Grid,
        "grid_V": grid_V,
        "iteration": i - 1,
        "tecr_traj": tecr_vec[:i],
        # the same vector under the name Spateo's docstring gives it (sparsevfc.py:155, :301 "tecr_vec"); dynamo's dict key is
        # believed to be "tecr_traj" (SURVEY.md App. A [VERIFY]): a consumer of either name finds it
        "tecr_vec": tecr_vec[:i],
        "E_traj": E_vec[:i],
    }


# =====================================================================================================================
# SvcVectorField (the class shape Spateo uses) backed by the fused evaluator kernel
# =====================================================================================================================
class SvcVectorField:
    """Counterpart of ``dynamo.vectorfield.scVectorField.SvcVectorField`` as used by
    ``differential_geometry.py:25-28``; in-tree twin ``GPVectorField.py:193-266``.  Every ``compute_*`` keeps the
    reference's return shapes, including the (n, 3, 3) broadcast of 3-D curl / torsion."""

    def __init__(self, dtype=None, device=None):
        self.data = {}
        self.vf_dict = None
        self.func = None
        self._dtype, self._device = dtype, device

    def from_adata(self, adata, basis=None, vf_key="VecFld"):
        if basis is not None and len(basis) > 0:
            vf_key = "%s_%s" % (vf_key, basis)
        if vf_key not in adata.uns.keys():
            raise ValueError(f"Vector field function {vf_key} is not included in the adata object!")
        vf_dict = adata.uns[vf_key]
        self.vf_dict = vf_dict
        self.func = lambda x: vector_field_function(x, vf_dict, dtype=self._dtype, device=self._device)
        self.data["X"] = vf_dict["X"]
        self.data["V"] = vf_dict["Y"]  # dynamo keeps the raw input velocities here
        return self

    def get_data(self):
        return self.data["X"], self.data["V"]

    def _eval(self, X, flags, rows3=False):
        X = np.asarray(X, dtype=np.float64)
        return _field_on_device(X, self.vf_dict, flags, self._dtype, self._device, rows3)

    @staticmethod
    def _check_method(method):
        if method != "analytical":
            raise NotImplementedError("only method='analytical' is supported (numdifftools path is out of scope)")

    def get_Jacobian(self, method="analytical", **kwargs):
        """Returns ``f(x) -> (d, d, n)`` (``(d, d)`` for a 1-D x); ``J[f, i] = d f_f / d x_i``."""
        self._check_method(method)

        def jac(x):
            x = np.asarray(x, dtype=np.float64)
            one = x.ndim == 1
            xx = x[None, :] if one else x
            d = xx.shape[1]
            J = self._eval(xx, _lib.EVAL_JAC)[_lib.EVAL_JAC][:d, :d, :]
            return J[:, :, 0] if one else J

        return jac

    def jacobian_with_det(self, X, method="analytical"):
        """(Js (d, d, n), det Js (n,)) from ONE evaluator pass: what ``morphofield_jacobian`` stores in ``uns`` and ``obs``
        (``differential_geometry.py:331-337`` loops ``np.linalg.det`` over the cells on the host; here the 3 x 3
        determinant is the kernel's MVF_EVAL_JDET output, a cofactor expansion in the registers that hold J)."""
        self._check_method(method)
        X = np.asarray(X, dtype=np.float64)
        d = X.shape[1]
        o = self._eval(X, _lib.EVAL_JAC | (_lib.EVAL_JDET if d == 3 else 0))
        J = o[_lib.EVAL_JAC][:d, :d, :]
        if d == 3:
            return J, o[_lib.EVAL_JDET]
        # the kernel works on zero-padded 3-D points: the d x d determinant of a 1-D / 2-D field is taken on the host
        return J, (J[0, 0] * J[1, 1] - J[0, 1] * J[1, 0] if d == 2 else J[0, 0].copy())

    def compute_velocity(self, X):
        return self.func(X)

    def compute_acceleration(self, X=None, method="analytical", **kwargs):
        self._check_method(method)
        X = self.data["X"] if X is None else X
        d = np.asarray(X).shape[1]
        acc = self._eval(X, _lib.EVAL_ACC)[_lib.EVAL_ACC][:, :d]
        return np.linalg.norm(acc, axis=1), acc

    def compute_curvature(self, X=None, method="analytical", formula=2, **kwargs):
        self._check_method(method)
        X = self.data["X"] if X is None else X
        d = np.asarray(X).shape[1]
        if formula == 2:
            cm = self._eval(X, _lib.EVAL_CURV)[_lib.EVAL_CURV][:, :d]
            return np.linalg.norm(cm, axis=1), cm
        elif formula == 1:
            o = self._eval(X, _lib.EVAL_V | _lib.EVAL_ACC)
            v, a = o[_lib.EVAL_V], o[_lib.EVAL_ACC]
            # ||v a^T||_F / ||v||^3  ==  ||v|| ||a|| / ||v||^3
            nv, na = np.linalg.norm(v, axis=1), np.linalg.norm(a, axis=1)
            return nv * na / nv**3, None
        n = len(np.asarray(X))
        return np.zeros(n), None  # the reference leaves zeros for any other formula value

    def compute_curl(self, X=None, method="analytical", dim1=0, dim2=1, dim3=2, **kwargs):
        self._check_method(method)
        X = self.data["X"] if X is None else np.asarray(X)
        cols = [dim1, dim2] if dim3 is None or X.shape[1] == 2 else [dim1, dim2, dim3]
        if cols != list(range(X.shape[1])):  # the default selection of a 2-D / 3-D X is X itself: no gather of n rows
            X = X[:, cols]
        if X.shape[1] == 2:
            return self._eval(X, _lib.EVAL_CURL)[_lib.EVAL_CURL][:, 2].copy()  # J10 - J01
        elif X.shape[1] == 3:
            # reference quirk (GPVectorField.py:64-68): the 3-vector is assigned into zeros((n, 3, 3))
            return self._eval(X, _lib.EVAL_CURL, rows3=True)[_lib.EVAL_CURL]
        raise ValueError("X has incorrect dimensions.")

    def compute_torsion(self, X=None, method="analytical", **kwargs):
        self._check_method(method)
        X = self.data["X"] if X is None else np.asarray(X)
        if X.shape[1] != 3:
            raise Exception("torsion is only defined in 3 dimension.")
        return self._eval(X, _lib.EVAL_TORS, rows3=True)[_lib.EVAL_TORS]  # same broadcast as GPVectorField.py:87-92

    def compute_divergence(self, X=None, method="analytical", vectorize_size=1000, **kwargs):
        self._check_method(method)
        X = self.data["X"] if X is None else X
        return self._eval(X, _lib.EVAL_DIV)[_lib.EVAL_DIV]


# =====================================================================================================================
# Gaussian-process morphofield variant: same kernels + norm_dict scaling + rigid part
# =====================================================================================================================
def _gp_scalars(vf_dict):
    nd = vf_dict["norm_dict"]
    sf, stt = np.asarray(nd["scale_fixed"], dtype=float), np.asarray(nd["scale_transformed"], dtype=float)
    if sf.size != 1 or stt.size != 1:
        raise NotImplementedError("per-axis norm_dict scales are not supported by the HIP path")
    if vf_dict["kernel_type"] == "geodist":
        raise NotImplementedError("geodist is not implemented yet")  # as the reference (gaussian_process.py:112-113)
    if vf_dict["kernel_type"] != "euc":
        raise ValueError("current only support cdist and geodist")
    return float(sf), float(stt), np.asarray(nd["mean_fixed"], dtype=float), np.asarray(nd["mean_transformed"], dtype=float)


def _gp_eval(X, vf_dict, flags, nonrigid_only=False, dtype=None, device=None, rows3=False):
    """Fused evaluator on the GP field: v = _gp_velocity(X) (``gaussian_process.py:102-127``), J = the reference's
    ``Jacobian_GP_gaussian_kernel`` (non-rigid part x scale_fixed/scale_transformed, ``GPVectorField.py:143-190``)."""
    dtype = dtype or _DEFAULT_DTYPE
    X = np.asarray(X, dtype=np.float64)
    sf, stt, mean_f, mean_t = _gp_scalars(vf_dict)
    ind = np.asarray(vf_dict["inducing_variables"], dtype=np.float64)
    Coff = np.asarray(vf_dict["Coff"], dtype=np.float64)
    d = ind.shape[1]
    if d != 3 or X.shape[1] != 3:
        raise NotImplementedError("the GP variant of the HIP path is 3-D")
    xn = (X - mean_t) / stt
    center = ind.mean(0)
    if nonrigid_only:
        A = (sf - stt) / 10000.0 * np.eye(3)
        b = np.zeros(3)
    else:
        R, t = np.asarray(vf_dict["R"], dtype=float), np.asarray(vf_dict["t"], dtype=float).reshape(3)
        A = (sf * R - stt * np.eye(3)) / 10000.0
        b = (sf * t + mean_f - mean_t) / 10000.0
    b = b + A @ center  # the kernel sees q = xn - center
    k = _shared_kernels(device, dtype)
    beta = float(vf_dict["beta"])

    def launch(fl):
        x4, c4 = k.to_x4(xn, center), k.to_x4(ind, center)
        Cd = torch.from_numpy(np.ascontiguousarray(Coff[:, :3])).to(k.device)
        return k.eval(x4, c4, beta, Cd, fl, affine=(sf / 10000.0, sf / stt, A, b))

    return _fused_eval(X, ("gp", ind, Coff, beta, sf, stt, A, b, mean_t), flags, k, launch, rows3)


def gp_velocity(X, vf_dict, nonrigid_only=False, *, dtype=None, device=None):
    """GPU ``_gp_velocity`` (``gaussian_process.py:102-127``)."""
    X = np.asarray(X, dtype=np.float64)
    one = X.ndim == 1
    v = _gp_eval(X[None, :] if one else X, vf_dict, _lib.EVAL_V, nonrigid_only, dtype, device)[_lib.EVAL_V]
    return v[0] if one else v


class GPVectorField(SvcVectorField):
    """Counterpart of the in-tree ``GPVectorField`` (``morphofield_dg/GPVectorField.py:193-266``): same methods as
    :class:`SvcVectorField`, evaluated on the GP field (norm_dict scaling, rigid part unless ``nonrigid_only``)."""

    def from_adata(self, adata, vf_key="VecFld", nonrigid_only=False):
        if vf_key in adata.uns.keys():
            vf_dict = adata.uns[vf_key]
        else:
            raise Exception(
                f"The {vf_key} that corresponds to the reconstructed vector field is not in ``anndata.uns``."
                f"Please run ``st.align.morpho_align(adata, vecfld_key_added='{vf_key}')`` before running this function."
            )
        self.vf_dict = vf_dict
        self.nonrigid_only = nonrigid_only
        self.func = lambda x: gp_velocity(x, vf_dict, nonrigid_only=nonrigid_only, dtype=self._dtype,
                                          device=self._device)
        self.data["X"] = vf_dict["X"]
        self.data["V"] = vf_dict["V"]
        return self

    def compute_velocity(self, X):
        return self.func(X)

    def _eval(self, X, flags, rows3=False):
        return _gp_eval(np.asarray(X, dtype=np.float64), self.vf_dict, flags, getattr(self, "nonrigid_only", False),
                        self._dtype, self._device, rows3)


# =====================================================================================================================
# trajectory integration (morphopath)
# =====================================================================================================================
def _default_t_end(X, V):
    """dynamo ``getTend``: extent of the data over the 1st percentile of the non-zero |velocity| entries."""
    V_abs = np.abs(np.asarray(V, dtype=float))
    V_abs = V_abs[np.isfinite(V_abs) & (V_abs > 0)]
    return float(np.max(X.max(0) - X.min(0)) / np.percentile(V_abs, 1))


def _hermite(tq, tk, xk, vk):
    """Cubic Hermite interpolation of trajectories: samples xk (n, K, d) with velocities vk at uniform times tk (K,),
    evaluated at per-trajectory times tq (n, Q) -> (n, Q, d).  O(h^4): with the fine RK4 samples this is the ODE's dense
    output to ~1e-8."""
    h = tk[1] - tk[0]
    u = (tq - tk[0]) / h
    i = np.clip(np.floor(u).astype(np.int64), 0, len(tk) - 2)
    w = (u - i)[..., None]
    rows = np.arange(xk.shape[0])[:, None]
    x0, x1, v0, v1 = xk[rows, i], xk[rows, i + 1], vk[rows, i], vk[rows, i + 1]
    h00, h10 = (1 + 2 * w) * (1 - w) ** 2, w * (1 - w) ** 2
    h01, h11 = w * w * (3 - 2 * w), w * w * (w - 1)
    return h00 * x0 + h10 * h * v0 + h01 * x1 + h11 * h * v1


def _arc_length_resample(tk, xk, vk, n_out, stop_tol=1e-5):
    """dynamo ``fate`` semantics on a finely sampled trajectory (``integrate_vf_ivp(..., sampling="arc_length")``): the
    integration ends where every |v| component drops below 1e-5 (its terminal event), the path is cut into n_out points
    EQUALLY SPACED IN ARC LENGTH, the times of those points come from linear interpolation along the polyline, and the
    states are the ODE solution at those times.  Returns (t (n, n_out), x (n, n_out, d))."""
    n, K, d = xk.shape
    slow = np.all(np.abs(vk) < stop_tol, axis=2)
    end = np.where(slow.any(1), slow.argmax(1), K - 1)  # first sample at rest, else the last one
    seg = np.linalg.norm(np.diff(xk, axis=1), axis=2)
    seg[np.arange(K - 1)[None, :] >= end[:, None]] = 0.0  # nothing moves after the terminal event
    s = np.concatenate([np.zeros((n, 1)), np.cumsum(seg, axis=1)], axis=1)
    L = s[:, -1]
    sq = np.linspace(0.0, 1.0, n_out)[None, :] * L[:, None]
    tq = np.empty((n, n_out))
    for r in range(n):  # monotone inverse s -> t, row by row (np.interp is 1-D)
        e = max(int(end[r]), 1)
        tq[r] = np.interp(sq[r], s[r, : e + 1], tk[: e + 1]) if L[r] > 0 else np.linspace(tk[0], tk[e], n_out)
    return tq, _hermite(tq, tk, xk, vk)


def integrate_field(vf_dict, init_states, t_end=None, interpolation_num=250, direction="forward", average=False,
                    nonrigid_only=False, substeps=4, dtype=None, device=None, max_cells_per_launch=1 << 16,
                    sampling="arc_length"):
    """Integrate dx/dt = v(x) from every row of ``init_states`` on the GPU (fused RK4 kernel).

    Returns ``(t, prediction)``: lists with one entry per trajectory, ``t[i]`` (n_t,), ``prediction[i]`` (n_t, d).
    ``sampling="arc_length"`` (dynamo ``fate``'s default, which ``morphopath`` inherits): ``interpolation_num`` points
    equally spaced in arc length along each path (twice as many for ``direction="both"``), every trajectory with its own
    times; the path ends early where the field is at rest (all |v| < 1e-5).  ``"uniform_time"``: ``interpolation_num``
    uniform times over [0, t_end] ("forward"), [-t_end, 0] ("backward") or both.
    ``average``: False | "origin" (one trajectory from the mean start) | "trajectory" / True (mean over cells per sample)."""
    dtype = dtype or _DEFAULT_DTYPE
    X0 = np.asarray(init_states, dtype=np.float64)
    if X0.ndim == 1:
        X0 = X0[None, :]
    if direction not in ("forward", "backward", "both"):
        raise ValueError("direction must be one of 'forward', 'backward', 'both'")
    if sampling not in ("arc_length", "uniform_time"):
        raise ValueError("sampling must be 'arc_length' or 'uniform_time'")
    method = vf_dict.get("method", "sparsevfc")
    if t_end is None:
        t_end = _default_t_end(np.asarray(vf_dict["X"], dtype=float), vf_dict["V"])
    t_end = float(t_end)
    n_t = int(interpolation_num)
    if n_t < 2:
        raise ValueError("interpolation_num must be >= 2")
    if average == "origin":
        X0 = X0.mean(0, keepdims=True)
    d = X0.shape[1]
    k = _shared_kernels(device, dtype)
    if method == "gaussian_process":
        sf, stt, mean_f, mean_t = _gp_scalars(vf_dict)
        ctrl = np.asarray(vf_dict["inducing_variables"], dtype=np.float64)
        Cc = np.asarray(vf_dict["Coff"], dtype=np.float64)
        center = ctrl.mean(0)
        if nonrigid_only:
            A, b = (sf - stt) / 10000.0 * np.eye(3), np.zeros(3)
        else:
            R, tt = np.asarray(vf_dict["R"], dtype=float), np.asarray(vf_dict["t"], dtype=float).reshape(3)
            A, b = (sf * R - stt * np.eye(3)) / 10000.0, (sf * tt + mean_f - mean_t) / 10000.0
        # integrate in normalised coordinates xn = (X - mean_t) / stt:  dxn/dt = v / stt
        affine = (sf / 10000.0 / stt, 1.0, A / stt, (b + A @ center) / stt)
        start = (X0 - mean_t) / stt
        to_world = lambda q: q * stt + mean_t  # noqa: E731
        vscale = stt
    else:
        ctrl = np.asarray(vf_dict["X_ctrl"], dtype=np.float64)
        Cc = np.asarray(vf_dict["C"], dtype=np.float64)
        center = ctrl.mean(0)
        affine = None
        start = X0
        to_world = lambda q: q  # noqa: E731
        vscale = 1.0
    if ctrl.shape[1] > 3 or Cc.shape[1] != ctrl.shape[1]:
        raise NotImplementedError("trajectory integration needs a field with Dy == D <= 3")
    C3 = np.zeros((len(ctrl), 3))
    C3[:, : Cc.shape[1]] = Cc
    Cd = torch.from_numpy(C3).to(k.device)
    c4 = k.to_x4(ctrl, center)
    beta = float(vf_dict["beta"])
    arc = sampling == "arc_length"
    n_fine = 4 * n_t + 1 if arc else n_t  # dense RK4 samples the arc-length resampling works from
    dt = t_end / (n_fine - 1)
    tf = np.linspace(0.0, t_end, n_fine)

    def run(sign):
        """(times (n, n_t) or (n_fine,), states (n, n_t, d)) in world coordinates for one direction."""
        ts, xs = [], []
        for lo in range(0, len(start), max_cells_per_launch):
            x4 = k.to_x4(start[lo : lo + max_cells_per_launch], center)
            # arc-length mode samples 4x finer than the output, so `substeps` RK4 steps per OUTPUT interval become
            # max(2, substeps // 2) per fine interval (the default 4 -> 2, i.e. 8 per output interval; more on request)
            tr = k.integrate(x4, c4, beta, Cd, sign * dt, max(2, int(substeps) // 2) if arc else substeps, n_fine,
                             affine=affine)
            if not arc:
                xs.append(tr.cpu().numpy()[:, :, :d] + center[None, None, :d])
                continue
            # velocities at the dense samples (fused evaluator; same affine as the integrator), then dynamo's resampling
            pts = tr.reshape(-1, 3)
            p4 = torch.zeros(pts.shape[0], 4, dtype=k.tdtype, device=k.device)
            p4[:, :3] = pts.to(k.tdtype)
            vel = k.eval(p4, c4, beta, Cd, _lib.EVAL_V, affine=affine)[_lib.EVAL_V].reshape(tr.shape[0], n_fine, 3)
            xk = tr.cpu().numpy()[:, :, :d]
            tq, xq = _arc_length_resample(sign * tf, xk, sign * vel.cpu().numpy()[:, :, :d], n_t,
                                          stop_tol=1e-5 / vscale)
            ts.append(tq)
            xs.append(xq + center[None, None, :d])
        x = to_world(np.concatenate(xs, axis=0))
        return (np.concatenate(ts, axis=0) if arc else sign * tf), x

    if direction == "forward":
        times, traj = run(+1.0)
    elif direction == "backward":
        times, traj = run(-1.0)
    else:
        (tb, back), (tfw, fwd) = run(-1.0), run(+1.0)
        if arc:  # dynamo doubles interpolation_num for "both": the backward half reversed, then the forward half
            traj = np.concatenate([back[:, ::-1], fwd], axis=1)
            times = np.concatenate([tb[:, ::-1], tfw], axis=1)
        else:
            traj = np.concatenate([back[:, :0:-1], fwd], axis=1)
            times = np.concatenate([tb[:0:-1], tfw])
    if average in ("trajectory", True):
        traj = traj.mean(0, keepdims=True)
        if arc:
            times = times.mean(0, keepdims=True)
    if arc:
        return [times[i].copy() for i in range(len(traj))], [traj[i] for i in range(len(traj))]
    return [times.copy() for _ in range(len(traj))], [traj[i] for i in range(len(traj))]


def genesis_states(vf_dict, init_states, time_vec, substeps=64, dtype=None, device=None):
    """The numeric core of ``construct_genesis`` (``spateo/tdr/models/models_migration/morphopath_model.py:138-148``):
    starting from ``init_states`` the cells are displaced step by step, ``pts <- odeint(f, pts, [0, time_vec[i]])[1]``
    for every entry of ``time_vec`` (each entry is the DURATION of that step, as in the reference's loop), and the
    positions after every step are returned as a list of (n, d) arrays (the reference's ``stages_X``).  One fused RK4
    launch per step (``substeps`` RK4 steps each) instead of one SciPy ``odeint`` call per cell and step."""
    dtype = dtype or _DEFAULT_DTYPE
    pts = np.asarray(init_states, dtype=np.float64)
    ctrl = np.asarray(vf_dict["X_ctrl"], dtype=np.float64)
    Cc = np.asarray(vf_dict["C"], dtype=np.float64)
    d = ctrl.shape[1]
    if pts.ndim != 2 or pts.shape[1] != d or d > 3 or Cc.shape[1] != d:
        raise NotImplementedError("genesis_states needs (n, d) states and a field with Dy == D <= 3")
    k = _shared_kernels(device, dtype)
    center = ctrl.mean(0)
    C3 = np.zeros((len(ctrl), 3))
    C3[:, :d] = Cc
    Cd = torch.from_numpy(C3).to(k.device)
    c4 = k.to_x4(ctrl, center)
    beta = float(vf_dict["beta"])
    stages = []
    for dt in np.asarray(time_vec, dtype=np.float64):
        if dt != 0.0:
            tr = k.integrate(k.to_x4(pts, center), c4, beta, Cd, float(dt), int(substeps), 2)
            pts = tr[:, 1, :d].cpu().numpy() + center[None, :d]
        stages.append(pts.copy())
    return stages


# =====================================================================================================================
# batched independent fits (BASELINE config 5: 32 organs x ~250 k cells, M = 500) - replicas only, no collective
# =====================================================================================================================
def SparseVFC_many(datasets, n_streams=4, device=None, distributed=False, group=None, **kwargs):
    """Fit several independent vector fields concurrently: ``datasets`` = list of ``(X, Y, Grid)``.

    Within a process the fits run on ``n_streams`` HIP streams (one host thread per stream; every libmvf call is
    asynchronous on the calling thread's current stream and host syncs are per stream), so small fits overlap on the
    GPU.  With ``distributed=True`` organ ``i`` is fitted by rank ``i % world`` and the result dicts are exchanged with
    ``all_gather_object`` - there is no data-path collective ("replicas only").
    Returns the list of result dicts in input order."""
    import threading

    rank, world = _dist_info(distributed, group)
    mine = [i for i in range(len(datasets)) if i % world == rank]
    results = {}
    errors = []
    if torch.cuda.is_available():
        dev = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        streams = [torch.cuda.Stream(device=dev) for _ in range(max(1, min(n_streams, len(mine))))]
    else:
        dev, streams = None, [None]

    def worker(slot):
        try:
            for pos in range(slot, len(mine), len(streams)):
                i = mine[pos]
                X, Y, Grid = datasets[i]
                if streams[slot] is None:
                    results[i] = SparseVFC(X, Y, Grid, device=device, **kwargs)
                else:
                    with torch.cuda.stream(streams[slot]):
                        results[i] = SparseVFC(X, Y, Grid, device=dev, **kwargs)
                        streams[slot].synchronize()
        except Exception as exc:  # surfaced in the caller's thread
            errors.append(exc)

    threads = [threading.Thread(target=worker, args=(s_,)) for s_ in range(len(streams))]
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    if world > 1:
        # (results, error) travel together and every rank takes part in the gather BEFORE anybody raises: a rank that
        # raised first (round 5) left the others waiting in all_gather_object for ever.  The error is sent as text - an
        # exception object need not pickle - and re-raised on EVERY rank, naming the rank it came from.
        import torch.distributed as dist

        mine_err = None if not errors else f"{type(errors[0]).__name__}: {errors[0]}"
        gathered = [None] * world
        dist.all_gather_object(gathered, (results, mine_err), group=group)
        if errors:
            raise errors[0]
        for r_, (_, err) in enumerate(gathered):
            if err is not None:
                raise _lib.MVFError(f"SparseVFC_many: a fit failed on rank {r_}: {err}")
        results = {k_: v for part, _ in gathered for k_, v in part.items()}
    elif errors:
        raise errors[0]
    return [results[i] for i in range(len(datasets))]
