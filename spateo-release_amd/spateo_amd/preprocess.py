"""Host-side preprocessing of SparseVFC: finite rows, `np.unique` rows, the
velocity-weighted control-point draw (dynamo's `sample_by_velocity`: in-tree copy spateo/alignment/methods/sampling.py:225-241)
and the kNN bandwidth rule -> beta.  NumPy on the host, bit-identical to the oracle on purpose; from 16 k rows / 1024 control
points on the unique-rows and neighbour searches run on the device (`mvf_unique_rows`, `mvf_knn_rowsum`)."""
from __future__ import annotations

import numpy as np
import torch

from . import _runtime as _rt
from ._runtime import _shared_kernels

_DEVICE_KNN_MIN_POINTS = 256  # (1024 until round 6: the kd-tree of M = 500 control points is 6 ms of host time per C5 organ)


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
    return _sample_by_norms(row_norms(V), n, seed)


def row_norms(V: np.ndarray) -> np.ndarray:
    """``np.linalg.norm(V, axis=1)`` bit for bit (= sqrt of the squares added column by column, which is what add.reduce does
    over a last axis shorter than its 8-wide unrolled blocks) at a third of the time: 4.1 -> 1.6 ms at 250 k x 3."""
    V = np.asarray(V)
    if V.ndim != 2 or not 1 <= V.shape[1] < 8 or V.dtype != np.float64:
        return np.linalg.norm(V, axis=1)
    acc = V[:, 0] * V[:, 0]
    for c in range(1, V.shape[1]):
        acc += V[:, c] * V[:, c]
    return np.sqrt(acc, out=acc)


def finite_rows(Y: np.ndarray) -> np.ndarray:
    """``np.where(np.isfinite(Y.sum(1)))[0]`` (Appendix A step 1).  Usual case - every entry finite and too small for a row
    sum to overflow - answered from min / max alone (0.5 instead of 3.9 ms at 250 k x 3); NaN poisons min / max, so any
    doubt takes the reference's own expression."""
    Y = np.asarray(Y)
    if Y.ndim == 2 and Y.size and Y.dtype.kind == "f":
        lo, hi = float(Y.min()), float(Y.max())
        if np.isfinite(lo) and np.isfinite(hi) and max(abs(lo), abs(hi)) * Y.shape[1] < 1e300:
            return np.arange(Y.shape[0])
    return np.where(np.isfinite(Y.sum(1)))[0]


def _sample_by_norms(tmp_V: np.ndarray, n: int, seed: int = 19491001) -> np.ndarray:
    """The draw of ``sample_by_velocity`` from the row norms themselves (same values in the same order: same indices)."""
    rs = np.random.RandomState(seed)
    p = tmp_V / np.sum(tmp_V)
    idx = rs.choice(np.arange(len(tmp_V)), size=n, p=p, replace=False)
    with _RNG_LOCK:
        np.random.set_state(rs.get_state())
    return idx


# (measured, tools/unique_rows_small_probe.py: device 0.94 / 0.99 / 1.10 / 1.16 / 1.28 ms at 10 / 20 / 50 / 100 / 200 k rows incl. both
# copies, host 0.66 / 1.44 / 3.89 / 8.28 / 17.5 ms - the round-4 threshold of 200 k rows left 2.8 ms of a 16 ms BASELINE config 2 call on the host)
_DEVICE_UNIQUE_MIN_ROWS = 16_000


def unique_rows(X: np.ndarray, device=None):
    """``np.unique(X, axis=0, return_index=True)`` (lexicographically sorted unique rows + index of the FIRST
    occurrence of each) without NumPy's structured-view sort, which is the slowest host step at millions of cells
    (12 s at 8 M).  From 16 k rows on, with a GPU: ``mvf_unique_rows`` (stable LSD radix sort over the columns +
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
    valid_ind = finite_rows(Y)
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
        idx = _sample_by_norms(row_norms(Yv)[uid], M)
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
