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

from . import _lib
from . import _runtime as _rt
from ._runtime import (DEFLATED_MIN_M, _check_lstsq_method, _Phases, _shared_kernels, _TLS, _to_host,  # noqa: F401
                       clear_eval_cache, last_fit_profile, set_default_dtype)
from .engine import SparseVFCEngine, _current_device, _dist_info, _gather_rows_np, shard_bounds  # noqa: F401
from .preprocess import (bandwidth_selector, finite_rows, sample_by_velocity, sparsevfc_preprocess, unique_rows,  # noqa: F401
                         _sample_by_norms, _sparsevfc_preprocess)

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


def __getattr__(name):
    """`vectorfield._DEFAULT_DTYPE` / `vectorfield.PROFILE_FITS` read the runtime module's current values (they are SET through
    `set_default_dtype` and `spateo_amd._runtime.PROFILE_FITS`)."""
    if name in ("_DEFAULT_DTYPE", "PROFILE_FITS"):
        return getattr(_rt, name)
    raise AttributeError(f"module {__name__!r} has no attribute {name!r}")


# =====================================================================================================================
# drop-in functions
# =====================================================================================================================
def con_K(x, y, beta: float = 0.1, method: str = "cdist", return_d: bool = False, *, dtype=None, device=None):
    """GPU ``con_K`` with the reference's signature and shape rules (``gaussian_process.py:16-36``): 1-D ``x`` is
    promoted to one row, a single-row result is flattened to 1-D (cdist path), ``return_d`` also returns
    ``D[n, :, m] = x_n - y_m``.  ``method`` is accepted for compatibility (both paths give the same K)."""
    dtype = dtype or _rt._DEFAULT_DTYPE
    x = np.asarray(x, dtype=np.float64)
    y = np.asarray(y, dtype=np.float64)
    if x.ndim == 1:
        x = x[None, :]
    k = _shared_kernels(device, dtype)
    npdt = np.float32 if dtype == "float32" else np.float64
    # translation invariance: centre before a possible cast to float32
    c = y.mean(0) if len(y) else np.zeros(x.shape[1])
    xd = k.h2d((x - c).astype(npdt))
    yd = k.h2d((y - c).astype(npdt))
    if return_d or method != "cdist":
        K, D = k.con_k(xd, yd, beta, return_d=True)
        K = _rt._d2h(k, K.to(torch.float64))
        K = np.squeeze(K)
        if return_d:
            return K, _rt._d2h(k, D.to(torch.float64))
        return K
    K = _rt._d2h(k, k.con_k(xd, yd, beta).to(torch.float64))
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
    dtype = dtype or _rt._DEFAULT_DTYPE
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
        Cd = k.h2d(C3)
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
        valid_loc = finite_rows(Y)
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
    try:
        tecr_vec, E_vec = eng.fit(a=a, gamma=gamma, lambda_=lambda_, minP=minP, MaxIter=MaxIter, theta=theta, ecr=ecr,
                                  lstsq_method=lstsq_method)
        ph.mark("em_s")
        V, P, C = eng.results(gather=gather)
        grid_V = eng.predict(Grid) if Grid is not None else None
        i = eng.iteration
    finally:
        # the RCCL communicator is destroyed here on EVERY path (a failure raises on all ranks in the same EM step, so they
        # all arrive here together) - not at garbage collection with the ranks out of step
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
def _gp_scalars(vf_dict, d):
    """(scale_fixed (d,), scale_transformed (d,), mean_fixed (d,), mean_transformed (d,), per_axis) of the GP variant's
    norm_dict: scalars or anything that broadcasts against one point, as ``_gp_velocity`` uses them
    (``gaussian_process.py:107,117-126``)."""
    nd = vf_dict["norm_dict"]
    if vf_dict["kernel_type"] == "geodist":
        raise NotImplementedError("geodist is not implemented yet")  # as the reference (gaussian_process.py:112-113)
    if vf_dict["kernel_type"] != "euc":
        raise ValueError("current only support cdist and geodist")
    sf_raw, stt_raw = np.asarray(nd["scale_fixed"], dtype=float), np.asarray(nd["scale_transformed"], dtype=float)
    vec = lambda a: np.broadcast_to(np.asarray(a, dtype=float).reshape(-1), (d,)).astype(float)  # noqa: E731
    return vec(sf_raw), vec(stt_raw), vec(nd["mean_fixed"]), vec(nd["mean_transformed"]), bool(sf_raw.size != 1 or stt_raw.size != 1)


def _gp_eval(X, vf_dict, flags, nonrigid_only=False, dtype=None, device=None, rows3=False):
    """Fused evaluator on the GP field: v = _gp_velocity(X) (``gaussian_process.py:102-127``), J = the reference's
    ``Jacobian_GP_gaussian_kernel`` (non-rigid part x scale_fixed/scale_transformed, ``GPVectorField.py:143-190``).
    1-D to 3-D fields (the kernel works on zero-padded 3-D points; round 5 refused anything but 3-D) and per-axis
    ``norm_dict`` scales for the velocity (round 5 refused them; the reference's own Jacobian multiplies a (d, d, n) array by
    ``scale_fixed / scale_transformed`` and therefore only broadcasts a scalar ratio - a per-axis one raises here too)."""
    dtype = dtype or _rt._DEFAULT_DTYPE
    X = np.asarray(X, dtype=np.float64)
    ind = np.asarray(vf_dict["inducing_variables"], dtype=np.float64)
    Coff = np.asarray(vf_dict["Coff"], dtype=np.float64)
    d = ind.shape[1]
    if not 1 <= d <= 3 or X.shape[1] != d or Coff.shape[1] > 3:
        raise NotImplementedError("the HIP path of the GP variant evaluates 1-D to 3-D fields on points of the same dimension")
    sf, stt, mean_f, mean_t, per_axis = _gp_scalars(vf_dict, d)
    ratio = sf / stt
    if flags & ~_lib.EVAL_V and not np.all(ratio == ratio[0]):
        raise ValueError("operands could not be broadcast together: the Jacobian of a GP field with per-axis "
                         "scale_fixed / scale_transformed ((d, d, n) * (d,)), as in the reference (GPVectorField.py:190)")
    xn = (X - mean_t) / stt
    center = ind.mean(0)
    pad = lambda v: np.concatenate([np.asarray(v, dtype=float).reshape(-1), np.zeros(3 - d)])  # noqa: E731
    A3 = np.zeros((3, 3))
    if nonrigid_only:
        A3[:d, :d] = np.diag((sf - stt) / 10000.0)
        b3 = np.zeros(3)
    else:
        R, t = np.asarray(vf_dict["R"], dtype=float), np.asarray(vf_dict["t"], dtype=float).reshape(-1)
        A3[:d, :d] = (sf[:, None] * R - np.diag(stt)) / 10000.0
        b3 = pad((sf * t + mean_f - mean_t) / 10000.0)
    b3 = b3 + A3 @ pad(center)  # the kernel sees q = xn - center
    alpha3 = pad(sf / 10000.0)
    k = _shared_kernels(device, dtype)
    beta = float(vf_dict["beta"])

    def launch(fl):
        x4, c4 = k.to_x4(xn, center), k.to_x4(ind, center)
        C3 = np.zeros((len(ind), 3))
        C3[:, : Coff.shape[1]] = Coff
        Cd = k.h2d(C3)
        return k.eval(x4, c4, beta, Cd, fl, affine=(alpha3, float(ratio[0]), A3, b3))

    return _fused_eval(X, ("gp", ind, Coff, beta, sf, stt, A3, b3, mean_t), flags, k, launch, rows3)


def gp_velocity(X, vf_dict, nonrigid_only=False, *, dtype=None, device=None):
    """GPU ``_gp_velocity`` (``gaussian_process.py:102-127``)."""
    X = np.asarray(X, dtype=np.float64)
    one = X.ndim == 1
    XX = X[None, :] if one else X
    v = _gp_eval(XX, vf_dict, _lib.EVAL_V, nonrigid_only, dtype, device)[_lib.EVAL_V][:, : XX.shape[1]]
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


from ._trajectory import genesis_states, integrate_field  # noqa: E402,F401  (imports gp helpers from this module)
