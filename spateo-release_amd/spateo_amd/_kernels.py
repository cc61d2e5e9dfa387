"""Thin Python layer over the libmvf C ABI: torch supplies device memory and the HIP stream, nothing else.

Every method launches hand-written HIP kernels asynchronously on torch's current stream of ``device`` and returns
device tensors.  Nothing here computes on the host and nothing falls back to torch/NumPy arithmetic.
"""
from __future__ import annotations

import numpy as np
import torch

from . import _lib

_DT = {"float32": (torch.float32, _lib.MVF_F32), "float64": (torch.float64, _lib.MVF_F64)}
_NP = {torch.float64: np.float64, torch.float32: np.float32, torch.int32: np.int32, torch.int64: np.int64}


def _ptr(t):
    return None if t is None else t.data_ptr()


def _on_device(fn):
    """Run a kernel method with the instance's GPU as the thread's current HIP device: libmvf launches on the stream it
    is handed, but hipFuncSetAttribute / hipMemsetAsync / occupancy queries inside it act on the CURRENT device, which
    need not be `self.device` (SparseVFC(device="cuda:1") without set_device, SparseVFC_many threads)."""
    import functools

    @functools.wraps(fn)
    def wrapped(self, *a, **kw):
        if torch.cuda.current_device() == self._dev_index:  # the usual case: no context switch (microseconds per launch,
            return fn(self, *a, **kw)                        # exposed at the head of a sub-millisecond EM iteration)
        with torch.cuda.device(self.device):
            return fn(self, *a, **kw)

    return wrapped


class HipKernels:
    """libmvf kernels bound to one GPU and one cell dtype ("float32" | "float64")."""

    def __init__(self, device=None, dtype="float32"):
        if not torch.cuda.is_available():
            raise RuntimeError(
                "spateo_amd needs an AMD GPU (HIP device) - torch.cuda.is_available() is False and there is no "
                "CPU fallback for the SparseVFC path."
            )
        self.lib = _lib.load()
        self.device = torch.device(device if device is not None else f"cuda:{torch.cuda.current_device()}")
        self._dev_index = self.device.index if self.device.index is not None else torch.cuda.current_device()
        self._gram_need = {}
        if dtype not in _DT:
            raise ValueError(f"dtype must be 'float32' or 'float64', got {dtype!r}")
        self.dtype_name = dtype
        self.tdtype, self.cdtype = _DT[dtype]
        self._gram_ws = None
        self._solve_ws = None
        self._mn_ws = None
        self._lr_ws = None
        self._mins = torch.empty(_lib.MVF_ESTEP_MIN_DOUBLES, dtype=torch.float64, device=self.device)
        self._scratch = None  # per-workgroup partials of the deterministic reductions (apply / estep_p / quadform)
        # optional per-launch timing of the dominant (Gram MFMA) kernel: list of (start, end) torch events recorded
        # on the launch stream; bench.py sets this to [] to enable it
        self.gram_events = None
        self._ublk = None       # cached float32 kernel values (layout Ublk[m/16][n][16]) for the Gram kernel
        self._ublk_key = None
    # ------------------------------------------------------------------ helpers
    def _stream(self):
        return torch.cuda.current_stream(self.device).cuda_stream

    @_on_device
    def to_x4(self, arr, center=None):
        """Host (n, d<=3) float64 array -> device (n, 4) tensor of the cell dtype (zero padded), optionally centred."""
        a = np.asarray(arr, dtype=np.float64)
        if a.ndim != 2 or not (1 <= a.shape[1] <= 3):
            raise ValueError(f"expected an (n, d) array with d <= 3, got shape {a.shape}")
        c = None if center is None else np.asarray(center, dtype=np.float64)[None, : a.shape[1]]
        return self.h2d_padded(a, 4, self.tdtype, minus=c)

    # pinned staging: host arrays up to this size travel through page-locked tensors of torch's caching host allocator
    # (one DMA, no driver-side bounce copies); larger ones keep the pageable path so that a fit of 8 M cells does not
    # leave hundreds of MB of the host page-locked for the life of the process
    PINNED_MAX_BYTES = 128 << 20
    DEVICE_PAD_MIN_ROWS = 4096   # from here on h2d_padded ships the raw rows and pads / centres / casts on the device

    def h2d(self, a, tdtype=None):
        """Host array -> device tensor (optionally cast) THROUGH page-locked staging.  Never `torch.from_numpy(a).to(device)`:
        a copy from a pageable range the runtime has not seen before costs 18 - 28 ms on this stack whatever its size (a
        500 x 3 array: 28 ms; measured with the caller keeping its results, i.e. every array of the next call at a fresh
        address - a second `SparseVFC` call cost 60 ms instead of 26).  Arrays above PINNED_MAX_BYTES keep the pageable path:
        at their size the fixed cost does not matter."""
        a = np.ascontiguousarray(a)
        t = torch.from_numpy(a)
        dt = tdtype or t.dtype
        if 0 < t.numel() * t.element_size() <= self.PINNED_MAX_BYTES:
            host = torch.empty(t.shape, dtype=dt, pin_memory=True)
            # (NumPy's single-threaded copy / cast, not torch's: a torch CPU op fans out over every core it sees - 128 threads on
            # the test boxes, whose containers hold a 16-CPU quota - and the spinning pool gets the whole process throttled for
            # the rest of the scheduler period: 50 - 90 ms stalls every few calls, `nr_throttled` in /sys/fs/cgroup/cpu.stat)
            np.copyto(host.numpy(), a, casting="unsafe")
            return host.to(self.device, non_blocking=True)
        if t.numel() == 0:
            return t.to(self.device).to(dt)
        # larger arrays: the same, streamed through two page-locked 32 MB staging buffers (the cast, if any, on the device)
        flat = a.reshape(-1)
        out = torch.empty(flat.shape, dtype=t.dtype, device=self.device)
        per = max(1, (32 << 20) // a.itemsize)
        stage = [torch.empty(per, dtype=t.dtype, pin_memory=True) for _ in range(2)]
        free = [None, None]
        stream = torch.cuda.current_stream(self.device)
        for i, lo in enumerate(range(0, flat.shape[0], per)):
            hi, b = min(lo + per, flat.shape[0]), i & 1
            if free[b] is not None:
                free[b].synchronize()
            np.copyto(stage[b].numpy()[: hi - lo], flat[lo:hi])
            out[lo:hi].copy_(stage[b][: hi - lo], non_blocking=True)
            free[b] = torch.cuda.Event()
            free[b].record(stream)
        out = out.reshape(t.shape)
        return out if dt == t.dtype else out.to(dt)

    def h2d_padded(self, a, width, tdtype, minus=None):
        """Host (n, d <= width) float64 array -> device (n, width) tensor of `tdtype`, zero padded.  minus (1 x d float64, may
        be None): subtracted in float64 on the way - the difference is rounded to `tdtype` as it is written into the staging
        buffer, ONE pass over the data (the evaluator API used to make a centred float64 temporary first)."""
        n, d = a.shape
        nbytes = n * width * (4 if tdtype == torch.float32 else 8)

        def fill(hv):
            if minus is None:
                hv[:, :d] = a
            else:
                np.subtract(a, minus, out=hv[:, :d], casting="same_kind")
            hv[:, d:] = 0

        if n >= self.DEVICE_PAD_MIN_ROWS and a.dtype == np.float64 and 0 < n * d * 8 <= self.PINNED_MAX_BYTES:
            # many rows: the RAW float64 rows travel (one contiguous copy into page-locked staging, one DMA) and the device
            # centres, casts and pads - the same float64 subtraction and the same rounding, i.e. the same bits.  NumPy's strided
            # subtract-and-cast into the padded staging buffer ran at 5 GB/s: 1.2 ms of the 2.5 ms the Jacobian + curl calls on
            # the 64^3 grid cost at the API, 1.1 ms per array of a 250 k-cell fit.
            host = torch.empty((n, d), dtype=torch.float64, pin_memory=True)
            np.copyto(host.numpy(), a)
            xd = host.to(self.device, non_blocking=True)
            if minus is not None:
                xd -= torch.tensor(np.asarray(minus, dtype=np.float64).reshape(-1)[:d].tolist(), dtype=torch.float64,
                                   device=self.device)
            out = torch.zeros((n, width), dtype=tdtype, device=self.device)
            out[:, :d] = xd
            return out
        if 0 < nbytes <= self.PINNED_MAX_BYTES and not (a.dtype == np.float64 and n >= self.DEVICE_PAD_MIN_ROWS):
            host = torch.empty((n, width), dtype=tdtype, pin_memory=True)
            fill(host.numpy())
            return host.to(self.device, non_blocking=True)
        if a.dtype == np.float64 and n >= self.DEVICE_PAD_MIN_ROWS:
            # millions of rows (8 M cells: 192 MB raw): the same device-side centring / cast / padding, the raw rows streamed
            # through two page-locked 32 MB staging buffers (NumPy's strided cast into a pageable (n, 4) buffer + a pageable
            # upload cost 70 ms per array at 8 M rows and left the runtime a fresh pageable range to digest)
            out = torch.zeros((n, width), dtype=tdtype, device=self.device)
            sub = None if minus is None else torch.tensor(np.asarray(minus, dtype=np.float64).reshape(-1)[:d].tolist(),
                                                          dtype=torch.float64, device=self.device)
            rows = max(1, (32 << 20) // (d * 8))
            stage = [torch.empty((rows, d), dtype=torch.float64, pin_memory=True) for _ in range(2)]
            free = [None, None]  # event after which a staging buffer may be refilled
            stream = torch.cuda.current_stream(self.device)
            for i, lo in enumerate(range(0, n, rows)):
                hi, b = min(lo + rows, n), i & 1
                if free[b] is not None:
                    free[b].synchronize()
                np.copyto(stage[b].numpy()[: hi - lo], a[lo:hi])
                xd = stage[b][: hi - lo].to(self.device, non_blocking=True)
                free[b] = torch.cuda.Event()
                free[b].record(stream)
                out[lo:hi, :d] = xd if sub is None else xd - sub
            return out
        buf = np.empty((n, width), dtype=np.float32 if tdtype == torch.float32 else np.float64)
        fill(buf)
        return torch.from_numpy(buf).to(self.device)

    def d2h(self, t):
        """Device tensor -> ordinary (pageable) host NumPy array, THROUGH page-locked staging: the runtime never sees the
        destination.  A `.cpu()` into a pageable range it has not seen before stalls for 50 - 100 ms every few calls on this
        stack (and so does the unmapping of such a range when the caller drops the array): 6 MB of unique rows cost 1.4 ms
        or 90 (tools/kept_results_probe.py)."""
        if not t.is_cuda:
            return t.numpy()
        return self.to_host([t], own_pinned=False)[0]

    @_on_device
    def to_host(self, tensors, own_pinned=True):
        """Device tensors -> host NumPy arrays with ONE stream synchronisation: page-locked destinations while the total
        stays under PINNED_MAX_BYTES (own_pinned: the arrays returned own their pinned block until they are garbage
        collected; else they are pageable copies of it and the block goes back to torch's host cache), otherwise pageable
        arrays filled through two page-locked 32 MB staging buffers."""
        tensors = [t.contiguous() for t in tensors]
        total = sum(t.numel() * t.element_size() for t in tensors)
        stream = torch.cuda.current_stream(self.device)
        if total <= self.PINNED_MAX_BYTES:
            hosts = [torch.empty(t.shape, dtype=t.dtype, pin_memory=True) for t in tensors]
            for h, t in zip(hosts, tensors):
                h.copy_(t, non_blocking=True)
            stream.synchronize()
            return [h.numpy() if own_pinned else h.numpy().copy() for h in hosts]
        outs = []
        chunk = 32 << 20
        stage = [torch.empty(chunk, dtype=torch.uint8, pin_memory=True) for _ in range(2)]
        events = [torch.cuda.Event(), torch.cuda.Event()]
        for t in tensors:
            out = np.empty(tuple(t.shape), dtype=torch.empty(0, dtype=t.dtype).numpy().dtype)
            flat_d = t.view(-1).view(torch.uint8)
            flat_h = out.reshape(-1).view(np.uint8)
            nb = flat_d.numel()
            pending = []  # (buffer index, lo, hi) in flight
            for i, lo in enumerate(range(0, nb, chunk)):
                b = i & 1
                if len(pending) == 2:
                    pb, plo, phi = pending.pop(0)
                    events[pb].synchronize()
                    flat_h[plo:phi] = stage[pb][: phi - plo].numpy()
                hi = min(lo + chunk, nb)
                stage[b][: hi - lo].copy_(flat_d[lo:hi], non_blocking=True)
                events[b].record(stream)
                pending.append((b, lo, hi))
            for pb, plo, phi in pending:
                events[pb].synchronize()
                flat_h[plo:phi] = stage[pb][: phi - plo].numpy()
            outs.append(out)
        return outs

    def _red(self, n, at_least=0):
        """Scratch of the deterministic reductions: mvf_reduce_scratch_doubles(n) float64, and never fewer than
        `at_least` (mvf_quadform writes one partial per control point: m float64, mvf.h)."""
        need = max(int(self.lib.mvf_reduce_scratch_doubles(int(n))), int(at_least))
        if self._scratch is None or self._scratch.numel() < need:
            self._scratch = torch.empty(need, dtype=torch.float64, device=self.device)
        return self._scratch.data_ptr()

    def empty(self, *shape, dtype=None):
        return torch.empty(*shape, dtype=dtype or self.tdtype, device=self.device)

    def zeros(self, *shape, dtype=None):
        return torch.zeros(*shape, dtype=dtype or self.tdtype, device=self.device)

    # ------------------------------------------------------------------ kernels
    @_on_device
    def unique_rows(self, X):
        """np.unique(X, axis=0, return_index=True) for a finite host float64 (n, d) array, on the device:
        returns host (unique rows sorted lexicographically, index of the first occurrence of each)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        n, d = X.shape
        xd = self.h2d(X)
        uid = torch.empty(n, dtype=torch.int64, device=self.device)
        rows = torch.empty(n, d, dtype=torch.float64, device=self.device)
        cnt = torch.zeros(1, dtype=torch.int64, device=self.device)
        need = int(self.lib.mvf_unique_rows_workspace_bytes(n, d))
        ws = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.mvf_unique_rows(_ptr(xd), n, d, _ptr(uid), _ptr(rows), _ptr(cnt), _ptr(ws), ws.numel(),
                                            self._stream()), "mvf_unique_rows")
        k = int(cnt.cpu()[0])
        # (page-locked arrays, no second copy: the callers gather from them and drop them - preprocess.unique_rows' result never
        # reaches the user as is)
        S, idx = self.to_host([rows[:k], uid[:k]])
        return S, idx

    @_on_device
    def knn_mean_distance(self, X, k):
        """Mean distance from each point to its k - 1 nearest other points (host float64 (m, d) in, float out): the
        neighbour search of dynamo's bandwidth_selector on the device (m <= 8192, d <= 8)."""
        X = np.ascontiguousarray(X, dtype=np.float64)
        m, d = X.shape
        xd = self.h2d(X)
        rows = torch.empty(m, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.mvf_knn_rowsum(_ptr(xd), m, d, int(k), _ptr(rows), self._stream()), "mvf_knn_rowsum")
        return float(np.sum(self.d2h(rows)) / (m * (k - 1)))

    @_on_device
    def con_k(self, x, y, beta, return_d=False, dtype=None):
        """x: (n, d), y: (m, d) device tensors (cell dtype, or `dtype`) -> K (n, m) [and D (n, d, m)]."""
        tdtype, cdtype = _DT[dtype] if dtype is not None else (self.tdtype, self.cdtype)
        if x.dtype != tdtype or y.dtype != tdtype:
            raise TypeError(f"con_k: expected {tdtype} tensors, got {x.dtype} / {y.dtype}")
        n, d = x.shape
        m = y.shape[0]
        x, y = x.contiguous(), y.contiguous()
        K = self.empty(n, m, dtype=tdtype)
        if return_d:
            D = self.empty(n, d, m, dtype=tdtype)
            for lo in range(0, n, 65535):  # mvf_con_k_d takes at most 65535 rows per call: chunk (rows are contiguous)
                nn = min(65535, n - lo)
                _lib.check(self.lib.mvf_con_k_d(_ptr(x[lo:]), nn, _ptr(y), m, d, float(beta), _ptr(K[lo:]), _ptr(D[lo:]),
                                                cdtype, self._stream()), "mvf_con_k_d")
            return K, D
        _lib.check(self.lib.mvf_con_k(_ptr(x), n, _ptr(y), m, d, float(beta), _ptr(K), cdtype, self._stream()),
                   "mvf_con_k")
        return K

    @_on_device
    def apply(self, x4, ctrl4, beta, C, y4=None, P=None, stats=None):
        """V4 = con_K(x, ctrl) @ C; with y4 also r = ||y - V||^2 and stats[0] += sum P r.  Returns (V4, r)."""
        n, m = x4.shape[0], ctrl4.shape[0]
        V4 = self.empty(n, 4)
        r = self.empty(n) if y4 is not None else None
        _lib.check(self.lib.mvf_apply(_ptr(x4), n, _ptr(ctrl4), m, float(beta), _ptr(C), _ptr(V4), _ptr(y4), _ptr(P),
                                      _ptr(r), _ptr(stats), self._red(n), self.cdtype, self._stream()), "mvf_apply")
        return V4, r

    @_on_device
    def estep_min(self, r, sigma2):
        """Returns a device float64 view (2,): [min non-zero t1, #zeros]."""
        _lib.check(self.lib.mvf_estep_min(_ptr(r), r.shape[0], float(sigma2), _ptr(self._mins), self.cdtype,
                                          self._stream()), "mvf_estep_min")
        return self._mins[:2]

    @_on_device
    def estep_p(self, r, sigma2, gamma, a, dy, minP, theta, zero_fill, P_out, stats):
        """zero_fill: a host float, or a device float64 tensor whose first element is the fill (estep_min's result,
        possibly MIN-all-reduced): then the two E-step phases chain on the stream without a host round trip."""
        fill_dev = zero_fill if torch.is_tensor(zero_fill) else None
        _lib.check(self.lib.mvf_estep_p(_ptr(r), r.shape[0], float(sigma2), float(gamma), float(a), int(dy),
                                        float(minP), float(theta), 0.0 if fill_dev is not None else float(zero_fill),
                                        _ptr(fill_dev), _ptr(P_out), _ptr(stats),
                                        self._red(r.shape[0]), self.cdtype, self._stream()), "mvf_estep_p")

    def read_host(self, t):
        """A contiguous device tensor as a host NumPy array, through ONE blocking copy on the current stream."""
        out = np.empty(t.shape, dtype=_NP[t.dtype])
        _lib.check(self.lib.mvf_read_back(out.ctypes.data, _ptr(t), out.nbytes, self._stream()), "mvf_read_back")
        return out

    @_on_device
    def estep(self, r, sigma2, gamma, a, dy, minP, theta, P_out, stats):
        """Both phases of the E-step in one call (one process: no MIN all-reduce between them); `stats` is overwritten."""
        _lib.check(self.lib.mvf_estep(_ptr(r), r.shape[0], float(sigma2), float(gamma), float(a), int(dy), float(minP),
                                      float(theta), _ptr(self._mins), _ptr(P_out), _ptr(stats), self._red(r.shape[0]),
                                      self.cdtype, self._stream()), "mvf_estep")

    def ublk_bytes(self, n, m):
        return int(self.lib.mvf_ublk_bytes(n, m, self.cdtype))

    @_on_device
    def build_ublk(self, x4, ctrl4, beta):
        """Materialise the float32 kernel values once per fit (U is constant across EM iterations); later `gram`
        calls with the same (x4, ctrl4, beta) stream them instead of regenerating them."""
        n, m = x4.shape[0], ctrl4.shape[0]
        need = self.ublk_bytes(n, m)
        self._ublk = None
        self._ublk = torch.empty(need // self._ublk_itemsize(), dtype=self.tdtype, device=self.device)
        _lib.check(self.lib.mvf_ublk_build(_ptr(x4), n, _ptr(ctrl4), m, float(beta), _ptr(self._ublk), need,
                                           self.cdtype, self._stream()), "mvf_ublk_build")
        self._ublk_key = (x4.data_ptr(), ctrl4.data_ptr(), n, m, float(beta))

    def _ublk_itemsize(self):
        return 4 if self.tdtype == torch.float32 else 8

    # ---- wide right-hand sides (Dy > 3): one pass over the cached kernel values for all columns (mvf_wide.hip) ----
    @staticmethod
    def wide_pads(n, m):
        """(cells, control points, columns -> padded extents) of the buffers mvf_rhs_cached / mvf_apply_cached read."""
        return -(-int(n) // 256) * 256, -(-int(m) // 128) * 128

    def _wide_ws(self, n, m):
        need = int(self.lib.mvf_wide_workspace_bytes(int(n), int(m)))
        if getattr(self, "_wide_buf", None) is None or self._wide_buf.numel() < need:
            self._wide_buf = None
            self._wide_buf = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
        return self._wide_buf

    @_on_device
    def rhs_wide(self, P, Yd, dy, m, R):
        """R (m x dy float64) = U^T diag(P) Yd[:, :dy] from the cache of build_ublk (Yd: padded rows x padded columns)."""
        if self._ublk is None:
            raise RuntimeError("rhs_wide needs the kernel-value cache (build_ublk)")
        n = P.shape[0]
        ws = self._wide_ws(n, m)
        _lib.check(self.lib.mvf_rhs_cached(_ptr(self._ublk), _ptr(P), _ptr(Yd), n, int(m), int(dy), Yd.shape[1], _ptr(R),
                                           R.shape[1], _ptr(ws), ws.numel(), self.cdtype, self._stream()), "mvf_rhs_cached")

    @_on_device
    def apply_wide(self, Cd, dy, m, Yd, P, Vd, r, stats):
        """Vd[:, :dy] = U Cd[:, :dy]; r = sum_d (Yd - Vd)^2; stats[0] += sum P r (Cd: padded rows x padded columns, float64)."""
        if self._ublk is None:
            raise RuntimeError("apply_wide needs the kernel-value cache (build_ublk)")
        n = Vd.shape[0]
        ws = self._wide_ws(n, m)
        _lib.check(self.lib.mvf_apply_cached(_ptr(self._ublk), n, int(m), _ptr(Cd), Cd.shape[1], int(dy), _ptr(Yd), Yd.shape[1],
                                             _ptr(P), _ptr(Vd), _ptr(r), _ptr(stats), _ptr(ws), ws.numel(), self.cdtype,
                                             self._stream()), "mvf_apply_cached")

    def drop_ublk(self):
        self._ublk = None
        self._ublk_key = None

    @_on_device
    def gram(self, x4, P, y4, ctrl4, beta, G, R, rhs_only=False, tiles_only=False):
        """G = U^T P U (m x m), R = U^T P Y (m x 3).  rhs_only: only R for this y4 (G unchanged) - for Y wider than 3,
        and the second half of a multi-rank step; tiles_only: only G (tile stage + its reduction) - the first half of a
        multi-rank step, whose all-reduce of G then overlaps the rhs kernels."""
        n, m = x4.shape[0], ctrl4.shape[0]
        key = (n, m, _lib.OPTION_EPOCH[0])
        need = self._gram_need.get(key)
        if need is None:  # (the call runs the launch plan's search: once per shape, not once per EM iteration)
            need = self._gram_need[key] = self.lib.mvf_gram_workspace_bytes(n, m, self.cdtype)
        if self._gram_ws is None or self._gram_ws.numel() < need:
            self._gram_ws = None
            self._gram_ws = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
        args = (_ptr(x4), _ptr(P), _ptr(y4), n, _ptr(ctrl4), m, float(beta), _ptr(G), _ptr(R), _ptr(self._gram_ws),
                self._gram_ws.numel())
        cached = self._ublk is not None and self._ublk_key == (x4.data_ptr(), ctrl4.data_ptr(), n, m, float(beta))
        if cached:
            def run(stages):
                _lib.check(self.lib.mvf_gram_cached(stages, _ptr(self._ublk), *args, self.cdtype, self._stream()),
                           "mvf_gram_cached")
        else:
            def run(stages):
                _lib.check(self.lib.mvf_gram_stages(stages, *args, self.cdtype, self._stream()), "mvf_gram_stages")
        if rhs_only:
            run(_lib.GRAM_RHS | _lib.GRAM_REDUCE_RHS)
            return
        rest = _lib.GRAM_REDUCE if tiles_only else _lib.GRAM_RHS | _lib.GRAM_REDUCE | _lib.GRAM_REDUCE_RHS
        if self.gram_events is None:
            run(_lib.GRAM_TILES | rest)
            return
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        run(_lib.GRAM_TILES)
        e1.record()
        self.gram_events.append((e0, e1))
        run(rest)

    @_on_device
    def solve(self, G, K, lambda_sigma2, jitter, R, C_out, info, pivots=None):
        """Cholesky solve of (G + ls2 K + jitter mean(diag) I) C = R.  info[0] != 0: non-positive pivot.
        pivots (float64[2], optional): [min L_jj^2, max L_jj^2] - the host's numerical-rank certificate."""
        m, nrhs = R.shape
        need = self.lib.mvf_solve_workspace_bytes(m, nrhs)
        if self._solve_ws is None or self._solve_ws.numel() < need:
            self._solve_ws = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.mvf_solve(_ptr(G), _ptr(K), float(lambda_sigma2), float(jitter), _ptr(R), m, nrhs,
                                      _ptr(C_out), _ptr(info), _ptr(pivots), _ptr(self._solve_ws),
                                      self._solve_ws.numel(), self._stream()), "mvf_solve")

    @_on_device
    def solve_minnorm(self, G, K, lambda_sigma2, shift, R, C_out, info, einfo, rcond=None, reuse=False, max_sweeps=60,
                      basis=None, warm=False):
        """Minimum-norm solve with the gelsd cut-off (eigenvalues below rcond * max|lambda| dropped; rcond = float64
        eps = scipy.linalg.lstsq's default).  reuse=True applies the decomposition left in the workspace by the
        previous call (same matrix) to another right-hand side.  ``basis`` (float64 tensor from ``minnorm_basis``)
        receives the eigenvectors; ``warm=True`` starts from the ones it holds (the previous EM iteration's).
        Synchronises the stream (once per Jacobi sweep)."""
        m, nrhs = R.shape
        need = self.lib.mvf_solve_minnorm_workspace_bytes(m, nrhs)
        if self._mn_ws is None or self._mn_ws.numel() < need:
            if reuse:
                raise RuntimeError("solve_minnorm(reuse=True) without a previous decomposition")
            self._mn_ws = None
            self._mn_ws = torch.empty(max(need, 1), dtype=torch.uint8, device=self.device)
        rc = float(np.finfo(np.float64).eps) if rcond is None else float(rcond)
        _lib.check(self.lib.mvf_solve_minnorm(_ptr(G), _ptr(K), float(lambda_sigma2), float(shift), rc, _ptr(R), m,
                                              nrhs, _ptr(C_out), _ptr(info), _ptr(einfo), int(max_sweeps),
                                              1 if reuse else 0, _ptr(basis), 1 if warm else 0, _ptr(self._mn_ws),
                                              self._mn_ws.numel(), self._stream()), "mvf_solve_minnorm")

    @_on_device
    def pinv_diag(self, x4, ctrl4, beta, rcond=None, lowrank=False):
        """diag(U pinv(A) U^T) (float64, n) from the decomposition the last solve_minnorm_lr (lowrank=True) / solve_minnorm
        call of this object left in its workspace; U = con_K(x, ctrl, beta) is regenerated, never materialised."""
        ws = self._lr_ws if lowrank else self._mn_ws
        if ws is None:
            raise RuntimeError("pinv_diag without a previous decomposition")
        n, m = x4.shape[0], ctrl4.shape[0]
        out = torch.empty(n, dtype=torch.float64, device=self.device)
        rc = float(np.finfo(np.float64).eps) if rcond is None else float(rcond)
        _lib.check(self.lib.mvf_pinv_diag(_ptr(x4), n, _ptr(ctrl4), m, float(beta), rc, 1 if lowrank else 0, _ptr(out),
                                          _ptr(ws), ws.numel(), self.cdtype, self._stream()), "mvf_pinv_diag")
        return out

    @_on_device
    def solve_minnorm_lr(self, G, K, lambda_sigma2, R, C_out, info, einfo, rcond=None, reuse=False, max_sweeps=60,
                         rank_hint=0, tolf=0.25, deflate=False):
        """The same truncated minimum-norm solve through the rank-revealing factor (pivoted Cholesky stopped at
        tolf * eps * lambda_max, Jacobi on the r kept columns only).  einfo[6] = r; pass it back as ``rank_hint`` for
        the next, nearby matrix.  deflate=True (mvf_solve_minnorm_lrd): only the invariant subspace below the cut-off
        is computed and projected out - same truncation, no full eigendecomposition (no pinv_diag afterwards).
        Synchronises the stream."""
        m, nrhs = R.shape
        fn = self.lib.mvf_solve_minnorm_lrd if deflate else self.lib.mvf_solve_minnorm_lr
        need = (self.lib.mvf_solve_minnorm_lrd_workspace_bytes if deflate else self.lib.mvf_solve_minnorm_lr_workspace_bytes)(m, nrhs)
        if self._lr_ws is None or self._lr_ws.numel() < need:
            if reuse:
                raise RuntimeError("solve_minnorm_lr(reuse=True) without a previous decomposition")
            self._lr_ws = None
            # (zeroed: the 64-byte state record in it must not be read as a finished factorisation - ADVICE r5; once per allocation)
            self._lr_ws = torch.zeros(max(need, 1), dtype=torch.uint8, device=self.device)
        rc = float(np.finfo(np.float64).eps) if rcond is None else float(rcond)
        _lib.check(fn(_ptr(G), _ptr(K), float(lambda_sigma2), float(tolf), rc, _ptr(R), m, nrhs, _ptr(C_out), _ptr(info),
                      _ptr(einfo), int(max_sweeps), 1 if reuse else 0, int(rank_hint), _ptr(self._lr_ws),
                      self._lr_ws.numel(), self._stream()), "mvf_solve_minnorm_lrd" if deflate else "mvf_solve_minnorm_lr")

    @_on_device
    def solve_minnorm_lrd_async(self, G, K, lambda_sigma2, R, C_out, info, einfo, form_hint, rcond=None, tolf=0.25):
        """The direct form of the deflated solve without any host synchronisation (mvf_solve_minnorm_lrd_async): form_hint =
        the previous call's einfo[8] on this workspace (1 factor form / 2 direct form, with einfo[6] == m).  einfo[9] == 1
        afterwards means NOT accepted: repeat through solve_minnorm_lr(deflate=True).  Needs the workspace of a previous call."""
        m, nrhs = R.shape
        if self._lr_ws is None or self._lr_ws.numel() < self.lib.mvf_solve_minnorm_lrd_workspace_bytes(m, nrhs):
            raise RuntimeError("solve_minnorm_lrd_async without the workspace of a previous deflated solve")
        rc = float(np.finfo(np.float64).eps) if rcond is None else float(rcond)
        _lib.check(self.lib.mvf_solve_minnorm_lrd_async(_ptr(G), _ptr(K), float(lambda_sigma2), float(tolf), rc, _ptr(R), m, nrhs,
                                                        _ptr(C_out), _ptr(info), _ptr(einfo), int(form_hint), _ptr(self._lr_ws),
                                                        self._lr_ws.numel(), self._stream()), "mvf_solve_minnorm_lrd_async")

    @_on_device
    def lr_pivot_order(self, m, with_values=False):
        """Host int array: the pivots (control-point indices, in the order taken) of the last solve_minnorm_lr call; with
        with_values also (the diagonal value of each pivot when it was taken, the stopping tolerance)."""
        import ctypes

        if self._lr_ws is None:
            raise RuntimeError("lr_pivot_order without a previous solve_minnorm_lr")
        order = (ctypes.c_int * int(m))()
        vals = (ctypes.c_double * int(m))()
        tol = ctypes.c_double(0.0)
        r = ctypes.c_int64(0)
        _lib.check(self.lib.mvf_lr_pivot_order(_ptr(self._lr_ws), self._lr_ws.numel(), int(m), order, vals, ctypes.byref(tol),
                                               ctypes.byref(r), self._stream()), "mvf_lr_pivot_order")
        idx = np.frombuffer(order, dtype=np.int32, count=int(r.value)).astype(np.int64)
        if with_values:
            return idx, np.frombuffer(vals, dtype=np.float64, count=int(r.value)).copy(), float(tol.value)
        return idx

    @_on_device
    def minnorm_basis(self, m):
        """Uninitialised eigenvector-basis buffer for solve_minnorm's warm start (m x m padded to multiples of 64)."""
        return torch.empty(int(self.lib.mvf_solve_minnorm_basis_bytes(m)) // 8, dtype=torch.float64, device=self.device)

    @_on_device
    def lincomb3(self, out, a, A, b=0.0, B=None, c=0.0, C=None):
        """out = a A + b B + c C (float64 device tensors of one shape; B, C optional; out may alias an input)."""
        _lib.check(self.lib.mvf_lincomb3(_ptr(out), float(a), _ptr(A), float(b), _ptr(B), float(c), _ptr(C),
                                         out.numel(), self._stream()), "mvf_lincomb3")
        return out

    @_on_device
    def hull_mask(self, points, equations, tol):
        """Host (n, 3) points and SciPy ConvexHull.equations (nf, 4) -> host bool (n,): inside the hull."""
        P = self.h2d(np.ascontiguousarray(points, dtype=np.float64))
        E = self.h2d(np.ascontiguousarray(equations, dtype=np.float64))
        out = torch.empty(P.shape[0], dtype=torch.uint8, device=self.device)
        _lib.check(self.lib.mvf_hull_mask(_ptr(P), P.shape[0], _ptr(E), E.shape[0], float(tol), _ptr(out),
                                          self._stream()), "mvf_hull_mask")
        return self.d2h(out).astype(bool)

    @_on_device
    def quadform(self, K, C, out):
        _lib.check(self.lib.mvf_quadform(_ptr(K), _ptr(C), K.shape[0], C.shape[1], _ptr(out),
                                         self._red(K.shape[0], at_least=K.shape[0]), self._stream()), "mvf_quadform")

    @_on_device
    def sym_pack(self, G, tri):
        """Packed upper triangle of the symmetric G (what the multi-GPU host all-reduces)."""
        _lib.check(self.lib.mvf_sym_pack(_ptr(G), G.shape[0], _ptr(tri), self._stream()), "mvf_sym_pack")

    @_on_device
    def sym_unpack(self, tri, G):
        _lib.check(self.lib.mvf_sym_unpack(_ptr(tri), G.shape[0], _ptr(G), self._stream()), "mvf_sym_unpack")

    @staticmethod
    def _affine_buf(affine):
        if affine is None:
            return None
        import ctypes

        alpha, jmul, A, b = affine
        al = np.broadcast_to(np.asarray(alpha, dtype=np.float64).reshape(-1), (3,))  # a scalar or one value per component
        vals = [float(x) for x in al] + [float(jmul)] + [float(x) for x in np.asarray(A, dtype=np.float64).reshape(9)] + \
               [float(x) for x in np.asarray(b, dtype=np.float64).reshape(3)]
        return (ctypes.c_double * 16)(*vals)

    @_on_device
    def integrate(self, x4, ctrl4, beta, C, dt, substeps, n_out, affine=None):
        """RK4 trajectories of dx/dt = v(x): returns a float64 device tensor (n, n_out, 3)."""
        n, m = x4.shape[0], ctrl4.shape[0]
        traj = torch.empty(n, n_out, 3, dtype=torch.float64, device=self.device)
        _lib.check(self.lib.mvf_integrate(_ptr(x4), n, _ptr(ctrl4), m, float(beta), _ptr(C), self._affine_buf(affine),
                                          float(dt), int(substeps), int(n_out), _ptr(traj), self.cdtype,
                                          self._stream()), "mvf_integrate")
        return traj

    @_on_device
    def eval(self, x4, ctrl4, beta, C, flags, affine=None):
        """Fused evaluator.  Returns a dict of float64 device tensors for the requested MVF_EVAL_* flags.
        `affine` = (alpha, jmul, A (3x3), b (3)) applies v = alpha K@C + A q + b, J = jmul J (GP variant)."""
        n, m = x4.shape[0], ctrl4.shape[0]
        f64 = torch.float64
        out = {}

        def buf(flag, *shape):
            if flags & flag:
                t = torch.empty(*shape, dtype=f64, device=self.device)
                out[flag] = t
                return t
            return None

        v = buf(_lib.EVAL_V, n, 3)
        jac = buf(_lib.EVAL_JAC, 3, 3, n)
        div = buf(_lib.EVAL_DIV, n)
        curl = buf(_lib.EVAL_CURL, n, 3)
        acc = buf(_lib.EVAL_ACC, n, 3)
        curv = buf(_lib.EVAL_CURV, n, 3)
        tors = buf(_lib.EVAL_TORS, n, 3)
        jdet = buf(_lib.EVAL_JDET, n)
        aff = self._affine_buf(affine)
        _lib.check(self.lib.mvf_eval_affine(_ptr(x4), n, _ptr(ctrl4), m, float(beta), _ptr(C), aff, int(flags), _ptr(v),
                                            _ptr(jac), _ptr(div), _ptr(curl), _ptr(acc), _ptr(curv), _ptr(tors),
                                            _ptr(jdet), self.cdtype, self._stream()), "mvf_eval_affine")
        return out
