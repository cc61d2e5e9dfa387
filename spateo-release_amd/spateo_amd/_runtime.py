"""Process-level runtime of the package: the ONE seam that binds the HIP kernels (`_make_kernels`; tests/ replace it with a
NumPy test double to exercise the host logic without a GPU), the per-thread cache of kernel objects of the stateless entry
points, the default cell dtype, the whole-call profile switch and the `lstsq_method` policy.  Everything else imports from
here; nothing here imports the engine or the API modules."""
from __future__ import annotations

import threading

import torch

from ._kernels import HipKernels

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


def _d2h(k, t):
    """One device tensor -> ordinary host NumPy array (through page-locked staging on the GPU path: `HipKernels.d2h`)."""
    return k.d2h(t) if hasattr(k, "d2h") else t.cpu().numpy()


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

def clear_eval_cache():
    """Drop the evaluator results kept on the device by the last ``SvcVectorField`` / ``GPVectorField`` /
    ``vector_field_function`` call of this thread (one entry: the quantities of the last (points, field) pair, at most
    ``_EVAL_PREFETCH_CAP`` bytes when prefetched).  A new fit (``SparseVFCEngine``) drops it by itself before it sizes its
    kernel-value cache against the free HBM."""
    _TLS.__dict__.pop("fused", None)
