"""ctypes binding of ``libmvf.so`` (C ABI declared in ``include/mvf.h``).

The library is built in-tree by ``__graft_entry__.build()`` / ``make -C spateo-release_amd/csrc`` and is the ONLY
compute path of this package: there is no CPU fallback.  ``load()`` raises if the shared object is missing.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# Developer overrides are honoured ONLY under the explicit gate MVF_DEV_KNOBS=1 (and announced on stderr): MVF_LIB_PATH
# (A/B runs of two builds of the library) and the legacy MVF_* kernel knobs, which are translated into mvf_debug_option
# calls at load time.  Without the gate the environment changes nothing.
DEV_KNOBS = os.environ.get("MVF_DEV_KNOBS") == "1"
LIB_PATH = (DEV_KNOBS and os.environ.get("MVF_LIB_PATH")) or os.path.join(_HERE, "lib", "libmvf.so")
DEBUG_OPTIONS = ("conk_form", "slice_len", "solve_small_off", "lr_timing", "lr_no_deflate", "defl_block", "defl_apps", "lr_no_direct",
                 "direct_accept")
_LEGACY_ENV = {  # environment name -> (option, value parser)
    "MVF_CONK": ("conk_form", lambda v: {"rows": 1, "flat": 2, "2d": 3}[v]),
    "MVF_SLICE_LEN": ("slice_len", int),
    "MVF_SOLVE_SMALL": ("solve_small_off", lambda v: 1 if v.startswith("0") else 0),
    "MVF_LR_TIMING": ("lr_timing", lambda v: 1),
}

MVF_F32, MVF_F64 = 0, 1
MVF_ESTEP_MIN_DOUBLES = 4098

MVF_COMM_ID_BYTES = 128
RED_SUM, RED_MIN = 0, 1
GRAM_TILES, GRAM_RHS, GRAM_REDUCE, GRAM_REDUCE_RHS = 1, 2, 4, 8
EVAL_V, EVAL_JAC, EVAL_DIV, EVAL_CURL, EVAL_ACC, EVAL_CURV, EVAL_TORS, EVAL_JDET = 1, 2, 4, 8, 16, 32, 64, 128

_p, _i64, _i, _d, _sz = C.c_void_p, C.c_int64, C.c_int, C.c_double, C.c_size_t

# name -> (restype, argtypes); mirrors include/mvf.h one to one
SIGNATURES = {
    "mvf_last_error": (C.c_char_p, []),
    "mvf_version": (_i, []),
    "mvf_read_back": (_i, [_p, _p, C.c_size_t, _p]),
    "mvf_debug_option": (_i, [C.c_char_p, C.c_longlong]),
    "mvf_debug_option_get": (C.c_longlong, [C.c_char_p]),
    "mvf_device_count": (_i, [C.POINTER(C.c_int)]),
    "mvf_unique_rows_workspace_bytes": (_sz, [_i64, _i]),
    "mvf_unique_rows": (_i, [_p, _i64, _i, _p, _p, _p, _p, _sz, _p]),
    "mvf_knn_rowsum": (_i, [_p, _i64, _i, _i, _p, _p]),
    "mvf_hull_mask": (_i, [_p, _i64, _p, _i64, _d, _p, _p]),
    "mvf_con_k": (_i, [_p, _i64, _p, _i64, _i, _d, _p, _i, _p]),
    "mvf_con_k_d": (_i, [_p, _i64, _p, _i64, _i, _d, _p, _p, _i, _p]),
    "mvf_reduce_scratch_doubles": (_sz, [_i64]),
    "mvf_apply": (_i, [_p, _i64, _p, _i64, _d, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "mvf_estep_min": (_i, [_p, _i64, _d, _p, _i, _p]),
    "mvf_estep_p": (_i, [_p, _i64, _d, _d, _d, _i, _d, _d, _d, _p, _p, _p, _p, _i, _p]),
    "mvf_estep": (_i, [_p, _i64, _d, _d, _d, _i, _d, _d, _p, _p, _p, _p, _i, _p]),
    "mvf_gram_workspace_bytes": (_sz, [_i64, _i64, _i]),
    "mvf_gram": (_i, [_p, _p, _p, _i64, _p, _i64, _d, _p, _p, _p, _sz, _i, _p]),
    "mvf_gram_stages": (_i, [_i, _p, _p, _p, _i64, _p, _i64, _d, _p, _p, _p, _sz, _i, _p]),
    "mvf_ublk_bytes": (_sz, [_i64, _i64, _i]),
    "mvf_ublk_build": (_i, [_p, _i64, _p, _i64, _d, _p, _sz, _i, _p]),
    "mvf_gram_cached": (_i, [_i, _p, _p, _p, _p, _i64, _p, _i64, _d, _p, _p, _p, _sz, _i, _p]),
    "mvf_solve_workspace_bytes": (_sz, [_i64, _i]),
    "mvf_solve": (_i, [_p, _p, _d, _d, _p, _i64, _i, _p, _p, _p, _p, _sz, _p]),
    "mvf_solve_minnorm_workspace_bytes": (_sz, [_i64, _i]),
    "mvf_solve_minnorm": (_i, [_p, _p, _d, _d, _d, _p, _i64, _i, _p, _p, _p, _i, _i, _p, _i, _p, _sz, _p]),
    "mvf_solve_minnorm_basis_bytes": (_sz, [_i64]),
    "mvf_solve_minnorm_lr_workspace_bytes": (_sz, [_i64, _i]),
    "mvf_solve_minnorm_lr": (_i, [_p, _p, _d, _d, _d, _p, _i64, _i, _p, _p, _p, _i, _i, _i, _p, _sz, _p]),
    "mvf_wide_workspace_bytes": (_sz, [_i64, _i64]),
    "mvf_rhs_cached": (_i, [_p, _p, _p, _i64, _i64, _i, _i64, _p, _i64, _p, _sz, _i, _p]),
    "mvf_apply_cached": (_i, [_p, _i64, _i64, _p, _i64, _i, _p, _i64, _p, _p, _p, _p, _p, _sz, _i, _p]),
    "mvf_solve_minnorm_lrd_workspace_bytes": (_sz, [_i64, _i]),
    "mvf_solve_minnorm_lrd": (_i, [_p, _p, _d, _d, _d, _p, _i64, _i, _p, _p, _p, _i, _i, _i, _p, _sz, _p]),
    "mvf_solve_minnorm_lrd_async": (_i, [_p, _p, _d, _d, _d, _p, _i64, _i, _p, _p, _p, _i, _p, _sz, _p]),
    "mvf_lr_pivot_order": (_i, [_p, _sz, _i64, C.POINTER(C.c_int), C.POINTER(C.c_double), C.POINTER(C.c_double),
                                C.POINTER(C.c_int64), _p]),
    "mvf_pinv_diag": (_i, [_p, _i64, _p, _i64, _d, _d, _i, _p, _p, _sz, _i, _p]),
    "mvf_lincomb3": (_i, [_p, _d, _p, _d, _p, _d, _p, _i64, _p]),
    "mvf_quadform": (_i, [_p, _p, _i64, _i, _p, _p, _p]),
    "mvf_sym_pack": (_i, [_p, _i64, _p, _p]),
    "mvf_sym_unpack": (_i, [_p, _i64, _p, _p]),
    "mvf_comm_unique_id": (_i, [_p]),
    "mvf_comm_create": (_i, [C.POINTER(_p), _i, _i, _p]),
    "mvf_comm_destroy": (_i, [_p]),
    "mvf_comm_info": (_i, [_p, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "mvf_allreduce_stats": (_i, [_p, _p, _i64, _i, _p]),
    "mvf_eval": (_i, [_p, _i64, _p, _i64, _d, _p, _i, _p, _p, _p, _p, _p, _p, _p, _p, _i, _p]),
    "mvf_eval_affine": (_i, [_p, _i64, _p, _i64, _d, _p, C.POINTER(C.c_double), _i, _p, _p, _p, _p, _p, _p, _p, _p, _i,
                             _p]),
    "mvf_integrate": (_i, [_p, _i64, _p, _i64, _d, _p, C.POINTER(C.c_double), _d, _i, _i, _p, _i, _p]),
}

_lib = None


class MVFError(RuntimeError):
    """Raised when a libmvf entry point returns a non-zero status."""


def load():
    """Load libmvf.so (once).  Fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} not found: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            f"g.build()'` (or `make -C spateo-release_amd/csrc`). There is no CPU fallback for this path."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the .so does not export a declared symbol
        fn.restype = res
        fn.argtypes = args
    _lib = lib
    if DEV_KNOBS:
        import sys

        applied = {}
        for env, (opt, parse) in _LEGACY_ENV.items():
            if env in os.environ:
                applied[opt] = int(parse(os.environ[env]))
                check(lib.mvf_debug_option(opt.encode(), applied[opt]), "mvf_debug_option")
        print(f"[spateo_amd] MVF_DEV_KNOBS=1: library {LIB_PATH}, developer options {applied}", file=sys.stderr, flush=True)
    return lib


OPTION_EPOCH = [0]  # bumped by every debug_option call: what callers key their cached launch-plan sizes with


def debug_option(name, value):
    """Set a developer option of the library (mvf.h: mvf_debug_option); returns the previous value."""
    OPTION_EPOCH[0] += 1
    lib = load()
    old = int(lib.mvf_debug_option_get(name.encode()))
    check(lib.mvf_debug_option(name.encode(), int(value)), "mvf_debug_option")
    return old


def debug_options():
    """{name: value} of the developer options that are not at their default (0)."""
    lib = load()
    vals = {n: int(lib.mvf_debug_option_get(n.encode())) for n in DEBUG_OPTIONS}
    return {n: v for n, v in vals.items() if v != 0}


def check(rc, what=""):
    if rc != 0:
        msg = load().mvf_last_error().decode("utf-8", "replace")
        raise MVFError(f"libmvf {what} failed: {msg}")


def device_count() -> int:
    n = C.c_int(0)
    check(load().mvf_device_count(C.byref(n)), "mvf_device_count")
    return n.value
