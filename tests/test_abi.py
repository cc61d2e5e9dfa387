"""CPU tests of the C-ABI boundary: libmvf.so loads without a GPU, exports every symbol include/mvf.h declares, the
ctypes table covers the header one to one, argument validation reports through the int status + mvf_last_error
channel (no compute is launched), and the product fails loudly - no CPU fallback - when no GPU is present."""
import ctypes
import os
import re

import numpy as np
import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "mvf.h")


def _declared_functions():
    src = open(HEADER).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mvf_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_entry_points():
    names = _declared_functions()
    for must in ["mvf_con_k", "mvf_apply", "mvf_estep_min", "mvf_estep_p", "mvf_gram", "mvf_solve", "mvf_eval",
                 "mvf_last_error"]:
        assert must in names
    # every entry point cites the reference interface it replaces
    text = open(HEADER).read()
    assert text.count("Replaces:") >= 6 and "gaussian_process.py:16-36" in text and "GPVectorField.py:143-190" in text


def test_library_exports_every_declared_symbol():
    from spateo_amd import _lib

    lib = _lib.load()
    declared = _declared_functions()
    for name in declared:
        assert hasattr(lib, name), f"libmvf.so does not export {name}"
    assert sorted(_lib.SIGNATURES) == declared, "ctypes table and include/mvf.h disagree"
    assert lib.mvf_version() == 7


def test_constants_match_header():
    from spateo_amd import _lib

    text = open(HEADER).read()
    assert int(re.search(r"#define MVF_ESTEP_MIN_DOUBLES (\d+)", text).group(1)) == _lib.MVF_ESTEP_MIN_DOUBLES
    for name, val in [("MVF_EVAL_V", _lib.EVAL_V), ("MVF_EVAL_JAC", _lib.EVAL_JAC), ("MVF_EVAL_DIV", _lib.EVAL_DIV),
                      ("MVF_EVAL_CURL", _lib.EVAL_CURL), ("MVF_EVAL_ACC", _lib.EVAL_ACC),
                      ("MVF_EVAL_CURV", _lib.EVAL_CURV), ("MVF_EVAL_TORS", _lib.EVAL_TORS),
                      ("MVF_EVAL_JDET", _lib.EVAL_JDET)]:
        assert int(re.search(rf"{name} = (\d+)", text).group(1)) == val


def test_error_channel_without_launching_anything():
    from spateo_amd import _lib

    lib = _lib.load()
    # bad shapes / null pointers are rejected before any HIP call
    rc = lib.mvf_con_k(None, 4, None, 4, 0, 0.1, None, _lib.MVF_F32, None)
    assert rc != 0 and b"bad shape" in lib.mvf_last_error()
    rc = lib.mvf_con_k(None, 4, None, 4, 3, 0.1, None, _lib.MVF_F32, None)
    assert rc != 0 and b"null pointer" in lib.mvf_last_error()
    info = ctypes.c_int(0)
    rc = lib.mvf_solve(None, None, 0.0, 0.0, None, -1, 3, None, ctypes.byref(info), None, None, 0, None)
    assert rc != 0 and b"mvf_solve" in lib.mvf_last_error()
    rc = lib.mvf_solve_minnorm(None, None, 0.0, 1e-11, 2.2e-16, None, 5, 9, None, ctypes.byref(info), None, 0, 0, None, 0, None,
                               0, None)
    assert rc != 0 and b"mvf_solve_minnorm" in lib.mvf_last_error()
    assert lib.mvf_solve_minnorm_workspace_bytes(3000, 3) >= 2 * 3008 * 3008 * 8
    rc = lib.mvf_solve_minnorm_lr(None, None, 0.0, 0.25, 2.2e-16, None, 5, 9, None, ctypes.byref(info), None, 0, 0, 0, None,
                                  0, None)
    assert rc != 0 and b"mvf_solve_minnorm_lr" in lib.mvf_last_error()
    assert lib.mvf_solve_minnorm_lr_workspace_bytes(3000, 3) >= 2 * 3008 * 3008 * 8
    with pytest.raises(_lib.MVFError, match="mvf_solve_minnorm_lr"):
        _lib.check(rc, "mvf_solve_minnorm_lr")
    # empty problems are fine and launch nothing
    assert lib.mvf_con_k(None, 0, None, 5, 3, 0.1, None, _lib.MVF_F32, None) == 0
    assert lib.mvf_gram_workspace_bytes(0, 10, _lib.MVF_F32) == 0
    assert lib.mvf_gram_workspace_bytes(100_000, 3000, _lib.MVF_F32) > 0
    assert lib.mvf_solve_workspace_bytes(3000, 3) >= 3008 * 3072 * 8


@pytest.mark.skipif(torch.cuda.is_available(), reason="checks the no-GPU behaviour")
def test_gram_workspace_is_bounded_by_a_fraction_of_the_kernel_value_cache():
    """The Gram tile stage runs 8 k-cell slices in phases that REUSE one partial-tile buffer: the workspace stays within
    ~10 % of the kernel-value cache (3 GB floor, 20 GB cap) whatever the cell count, instead of one 128 KB tile per
    (slice, tile pair) - 38 GB at 8 M cells x 3000 control points."""
    from spateo_amd import _lib

    lib = _lib.load()
    tile = 128 * 128 * 8
    for n, m, dt, dsize in ((8_000_000, 3000, _lib.MVF_F32, 4), (8_000_000, 3000, _lib.MVF_F64, 8), (1_000_000, 3000, _lib.MVF_F32, 4),
                            (50_000, 500, _lib.MVF_F32, 4), (300_000_000, 3000, _lib.MVF_F32, 4), (1000, 100, _lib.MVF_F64, 8)):
        nt = -(-m // 128)
        npairs = nt * (nt + 1) // 2
        ws = lib.mvf_gram_workspace_bytes(n, m, dt)
        rhs_part = 1100 * m * 4 * 8  # ~1024 rhs slices of m x 4 float64
        assert ws >= npairs * tile
        assert ws <= max(3e9, min(20e9, 0.1 * n * m * dsize)) + npairs * tile + rhs_part + 4096, (n, m, ws)
    all_tiles = (8_000_000 // 8192 + 1) * 300 * tile
    assert lib.mvf_gram_workspace_bytes(8_000_000, 3000, _lib.MVF_F32) < all_tiles / 3


@pytest.mark.skipif(torch.cuda.is_available(), reason="a statement about machines WITHOUT a GPU")
def test_no_gpu_means_loud_failure_not_a_fallback():
    import spateo_amd as st
    from spateo_amd import _lib

    assert _lib.device_count() == 0
    X = np.random.default_rng(0).standard_normal((50, 3))
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        st.SparseVFC(X, X, None, M=5)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        st.con_K(X, X[:3], 0.1)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        st.vector_field_function(X, {"X_ctrl": X[:3], "C": X[:3], "beta": 0.1})


def test_product_never_imports_the_oracle():
    pkg = os.path.join(ROOT, "spateo-release_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".hip", ".h")):
                src = open(os.path.join(dirpath, f)).read()
                assert not re.search(r"^\s*(from|import)\s+oracle\b", src, flags=re.M), f"{f} imports the oracle"


def test_developer_options_are_an_explicit_call_not_the_environment(monkeypatch):
    """Rounds 1 - 3 had getenv knobs in launch paths; the library no longer reads the environment at all, developer
    options travel through mvf_debug_option (no GPU needed), and the legacy MVF_* names only act behind MVF_DEV_KNOBS=1."""
    from spateo_amd import _lib

    src = "".join(open(os.path.join(ROOT, "spateo-release_amd", "csrc", f)).read()
                  for f in os.listdir(os.path.join(ROOT, "spateo-release_amd", "csrc")) if f.endswith((".hip", ".h")))
    assert "getenv" not in src
    lib = _lib.load()
    assert _lib.debug_options() == {}
    assert _lib.debug_option("slice_len", 4096) == 0 and _lib.debug_options() == {"slice_len": 4096}
    assert _lib.debug_option("slice_len", 0) == 4096 and _lib.debug_options() == {}
    assert lib.mvf_debug_option(b"no_such_option", 1) != 0 and b"unknown option" in lib.mvf_last_error()
    assert lib.mvf_debug_option_get(b"no_such_option") == -1
    assert not _lib.DEV_KNOBS and _lib.LIB_PATH.endswith(os.path.join("spateo_amd", "lib", "libmvf.so"))


def test_communicator_entry_points_report_misuse_without_a_gpu():
    """SURVEY 8(b)'s communicator entry points (ABI 5): libmvf.so links librccl, loads without a GPU, hands out a unique id,
    and reports misuse through the int status + mvf_last_error channel (nothing is launched, no communicator is created)."""
    from spateo_amd import _lib

    lib = _lib.load()
    text = open(HEADER).read()
    assert int(re.search(r"#define MVF_COMM_ID_BYTES (\d+)", text).group(1)) == _lib.MVF_COMM_ID_BYTES == 128
    assert re.search(r"MVF_RED_SUM = 0, MVF_RED_MIN = 1", text) and (_lib.RED_SUM, _lib.RED_MIN) == (0, 1)
    buf = ctypes.create_string_buffer(_lib.MVF_COMM_ID_BYTES)
    assert lib.mvf_comm_unique_id(buf) == 0 and any(buf.raw)
    assert lib.mvf_comm_unique_id(None) != 0 and b"null pointer" in lib.mvf_last_error()
    h = ctypes.c_void_p(None)
    assert lib.mvf_comm_create(ctypes.byref(h), 0, 0, buf) != 0 and b"rank" in lib.mvf_last_error()
    assert lib.mvf_comm_create(ctypes.byref(h), 2, 5, buf) != 0 and h.value is None
    assert lib.mvf_comm_create(None, 1, 0, buf) != 0 and lib.mvf_comm_create(ctypes.byref(h), 1, 0, None) != 0
    assert lib.mvf_allreduce_stats(None, None, 4, _lib.RED_SUM, None) != 0 and b"null communicator" in lib.mvf_last_error()
    assert lib.mvf_comm_info(None, None, None, None) != 0
    assert lib.mvf_comm_destroy(None) == 0
    if not torch.cuda.is_available():  # no device: creation fails in hipGetDevice, loudly, before RCCL is touched
        assert lib.mvf_comm_create(ctypes.byref(h), 1, 0, buf) != 0 and h.value is None
        assert b"hipGetDevice" in lib.mvf_last_error()
    # the shared object really links RCCL (the collective is the library's, not a re-implementation)
    import subprocess

    out = subprocess.run(["readelf", "-d", _lib.LIB_PATH], capture_output=True, text=True).stdout
    assert "librccl" in out and "libamdhip64" in out
