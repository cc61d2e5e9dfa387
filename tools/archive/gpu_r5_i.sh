#!/bin/bash
# round 5, GPU call I: the LDS-shared-panel float64 Gram kernel: bit identity, then same-box A/B at 8 M and 1 M cells
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_gpu_kernels.py -x -q -k "gram_cached_u_is_bit_identical or partial_tiles" > gpurun_out/r5i_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r5i_tests.log
python - <<'PY'
import os, sys, json, time
sys.path.insert(0, "."); sys.path.insert(0, "spateo-release_amd")
import numpy as np, torch
from spateo_amd import _lib
from spateo_amd._kernels import HipKernels
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess
res = {}
for cells in (8_000_000, 1_000_000):
    X, V, M = make_config("C4", N=cells)
    valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=M, seed=0)
    kern = HipKernels("cuda:0", "float64")
    eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype="float64", device="cuda:0", kernels=kern)
    eng.init_state(0.9)
    eng.lstsq_method = "cholesky"
    for order in ((0, 1), (1, 0), (0, 1)):
        for opt in order:
            _lib.debug_option("gram_f64_lds", opt)
            eng.em_step(lambda_=0.02)
            kern.gram_events = []
            for _ in range(3):
                eng.em_step(lambda_=0.02)
            torch.cuda.synchronize()
            ms = [a.elapsed_time(b) for a, b in kern.gram_events]
            kern.gram_events = None
            tf = cells * 3000.0 * 3001.0 / (np.mean(ms) * 1e-3) / 1e12
            res.setdefault(f"{cells}", {}).setdefault("lds" if opt else "regs", []).append((round(float(np.mean(ms)), 2), round(tf, 2)))
            print(cells, "lds" if opt else "regs", np.round(ms, 2), f"{tf:.2f} TF", flush=True)
    _lib.debug_option("gram_f64_lds", 0)
    kern.drop_ublk(); del eng, kern; torch.cuda.empty_cache()
json.dump(res, open("gpurun_out/r05_gram_f64_lds_ab.json", "w"), indent=1)
print(json.dumps(res))
PY
