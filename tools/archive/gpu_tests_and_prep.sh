#!/bin/bash
# the whole GPU suite + the host/device preprocessing time at BASELINE config 4
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final; mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu > $OUT/tests_last.log 2>&1; echo "tests rc $?"; tail -2 $OUT/tests_last.log
python - <<'PY'
import sys, time
sys.path.insert(0, "spateo-release_amd")
import numpy as np, torch
from spateo_amd._synthetic import make_config
from spateo_amd import vectorfield as vfm
X, V, M = make_config("C4")
vfm.sparsevfc_preprocess(X[:300000], V[:300000], M=1100, seed=0, device="cuda:0")  # warm the library / allocator
torch.cuda.synchronize()
for tag, M_ in (("C4", M),):
    t = time.perf_counter(); out = vfm.sparsevfc_preprocess(X, V, M=M_, seed=0, device="cuda:0"); torch.cuda.synchronize()
    print(f"{tag} preprocessing (8 M cells, M = {M_}): {time.perf_counter() - t:.3f} s, beta {out[5]:.6g}")
    t = time.perf_counter(); h = vfm.bandwidth_selector(out[4], "cuda:0"); print(f"  of which bandwidth on the device: {time.perf_counter() - t:.4f} s")
    old = vfm._DEVICE_KNN_MIN_POINTS; vfm._DEVICE_KNN_MIN_POINTS = 10**9
    t = time.perf_counter(); h2 = vfm.bandwidth_selector(out[4]); print(f"  kd-tree on the host: {time.perf_counter() - t:.4f} s, rel diff {abs(h - h2) / h2:.1e}")
    vfm._DEVICE_KNN_MIN_POINTS = old
PY
