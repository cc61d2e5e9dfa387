#!/bin/bash
# round 5: phase count of the float32 Gram tile stage at 8 M x 3000 (partial-tile budget 9.6 GB = 4 phases / 20 GB = 2 / 40 GB = 1)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
python - <<'PY'
import sys, json
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
    for order in ((0, 20, 40), (40, 20, 0)):
        for gb in order:
            _lib.debug_option("gram_budget_gb", gb)
            kern = HipKernels("cuda:0", "float32")
            eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype="float32", device="cuda:0", kernels=kern)
            eng.init_state(0.9); eng.lstsq_method = "cholesky"
            eng.em_step(lambda_=0.02)
            torch.cuda.synchronize()
            import time
            t0 = time.perf_counter()
            for _ in range(4): eng.em_step(lambda_=0.02)
            torch.cuda.synchronize()
            ms = 1e3 * (time.perf_counter() - t0) / 4
            res.setdefault(str(cells), {}).setdefault(str(gb), []).append(round(ms, 2))
            print(cells, "budget", gb, "GB:", round(ms, 2), "ms/step", flush=True)
            kern.drop_ublk(); del eng, kern; torch.cuda.empty_cache()
_lib.debug_option("gram_budget_gb", 0)
print(json.dumps(res))
json.dump(res, open("gpurun_out/r05_gram_budget_ab.json", "w"))
PY
