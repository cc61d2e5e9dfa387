#!/bin/bash
# round 6 (re-entry): per-launch timeline of the FIRST EM iteration of a C2 fit (cold pivoted factorisation: pchol_steps_kernel)
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
cat > /tmp/first_iter.py <<PY
import sys; sys.path[:0] = ["$R", "$R/spateo-release_amd"]
import torch
from spateo_amd._synthetic import make_config
from spateo_amd.vectorfield import SparseVFCEngine, sparsevfc_preprocess
X, V, _ = make_config("C2", N=50_000)
valid, Xv, Yv, idx, ctrl, beta = sparsevfc_preprocess(X, V, M=500, seed=0)
for rep in range(3):
    eng = SparseVFCEngine(Xv, Yv, ctrl, beta, dtype="float32", device="cuda:0")
    eng.init_state(0.9)
    eng.em_step(a=5.0, lambda_=0.02, minP=1e-5, theta=0.75)
    torch.cuda.synchronize()
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/r6u_prof" -o p -- python /tmp/first_iter.py > "$R/gpurun_out/r6u.log" 2>&1); echo "trace rc=$?"
DB=$(find gpurun_out/r6u_prof -name "*.db" | head -1)
python tools/rocpd_timeline.py "$DB" estep_min_kernel 120 > gpurun_out/r6u_c2_first_iteration_timeline.md
rm -rf gpurun_out/r6u_prof
grep -c pchol_steps gpurun_out/r6u_c2_first_iteration_timeline.md; grep "pchol_steps" gpurun_out/r6u_c2_first_iteration_timeline.md | head -4; tail -1 gpurun_out/r6u_c2_first_iteration_timeline.md
