#!/bin/bash
# round 5, GPU call C: per-launch timeline of the M = 500 deflated solve; host profile of a whole 8 M x 3000 call
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/r5c_prof" -o p -- python "$R/tools/small_m_trace.py" deflated > "$R/gpurun_out/r5c_trace.log" 2>&1); echo "trace rc=$?"
DB=$(find gpurun_out/r5c_prof -name "*.db" | head -1)
python tools/rocpd_timeline.py "$DB" assemble_kernel 400 > gpurun_out/r5c_small_m_timeline.md
tail -3 gpurun_out/r5c_small_m_timeline.md
rm -rf gpurun_out/r5c_prof
timeout 600 python tools/fit_profile.py > gpurun_out/r5c_fit_profile.log 2>&1; echo "fit_profile rc=$?"
head -60 gpurun_out/r5c_fit_profile.log
