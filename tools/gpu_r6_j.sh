#!/bin/bash
# round 6, GPU call J: wide-Y kernels (test + probe), the headline's own solve under the kernel trace
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_em.py -q -s -x -k "wide_y or kernel_interpolation or dy" > gpurun_out/r6j_wide_tests.log 2>&1; echo "wide tests rc=$?"; grep -E "wide Dy|passed|failed|^E  " gpurun_out/r6j_wide_tests.log | tail
timeout 900 python tools/wide_y_probe.py 2000000 2000 48 --out gpurun_out/r06_wide_y_probe.json > gpurun_out/r6j_wide_probe.log 2>&1; echo "wide probe rc=$?"; tail -2 gpurun_out/r6j_wide_probe.log | cut -c1-900
timeout 900 python tools/lr_phase_probe.py 3000 8000000 8 > gpurun_out/r6j_phase8m.log 2>&1; grep -E "mvf_solve|^\[" gpurun_out/r6j_phase8m.log | tail -5
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d "$R/gpurun_out/r6j_prof" -o p -- python "$R/tools/lr_phase_probe.py" 3000 8000000 7 > "$R/gpurun_out/r6j_trace.log" 2>&1); echo "trace rc=$?"
DB=$(find gpurun_out/r6j_prof -name "*.db" | head -1)
python tools/rocpd_timeline.py "$DB" assemble_kernel 700 > gpurun_out/r6j_headline_solve_timeline.md
tail -2 gpurun_out/r6j_headline_solve_timeline.md
rm -rf gpurun_out/r6j_prof
