#!/bin/bash
# round 6 (re-entry): per-launch timelines on the current tree - one C2 EM iteration (50 k x 500, steady state), one EM iteration at
# 200 k x 2000 (factor form, block 128 / 256), the headline's own solve (8 M x 3000)
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
trace() {  # name, first kernel, count, command...
  local name=$1 first=$2 count=$3; shift 3
  (cd /tmp && timeout 600 rocprofv3 --kernel-trace -d "$R/gpurun_out/r6t_prof" -o p -- "$@" > "$R/gpurun_out/r6t_$name.log" 2>&1); echo "$name trace rc=$?"
  DB=$(find gpurun_out/r6t_prof -name "*.db" | head -1)
  python tools/rocpd_timeline.py "$DB" $first $count > gpurun_out/r6t_${name}_timeline.md
  rm -rf gpurun_out/r6t_prof
  tail -1 gpurun_out/r6t_${name}_timeline.md
}
trace c2_step estep_min_kernel 90 python $R/tools/small_step_profile.py 50000 500 float32 40
trace m2000_step estep_min_kernel 400 python $R/tools/small_step_profile.py 200000 2000 float32 12
trace headline_solve assemble_kernel 700 python $R/tools/lr_phase_probe.py 3000 8000000 7
