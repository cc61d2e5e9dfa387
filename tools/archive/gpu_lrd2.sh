#!/bin/bash
# kernel-level breakdown of the deflated solve (rocprofv3 kernel trace of the EM-loop probe, summarised)
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/lrd; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof -o lrd -- python $R/tools/minnorm_probe.py ${1:-3000} ${2:-60000} ${3:-12} 0.02 deflated > $OUT/prof_probe.json 2> $OUT/prof_probe.err
echo rc $?
python $R/tools/rocpd_summary.py $OUT/prof/lrd_results.db > $OUT/lrd_kernel_stats.md 2> $OUT/lrd_kernel_stats.err
rm -rf $OUT/prof
head -40 $OUT/lrd_kernel_stats.md | cut -c1-200
