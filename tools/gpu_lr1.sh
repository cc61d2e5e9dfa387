#!/bin/bash
# first GPU pass on the rank-revealing minimum-norm solve: its kernel tests, the timing probe at M = 3000 / 2000 / 500
# (both methods), a kernel trace of the lowrank path
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/lr1; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -s -k "lowrank" > $OUT/tests.log 2>&1; echo "tests rc $?"
tail -5 $OUT/tests.log
timeout 400 python tools/minnorm_probe.py 3000 60000 6 0.02 > $OUT/probe3000.json 2> $OUT/probe3000.err; echo "probe3000 rc $?"
timeout 300 python tools/minnorm_probe.py 2000 40000 6 0.02 > $OUT/probe2000.json 2> $OUT/probe2000.err
timeout 300 python tools/minnorm_probe.py 500 50000 8 0.02 > $OUT/probe500.json 2> $OUT/probe500.err
cat $OUT/probe3000.json $OUT/probe2000.json $OUT/probe500.json
cd /tmp; export TMPDIR=/tmp
timeout 400 rocprofv3 --kernel-trace -d $OUT/prof -o p -- python $R/tools/minnorm_probe.py 3000 60000 4 0.02 lowrank > $OUT/prof.log 2>&1
python $R/tools/rocpd_summary.py $OUT/prof/p_results.db > $OUT/prof_kernels.md 2> $OUT/prof_kernels.err
rm -rf $OUT/prof
head -20 $OUT/prof_kernels.md
