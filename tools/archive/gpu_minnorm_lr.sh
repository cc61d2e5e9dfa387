#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/minnorm_lr; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -s -k "minnorm or rank_deficient" > $OUT/tests.log 2>&1; echo "tests rc $?"
grep -E "greedy|nearby|passed|failed|Error" $OUT/tests.log | tail -8
export MVF_LR_TIMING=1
for cfg in "3000 60000 6" "2000 40000 6" "1500 30000 6"; do
  set -- $cfg
  timeout 400 python tools/minnorm_probe.py $1 $2 $3 0.02 lowrank > $OUT/probe$1.json 2> $OUT/probe$1.err
  cat $OUT/probe$1.json; grep mvf_solve $OUT/probe$1.err | tail -3
done
