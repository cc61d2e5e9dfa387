#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/lr2; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -s -k "minnorm or rank_deficient" > $OUT/tests.log 2>&1; echo "tests rc $?"
tail -3 $OUT/tests.log
export MVF_LR_TIMING=1
for cfg in "3000 60000 6" "2000 40000 6" "1500 30000 6" "1000 30000 6" "500 50000 8"; do
  set -- $cfg
  timeout 400 python tools/minnorm_probe.py $1 $2 $3 0.02 > $OUT/probe$1.json 2> $OUT/probe$1.err
  cat $OUT/probe$1.json; grep mvf_solve $OUT/probe$1.err | tail -2
done
