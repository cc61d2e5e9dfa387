#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3i; mkdir -p $OUT
cd $R
timeout 300 python tools/conk_ab.py $OUT/conk_ab.json > $OUT/conk_ab.log 2>&1; grep -v amdgpu.ids $OUT/conk_ab.log
MVF_CONK_WIDE=1 timeout 300 python -m pytest tests/test_gpu_kernels.py -q -m gpu -k "con_k" > $OUT/tests.log 2>&1; tail -2 $OUT/tests.log
