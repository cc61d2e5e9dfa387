#!/bin/bash
# deflated solve: the tests that failed in the first full run, one rank's share of the 2/4/8-GPU runs, kernel stats
R=$GRAFT_REPO_ROOT; OUT=$R/gpurun_out/lrd; mkdir -p $OUT; cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_scale.py -m gpu -q -k "exactly_rank_deficient or deflated or (pivot_mode and m3000)" 2>&1 | tail -5
timeout 1500 python tools/rank_sizes.py --out $OUT/rank_sizes.json 2> $OUT/rank_sizes.err | cut -c1-600
bash tools/gpu_lrd2.sh 3000 60000 > /dev/null 2>&1; head -30 $OUT/lrd_kernel_stats.md | cut -c1-160
