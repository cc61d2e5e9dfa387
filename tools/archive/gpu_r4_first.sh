#!/bin/bash
# round 4, first GPU call: kernel tests (compact rank-revealing solve, fused evaluator pass, pinned transfers), the solve
# inside the EM loop in both forms, the evaluator path at the API + its kernel row under rocprofv3
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4a; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -s -x > $OUT/kernels.log 2>&1; echo "kernel tests rc $?"; tail -3 $OUT/kernels.log
grep -E "^m=|compact form" $OUT/kernels.log | cut -c1-400
for M in 3000 2000 1100; do
  MVF_LR_TIMING=1 timeout 600 python tools/minnorm_probe.py $M $((20*M)) 8 0.02 lowrank,lowrank_rows > $OUT/minnorm_$M.json 2> $OUT/minnorm_$M.err; echo "probe $M rc $?"
  cat $OUT/minnorm_$M.json | cut -c1-1500
  grep "mvf_solve_minnorm_lr" $OUT/minnorm_$M.err | tail -4
done
timeout 600 python tools/eval_api_probe.py > $OUT/eval_api.json 2> $OUT/eval_api.err; echo "eval probe rc $?"; cat $OUT/eval_api.json; tail -3 $OUT/eval_api.err
cd /tmp; export TMPDIR=/tmp
timeout 600 rocprofv3 --kernel-trace -d $OUT/prof -o p -- python $R/tools/eval_api_probe.py 200000 > $OUT/eval_api_under_rocprof.json 2> $OUT/prof.log
python $R/tools/rocpd_summary.py $OUT/prof/p_results.db > $OUT/eval_kernel_stats.md 2> $OUT/eval_kernel_stats.err
rm -rf $OUT/prof
grep -i "eval_kernel" $OUT/eval_kernel_stats.md | cut -c1-250
