#!/bin/bash
# round 5, GPU call A: the RCCL path for the first time, the headline's own solve against scipy, integrate_kernel evidence
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
rm -f gpurun_out/headline_solve_parity.jsonl
timeout 900 python -m pytest tests/test_gpu_rccl.py -x -q -s > gpurun_out/r5a_rccl.log 2>&1; echo "rccl rc=$?" | tee -a gpurun_out/r5a_rccl.log
tail -15 gpurun_out/r5a_rccl.log
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -x -q > gpurun_out/r5a_contract.log 2>&1; echo "contract rc=$?" | tee -a gpurun_out/r5a_contract.log
tail -5 gpurun_out/r5a_contract.log
timeout 1200 python -m pytest tests/test_gpu_scale.py -x -q -s -k "deflated_solve_on_the_benchmark" > gpurun_out/r5a_headline_solve.log 2>&1; echo "solve rc=$?" | tee -a gpurun_out/r5a_headline_solve.log
grep -E "c4_solve|passed|failed|Error" gpurun_out/r5a_headline_solve.log | tail -8
R="$GRAFT_REPO_ROOT"
(cd /tmp && timeout 600 rocprofv3 --kernel-trace -d "$R/gpurun_out/r5a_integrate_prof" -o p -- python "$R/tools/integrate_probe.py" > "$R/gpurun_out/r5a_integrate.log" 2>&1); echo "integrate rc=$?"
tail -6 gpurun_out/r5a_integrate.log
python tools/rocpd_summary.py $(find gpurun_out/r5a_integrate_prof -name "*.db" | head -1) > gpurun_out/r5a_integrate_kernel_stats.md 2> gpurun_out/r5a_integrate_kernel_stats.err
head -8 gpurun_out/r5a_integrate_kernel_stats.md
rm -rf gpurun_out/r5a_integrate_prof
