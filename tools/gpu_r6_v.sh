#!/bin/bash
# round 6 (re-entry): the bench's small-configuration legs (C2, one C5 organ, four / 32 organs, evaluators) under rocprofv3 --kernel-trace:
# per-(kernel, grid) statistics of the launches the headline's trace does not contain
R=$GRAFT_REPO_ROOT; cd $R; export TMPDIR=/tmp; mkdir -p gpurun_out
(cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/r6v_prof -o p -- python $R/bench.py --no-conk --cpu-cells 0 --no-f64 --no-c3 --no-measure-traffic --no-rccl-world1 --steps 2 --warmup 1 > $R/gpurun_out/r6v_bench.json 2> $R/gpurun_out/r6v.err); echo "rc=$?"
DB=$(find gpurun_out/r6v_prof -name "*.db" | head -1)
python tools/rocpd_summary.py $DB --by-grid > gpurun_out/r6v_small_configs_kernel_stats_by_grid.md 2> gpurun_out/r6v_stats.err
rm -rf gpurun_out/r6v_prof
wc -l gpurun_out/r6v_small_configs_kernel_stats_by_grid.md; grep -E "pchol_steps|jac_eig_kernel<true>|apply_kernel<float, 1>|estep_min_kernel<float>" gpurun_out/r6v_small_configs_kernel_stats_by_grid.md | cut -c1-40,90-200 | head -8
