#!/bin/bash
# which counters exist for address translation, and their values on the float64 cached Gram kernel at 8 M vs 1 M cells
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/tlb; mkdir -p $OUT
rocprofv3 -L 2>/dev/null | grep -i -E "utcl|tlb|translat" | head -40 > $OUT/counters.txt
cat $OUT/counters.txt | head -30
B="python $R/bench.py --no-conk --cpu-cells 0 --no-f64 --lstsq cholesky --steps 1 --warmup 1 --dtype float64"
CTR="TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_REQUEST_sum"
for cells in 8000000 1000000; do
  timeout 600 rocprofv3 --pmc $CTR --kernel-trace -d $OUT/c$cells -o p -- $B --cells $cells > $OUT/c$cells.log 2>&1
  python $R/tools/rocpd_pmc.py $OUT/c$cells/p_results.db gram_cached > $OUT/c$cells.md 2> $OUT/c$cells.err
  cat $OUT/c$cells.md
  rm -rf $OUT/c$cells
done
