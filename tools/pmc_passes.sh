#!/bin/bash
# PMC passes on the dominant kernel (run on the GPU box through gpurun): separate rocprofv3 runs per counter group,
# --kernel-trace only (never combined with other trace domains).  Outputs: gpurun_out/r2pmc/<config>_<group>/p_results.db
# The coefficient solve runs in the jitter-Cholesky mode here (the counters are about the Gram kernel; the eigensolver's
# thousands of small launches would only slow the counter collection down).
set -u
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r2pmc
mkdir -p $OUT
B="python $R/bench.py --no-conk --cpu-cells 0 --no-f64 --lstsq cholesky --steps 1 --warmup 1"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
ALL=("8m_f32:--cells 8000000 --dtype float32" "8m_f64:--cells 8000000 --dtype float64" "1m_f32:--cells 1000000 --dtype float32" "1m_f64:--cells 1000000 --dtype float64")
# PMC_CONFIGS="8m_f32 8m_f64" restricts the passes to some configurations
for cfg in "${ALL[@]}"; do
  if [ -n "${PMC_CONFIGS:-}" ] && [[ " $PMC_CONFIGS " != *" ${cfg%%:*} "* ]]; then continue; fi
  name=${cfg%%:*}; args=${cfg#*:}
  for ctr in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $ctr --kernel-trace -d $OUT/${name}_$ctr -o p -- $B $args > /dev/null 2>&1
  done
  rocprofv3 --pmc $SQ --kernel-trace -d $OUT/${name}_SQ -o p -- $B $args > /dev/null 2>&1
done
ls $OUT
# summaries only travel back (the databases exceed gpurun's 64 MiB merge limit)
SUM=$R/gpurun_out/r2pmc_summary; mkdir -p $SUM
for d in $OUT/*; do
  n=$(basename $d)
  python $R/tools/rocpd_pmc.py $d/p_results.db > $SUM/$n.md 2> $SUM/$n.err
  python $R/tools/rocpd_summary.py $d/p_results.db 2>/dev/null | head -8 > $SUM/${n}_kernels.md
done
rm -rf $OUT
