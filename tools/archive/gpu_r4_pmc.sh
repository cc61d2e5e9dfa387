#!/bin/bash
# round 4 PMC passes (one counter group per rocprofv3 run, --kernel-trace only): the float32 Gram kernel at the headline
# size (8 M x 3000) and at the size the pivot mode leaves it with (8 M x 896).  Summaries only travel back.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4pmc; mkdir -p $OUT/pmc
cd /tmp; export TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
for cfg in "m3000:--ctrl 3000" "m896:--ctrl 896"; do
  name=${cfg%%:*}; args=${cfg#*:}
  B="python $R/bench.py --no-conk --cpu-cells 0 --no-f64 --no-pivot --lstsq cholesky --steps 1 --warmup 1 --cells 8000000 --dtype float32 $args"
  for ctr in FETCH_SIZE WRITE_SIZE; do
    timeout 500 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/p_$ctr -o p -- $B > $OUT/${name}_$ctr.json 2> /dev/null
    python $R/tools/rocpd_pmc.py $OUT/p_$ctr/p_results.db gram_cached > $OUT/pmc/8m_f32_${name}_$ctr.md 2> $OUT/pmc/8m_f32_${name}_$ctr.err; rm -rf $OUT/p_$ctr
  done
  timeout 500 rocprofv3 --pmc $SQ --kernel-trace -d $OUT/p_SQ -o p -- $B > $OUT/${name}_SQ.json 2> /dev/null
  python $R/tools/rocpd_pmc.py $OUT/p_SQ/p_results.db gram_cached > $OUT/pmc/8m_f32_${name}_SQ.md 2> $OUT/pmc/8m_f32_${name}_SQ.err
  python $R/tools/rocpd_summary.py $OUT/p_SQ/p_results.db 2>/dev/null | head -5 > $OUT/pmc/8m_f32_${name}_SQ_kernels.md; rm -rf $OUT/p_SQ
done
cat $OUT/pmc/*.md | cut -c1-200
