#!/bin/bash
# PMC passes of round 3, part 2: the float64 Gram kernel at 8 M cells (FETCH / WRITE / SQ) and con_K (WRITE / FETCH)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3f; mkdir -p $OUT/pmc
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-conk --cpu-cells 0 --no-f64 --lstsq cholesky --steps 1 --warmup 1 --cells 8000000 --dtype float64"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 500 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/p_$ctr -o p -- $B > /dev/null 2>&1
  python $R/tools/rocpd_pmc.py $OUT/p_$ctr/p_results.db gram_cached > $OUT/pmc/8m_f64_$ctr.md 2> $OUT/pmc/8m_f64_$ctr.err; rm -rf $OUT/p_$ctr
done
timeout 500 rocprofv3 --pmc $SQ --kernel-trace -d $OUT/p_SQ -o p -- $B > /dev/null 2>&1
python $R/tools/rocpd_pmc.py $OUT/p_SQ/p_results.db gram_cached > $OUT/pmc/8m_f64_SQ.md 2> $OUT/pmc/8m_f64_SQ.err
python $R/tools/rocpd_summary.py $OUT/p_SQ/p_results.db 2>/dev/null | head -5 > $OUT/pmc/8m_f64_SQ_kernels.md; rm -rf $OUT/p_SQ
for ctr in WRITE_SIZE FETCH_SIZE; do
  timeout 300 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/c_$ctr -o p -- python $R/tools/conk_once.py > $OUT/conk_$ctr.log 2>&1
  python $R/tools/rocpd_pmc.py $OUT/c_$ctr/p_results.db conk > $OUT/pmc/conk_c3_f32_$ctr.md 2> $OUT/pmc/conk_$ctr.err
  python $R/tools/rocpd_summary.py $OUT/c_$ctr/p_results.db 2>/dev/null | head -4 > $OUT/pmc/conk_c3_f32_${ctr}_kernels.md; rm -rf $OUT/c_$ctr
done
cat $OUT/pmc/*.md | cut -c1-220
