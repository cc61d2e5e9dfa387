#!/bin/bash
# round 3, third GPU call: con_K row-span form A/B (rows per workgroup 8 / 16 / 32 / 64), the whole GPU suite again,
# PMC passes (FETCH_SIZE, WRITE_SIZE, SQ) on the headline float32 Gram kernel for this round's profile
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3c; mkdir -p $OUT
cd $R
timeout 300 python tools/conk_ab.py $OUT/conk_ab.json > $OUT/conk_ab.log 2>&1; grep -v amdgpu.ids $OUT/conk_ab.log
timeout 1800 python -m pytest tests -q -m gpu -s > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -3 $OUT/tests.log
cd /tmp; export TMPDIR=/tmp
B="python $R/bench.py --no-conk --cpu-cells 0 --no-f64 --lstsq cholesky --steps 1 --warmup 1 --cells 8000000 --dtype float32"
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
for ctr in FETCH_SIZE WRITE_SIZE; do
  timeout 400 rocprofv3 --pmc $ctr --kernel-trace -d $OUT/pmc_$ctr -o p -- $B > /dev/null 2>&1
done
timeout 400 rocprofv3 --pmc $SQ --kernel-trace -d $OUT/pmc_SQ -o p -- $B > /dev/null 2>&1
mkdir -p $OUT/pmc
for n in FETCH_SIZE WRITE_SIZE SQ; do
  python $R/tools/rocpd_pmc.py $OUT/pmc_$n/p_results.db gram_cached > $OUT/pmc/8m_f32_$n.md 2> $OUT/pmc/8m_f32_$n.err
  python $R/tools/rocpd_summary.py $OUT/pmc_$n/p_results.db 2>/dev/null | head -6 > $OUT/pmc/8m_f32_${n}_kernels.md
  rm -rf $OUT/pmc_$n
done
cat $OUT/pmc/8m_f32_FETCH_SIZE.md $OUT/pmc/8m_f32_WRITE_SIZE.md | cut -c1-200; head -12 $OUT/pmc/8m_f32_SQ.md | cut -c1-200
