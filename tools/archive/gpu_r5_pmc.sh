#!/bin/bash
# round 5 SQ counter pass (one counter group per rocprofv3 run, --kernel-trace only) on the Gram kernel of the final tree at the
# headline size, float32 and float64 cells (the FETCH_SIZE / WRITE_SIZE passes run inside bench.py since this round)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5pmc; mkdir -p $OUT/pmc
cd /tmp; export TMPDIR=/tmp
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
for dt in float32 float64; do
  B="python $R/bench.py --no-conk --cpu-cells 0 --no-f64 --no-pivot --no-measure-traffic --no-whole-fit --no-rccl-world1 --lstsq cholesky --steps 1 --warmup 1 --cells 8000000 --dtype $dt"
  timeout 500 rocprofv3 --pmc $SQ --kernel-trace -d $OUT/p_SQ -o p -- $B > $OUT/${dt}_SQ.json 2> /dev/null
  DB=$(find $OUT/p_SQ -name "*.db" | head -1)
  python $R/tools/rocpd_pmc.py $DB gram_cached > $OUT/pmc/8m_${dt}_m3000_SQ.md 2> $OUT/pmc/8m_${dt}_m3000_SQ.err
  python $R/tools/rocpd_summary.py $DB 2>/dev/null | head -5 > $OUT/pmc/8m_${dt}_m3000_SQ_kernels.md; rm -rf $OUT/p_SQ
done
cat $OUT/pmc/*.md | cut -c1-220
