#!/bin/bash
# round 6, GPU call O: counters of the MFMA evaluator kernel (separate --pmc passes, --kernel-trace only)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
OUT=/tmp/evpmc; mkdir -p $OUT $R/gpurun_out/r6o
python $R/tools/eval_kernel_only.py float32
SQ1="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
SQ2="SQ_INSTS_VALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_INSTS_SALU SQ_INSTS_MFMA"
rocprofv3 --pmc $SQ1 --kernel-trace -d $OUT/sq1 -o p -- python $R/tools/eval_kernel_only.py float32 > $R/gpurun_out/r6o/sq1.log 2>&1
rocprofv3 --pmc $SQ2 --kernel-trace -d $OUT/sq2 -o p -- python $R/tools/eval_kernel_only.py float32 > $R/gpurun_out/r6o/sq2.log 2>&1
for n in sq1 sq2; do
  f=$(find $OUT/$n -name "*.db" | head -1)
  [ -n "$f" ] && python $R/tools/rocpd_pmc.py $f eval > $R/gpurun_out/r6o/$n.md 2> $R/gpurun_out/r6o/$n.err
  [ -n "$f" ] && python $R/tools/rocpd_summary.py $f 2>/dev/null | head -8 > $R/gpurun_out/r6o/${n}_kernels.md
done
cat $R/gpurun_out/r6o/sq1.md $R/gpurun_out/r6o/sq2.md; tail -3 $R/gpurun_out/r6o/sq2.log
