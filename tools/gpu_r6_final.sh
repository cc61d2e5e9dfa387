#!/bin/bash
# round 6 evidence on the final tree: the GPU suite (-s: parity lines), smoke, the driver-style bench line, the same command
# under rocprofv3 --kernel-trace (per-kernel and per-(kernel, grid) statistics), SQ counter passes of the Gram kernel
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r6final; mkdir -p $OUT
cd $R; export TMPDIR=/tmp
S=$(date +%s)
timeout 3000 python -m pytest tests -q -m gpu -x -s > $OUT/gpu_tests.txt 2>&1; echo "gpu suite rc=$? in $(( $(date +%s) - S )) s"; grep -E "passed|failed|^FAILED|^E  " $OUT/gpu_tests.txt | tail -6
python tools/parity_table.py $OUT/gpu_tests.txt > $OUT/parity_table.md 2>/dev/null; wc -l $OUT/parity_table.md
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc=$?"; tail -2 $OUT/smoke.log
S=$(date +%s)
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? in $(( $(date +%s) - S )) s"
python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print("headline", d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'])
print("config", json.dumps(d['config']))
print("cpu", d['cpu_baseline']['value'], d['cpu_baseline']['sample_value'], d.get('speedup_vs_cpu_baseline'))
PY
# the multi-rank code path of bench.py on this ONE GPU (2 and 4 ranks on cuda:0, gloo collectives: RCCL refuses two ranks per device)
for N in 2 4; do
  MVF_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2951$N bench.py --gpus $N --steps 2 --warmup 1 --cells 2000000 > $OUT/bench_${N}ranks_one_device.json 2> $OUT/bench_${N}ranks_one_device.err; echo "ranks $N rc $?"
  python -c "import json;d=json.load(open('$OUT/bench_${N}ranks_one_device.json'));print(d['n_gpus'],d['value'],d['ms_per_step'],d['config'].get('collectives_per_step'))" 2>&1 | tail -1
done
cd /tmp
timeout 1200 rocprofv3 --kernel-trace --stats -d $OUT/prof -o p -- python $R/bench.py --no-conk --cpu-cells 0 --no-measure-traffic --no-whole-fit --no-rccl-world1 > $OUT/bench_under_rocprof.json 2> $OUT/prof.log; echo "rocprof bench rc=$?"
DB=$(find $OUT/prof -name "*.db" | head -1)
python $R/tools/rocpd_summary.py $DB > $OUT/bench_kernel_stats.md 2> $OUT/bench_kernel_stats.err
python $R/tools/rocpd_summary.py $DB --by-grid > $OUT/bench_kernel_stats_by_grid.md 2>> $OUT/bench_kernel_stats.err
find $OUT/prof -name "*stats*.csv" | head -3 | while read f; do cp $f $OUT/$(basename $f); done
rm -rf $OUT/prof
head -8 $OUT/bench_kernel_stats.md | cut -c1-220
SQ="SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE"
mkdir -p $OUT/pmc
for dt in float32 float64; do
  B="python $R/bench.py --no-conk --cpu-cells 0 --no-f64 --no-c3 --no-organs32 --no-measure-traffic --no-whole-fit --no-rccl-world1 --lstsq cholesky --steps 1 --warmup 1 --cells 8000000 --dtype $dt"
  timeout 600 rocprofv3 --pmc $SQ --kernel-trace -d $OUT/p_SQ -o p -- $B > $OUT/pmc/${dt}_SQ.json 2> /dev/null
  DB=$(find $OUT/p_SQ -name "*.db" | head -1)
  python $R/tools/rocpd_pmc.py $DB gram_cached > $OUT/pmc/8m_${dt}_m3000_SQ.md 2> $OUT/pmc/8m_${dt}_m3000_SQ.err
  python $R/tools/rocpd_summary.py $DB 2>/dev/null | head -5 > $OUT/pmc/8m_${dt}_m3000_SQ_kernels.md; rm -rf $OUT/p_SQ
done
cat $OUT/pmc/*.md | cut -c1-200 | head -30
