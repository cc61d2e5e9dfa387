#!/bin/bash
# round 4, second GPU call: smoke, the whole GPU suite, a short bench (no CPU leg), the multi-rank path with 2 / 4 ranks
# sharing the one GPU (gloo; numbers meaningless, the overlapped-collective code path is the point)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4b; mkdir -p $OUT
cd $R
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?"; tail -3 $OUT/smoke.log
timeout 2400 python -m pytest tests -q -m gpu -s > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -3 $OUT/tests.log; grep -E "FAILED|Error" $OUT/tests.log | head
timeout 900 python bench.py --cpu-cells 0 --no-conk > $OUT/bench_short.json 2> $OUT/bench_short.err; echo "bench rc $?"
python -c "import json;d=json.load(open('$OUT/bench_short.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['solve']['avg_ms'], d['f64']['value'], d['f64']['roofline']['frac'], d['env'], d['developer_options'])"
for N in 2 4; do
  MVF_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N bench.py --gpus $N --steps 2 --warmup 1 --cells 2000000 > $OUT/bench_${N}ranks_one_device.json 2> $OUT/bench_${N}ranks_one_device.err; echo "ranks $N rc $?"
  python -c "import json;d=json.load(open('$OUT/bench_${N}ranks_one_device.json'));print(d['n_gpus'], d['ms_per_step'], d['comm'], [ (r['rank'], r['cells'], round(r['gram_ms'],1), round(r['solve_ms'],1), r['allreduce_ms'], r['rhs_stats_allreduce_ms']) for r in d['per_rank']])"
  grep "bench rank" $OUT/bench_${N}ranks_one_device.err | head -2 | cut -c1-300
done
