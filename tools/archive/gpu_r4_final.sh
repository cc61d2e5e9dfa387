#!/bin/bash
# round 4, end-of-round evidence on the final tree: smoke(), the whole GPU suite (pytest -s log -> parity table), the default
# bench line, bench --gram-mode pivot, and the multi-rank bench path with 2 and 4 ranks sharing the one GPU (gloo collectives;
# numbers meaningless, the overlapped-collective code path is the point)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4z; mkdir -p $OUT
cd $R
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?"; tail -3 $OUT/smoke.log
timeout 2700 python -m pytest tests -q -m gpu -s > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -2 $OUT/tests.log; grep -E "FAILED|ERROR" $OUT/tests.log | head
python tools/parity_table.py $OUT/tests.log | grep -v -- "— | — | — | — | — | — |" > $OUT/parity_table.md
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['solve']['avg_ms'], d['f64']['value'], d['f64']['roofline']['frac'], d['con_k']['GBps'], d['speedup_vs_cpu_baseline'], d['pivot_subset']['value'], d['eval']); print(json.dumps(d['parity']))"
timeout 600 python bench.py --gram-mode pivot --cpu-cells 0 --no-conk --no-f64 > $OUT/bench_pivot.json 2> $OUT/bench_pivot.err; echo "bench pivot rc $?"
for N in 2 4; do
  MVF_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N bench.py --gpus $N --steps 2 --warmup 1 --cells 2000000 > $OUT/bench_${N}ranks_one_device.out 2> $OUT/bench_${N}ranks_one_device.err; echo "ranks $N rc $?"
  tail -1 $OUT/bench_${N}ranks_one_device.out > $OUT/bench_${N}ranks_one_device.json
  python -c "import json;d=json.load(open('$OUT/bench_${N}ranks_one_device.json'));print(d['n_gpus'], d['ms_per_step'], d['comm']['collectives_per_step'])"
done
