#!/bin/bash
# round 3, end-of-round evidence: smoke(), the whole GPU suite, the bench line, the multi-rank bench path with 2 and 4
# ranks sharing the one GPU (gloo collectives; numbers meaningless, the code path is the point)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3e; mkdir -p $OUT
cd $R
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > $OUT/smoke.log 2>&1; echo "smoke rc $?"; tail -7 $OUT/smoke.log
timeout 1800 python -m pytest tests -q -m gpu -s > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -2 $OUT/tests.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['solve']['avg_ms'], d['f64']['value'], d['f64']['roofline']['frac'], d['con_k']['GBps'], d['speedup_vs_cpu_baseline']); print(json.dumps(d['parity']))"
for N in 2 4; do
  MVF_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $N --master-addr 127.0.0.1 --master-port 2950$N bench.py --gpus $N --steps 2 --warmup 1 --cells 2000000 > $OUT/bench_${N}ranks_one_device.json 2> $OUT/bench_${N}ranks_one_device.err; echo "ranks $N rc $?"
  python -c "import json;d=json.load(open('$OUT/bench_${N}ranks_one_device.json'));print(d['n_gpus'], d['ms_per_step'], d['comm'], [ (r['rank'], r['cells'], round(r['gram_ms'],1), round(r['solve_ms'],1)) for r in d['per_rank']])"
  grep "bench rank" $OUT/bench_${N}ranks_one_device.err | head -4 | cut -c1-200
done
