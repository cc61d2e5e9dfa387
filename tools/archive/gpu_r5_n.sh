#!/bin/bash
# round 5, last tree: the multi-rank bench path once more with 2 ranks on the one GPU (gloo; numbers meaningless)
set -u
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp; mkdir -p gpurun_out
n=2
MVF_BENCH_ONE_DEVICE=1 timeout 600 python -m torch.distributed.run --nnodes=1 --nproc-per-node $n --master-addr 127.0.0.1 --master-port 2950$n bench.py --gpus $n --steps 3 --warmup 1 --cells 2000000 > gpurun_out/r5n_bench_${n}ranks.json 2> gpurun_out/r5n_bench_${n}ranks.err; echo "bench $n ranks rc=$?"
python - <<PY
import json
lines=[l for l in open('gpurun_out/r5n_bench_${n}ranks.json') if l.strip()]
print(len(lines), "stdout line(s)")
d=json.loads(lines[-1]); print(d['n_gpus'], d['value'], d['ms_per_step'], d['comm']['collectives_per_step'], d['comm']['backend'], [(r['rank'], round(r['gram_ms'],1), round(r['allreduce_ms'],2)) for r in d['per_rank']])
PY
tail -3 gpurun_out/r5n_bench_${n}ranks.err
