#!/bin/bash
# per-rank workloads of the 2 / 4 / 8-GPU strong-scaling runs, measured on ONE GPU: bench.py at 4 M, 2 M, 1 M cells x 3000
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3l; mkdir -p $OUT
cd $R
for cells in 4000000 2000000 1000000; do
  timeout 400 python bench.py --no-f64 --no-conk --cpu-cells 0 --steps 3 --warmup 1 --cells $cells > $OUT/b_$cells.json 2> $OUT/b_$cells.err
  python -c "import json;d=json.load(open('$OUT/b_$cells.json'));print($cells, 'step ms', round(d['ms_per_step'],1), 'gram ms', round(d['roofline']['avg_kernel_ms'],1), 'TF', round(d['roofline']['achieved'],2), 'solve ms', round(d['solve']['avg_ms'],1))"
done
