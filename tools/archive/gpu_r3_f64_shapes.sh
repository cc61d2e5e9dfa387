#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3j; mkdir -p $OUT
cd $R
for v in base shapes1 shapes2 base; do
  if [ $v = base ]; then unset MVF_LIB_PATH; else export MVF_LIB_PATH=$R/tools/ab/libmvf_$v.so; fi
  timeout 400 python bench.py --dtype float64 --no-f64 --no-conk --cpu-cells 0 --steps 2 --warmup 1 > $OUT/b_$v.json 2> $OUT/b_$v.err
  python -c "import json;d=json.load(open('$OUT/b_$v.json'));print('$v f64 gram ms', round(d['roofline']['avg_kernel_ms'],1), 'TF', round(d['roofline']['achieved'],2), 'frac', round(d['roofline']['frac'],4))"
done
