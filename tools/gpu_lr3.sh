#!/bin/bash
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/lr3; mkdir -p $OUT
cd $R
export MVF_LR_TIMING=1
for w in 512 256 128; do
  MVF_JAC_GRAM_WGS=$w timeout 400 python tools/minnorm_probe.py 3000 60000 5 0.02 lowrank > $OUT/p$w.json 2> $OUT/p$w.err
  echo "gram wgs $w"; python -c "import json;d=json.load(open('$OUT/p$w.json'));print(d['lowrank']['solve_ms'])"; grep -A1 mvf_solve $OUT/p$w.err | tail -2
done
