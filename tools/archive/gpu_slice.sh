#!/bin/bash
# Gram tile stage probes: MVF_SLICE_LEN overrides the plan's slice cap ("dtype slice_len" pairs in SLICE_PROBES)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/gram_ab; mkdir -p $OUT
cd $R
B="python bench.py --no-conk --cpu-cells 0 --no-f64 --lstsq cholesky --steps 3 --warmup 1 --cells ${CELLS:-8000000}"
for cfg in ${SLICE_PROBES:-float64:0 float32:0}; do
  dt=${cfg%%:*}; sl=${cfg#*:}
  if [ "$sl" = "0" ]; then unset MVF_SLICE_LEN; else export MVF_SLICE_LEN=$sl; fi
  timeout 600 $B --dtype $dt > $OUT/b_${dt}_$sl.json 2> $OUT/b_${dt}_$sl.err
  python - <<PY
import json
d = json.load(open("$OUT/b_${dt}_$sl.json"))
print("$dt slice_len $sl: ms/step %.1f gram ms %.1f TF %.2f" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["achieved"]))
PY
done
