#!/bin/bash
# after the phase-aware plan: Gram kernel tests, then the bench's Gram numbers at 8 M, 2 M (C3 shape) and 1 M cells
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/slice4; mkdir -p $OUT
cd $R
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "gram" > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -2 $OUT/tests.log
B="python bench.py --no-conk --cpu-cells 0 --no-f64 --lstsq cholesky --steps 3 --warmup 1"
for cfg in "8000000 3000" "1000000 3000" "2000000 2000"; do
  set -- $cfg
  for dt in float64 float32; do
    timeout 600 $B --dtype $dt --cells $1 --ctrl $2 > $OUT/b_${dt}_$1.json 2> $OUT/b_${dt}_$1.err
    python - <<PY
import json
d = json.load(open("$OUT/b_${dt}_$1.json"))
print("$dt cells $1 x $2: ms/step %.1f gram ms %.1f TF %.2f" % (d["ms_per_step"], d["roofline"]["avg_kernel_ms"], d["roofline"]["achieved"]))
PY
  done
done
