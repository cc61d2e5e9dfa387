#!/bin/bash
# same-box A/B of the cached Gram kernel: previous build (tools/ab/libmvf_prev.so) vs this build; float32 + float64 at 8 M
# cells and at the per-rank size 1 M; then the Gram-related GPU tests on this build
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3d; mkdir -p $OUT
cd $R
for cells in 8000000 1000000; do
  for lib in prev new; do
    if [ $lib = prev ]; then export MVF_LIB_PATH=$R/tools/ab/libmvf_prev.so; else unset MVF_LIB_PATH; fi
    timeout 600 python bench.py --no-conk --cpu-cells 0 --steps 3 --warmup 1 --cells $cells > $OUT/bench_${cells}_$lib.json 2> $OUT/bench_${cells}_$lib.err
    python -c "import json;d=json.load(open('$OUT/bench_${cells}_$lib.json'));print('$cells $lib f32 gram ms', round(d['roofline']['avg_kernel_ms'],2), 'TF', round(d['roofline']['achieved'],2), 'step', round(d['ms_per_step'],1), '| f64 gram ms', round(d['f64']['roofline']['avg_kernel_ms'],2), 'TF', round(d['f64']['roofline']['achieved'],2), 'frac', round(d['f64']['roofline']['frac'],4), 'sigma2', d['sigma2_after'] if 'sigma2_after' in d else d['config']['sigma2_after'], d['f64']['sigma2_after'])"
  done
done
unset MVF_LIB_PATH
timeout 900 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_em.py -q -m gpu -x > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -2 $OUT/tests.log
