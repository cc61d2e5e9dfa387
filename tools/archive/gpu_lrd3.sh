#!/bin/bash
# deflated solve: solver unit tests, then the parity tests at the benchmark's sizes with it as the default
OUT=gpurun_out/lrd; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -x -k "minnorm or deflated or pinv" -s 2>&1 | tail -40 > $OUT/solver_tests.log; tail -25 $OUT/solver_tests.log
timeout 1500 python -m pytest tests/test_gpu_scale.py -m gpu -q -x -s 2>&1 | tail -60 > $OUT/scale_tests.log; tail -30 $OUT/scale_tests.log | cut -c1-260
