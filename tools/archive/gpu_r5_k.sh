#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_em.py tests/test_gpu_rccl.py tests/test_gpu_scale.py -q -s -k "not pivot and not streamed and not benchmark and not float32_mode_vs" > gpurun_out/r5k_tests.log 2>&1; echo "tests rc=$?"
tail -3 gpurun_out/r5k_tests.log
grep -E "^\.?(C2 lambda|C5 organ)" gpurun_out/r5k_tests.log | cut -c1-400
