#!/bin/bash
# round 6, GPU call N: the evaluator as an MFMA product (eval_mfma_kernel): kernel + EM + scale test files, API probe
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_em.py -q -x > gpurun_out/r6n_tests.log 2>&1; echo "tests rc=$?"; tail -15 gpurun_out/r6n_tests.log
timeout 900 python -m pytest tests/test_gpu_scale.py -q -x -k "jacobian_and_curl" >> gpurun_out/r6n_tests.log 2>&1; echo "scale eval rc=$?"; tail -3 gpurun_out/r6n_tests.log
timeout 600 python tools/eval_api_probe.py 2000000 > gpurun_out/r06_eval_api_mfma.json 2> gpurun_out/r6n_eval.err; tail -1 gpurun_out/r06_eval_api_mfma.json | cut -c1-1800
