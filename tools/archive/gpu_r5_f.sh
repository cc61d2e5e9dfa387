#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -q -s -k "minnorm or deflated" > gpurun_out/r5f_kernels.log 2>&1; echo "kernels rc=$?"
grep -E "^m=|direct form|passed|failed|Error|assert" gpurun_out/r5f_kernels.log | tail -30
