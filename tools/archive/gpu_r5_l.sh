#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bench_contract.py tests/test_abi.py -x -q > gpurun_out/r5l_contract.log 2>&1; echo "contract rc=$?"
tail -3 gpurun_out/r5l_contract.log
timeout 900 python bench.py --cells 1000000 --cpu-cells 0 --no-measure-traffic --no-f64 --no-pivot --no-rccl-world1 --steps 2 --warmup 1 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print(json.dumps(d['small_configs']['c5_four_organs'])); print({k:(round(v['ms_per_em_step'],2), round(v['solve_ms'],2)) for k,v in d['small_configs'].items() if isinstance(v,dict) and 'ms_per_em_step' in v})"
