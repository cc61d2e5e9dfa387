#!/bin/bash
# round 6, GPU call L: evaluator path (CPT = 1 on grid-sized launches, host side), the kernel + EM test files, C2 probe
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_em.py -q -x > gpurun_out/r6l_tests.log 2>&1; echo "tests rc=$?"; tail -3 gpurun_out/r6l_tests.log
python tools/eval_api_profile.py float32 2>&1 | grep -E "wall ms|repeat|to_x4|h2d_padded|synchronize|_digest|matches" | head -12
python tools/eval_api_profile.py float64 2>&1 | grep -E "wall ms" 
timeout 600 python tools/eval_api_probe.py 2000000 > gpurun_out/r06_eval_api.json 2> gpurun_out/r6l_eval.err; tail -1 gpurun_out/r06_eval_api.json | cut -c1-1500
timeout 900 python tools/small_m_probe.py --out gpurun_out/r06_small_m_probe_l.json > gpurun_out/r6l_small_m.log 2>&1
python - <<PY
import json
d=json.load(open('gpurun_out/r06_small_m_probe_l.json'))
for k,v in d.items():
    try: print(k, {m:(round(v[m]['steady_step_ms'],3), round(v[m]['steady_solve_ms'],3)) for m in ('full','deflated')}, v['field_maxrel_between_methods'])
    except Exception as e: pass
PY
