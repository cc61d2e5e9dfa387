#!/bin/bash
# round 5, GPU call B: bench contract (stdout fix), the small-M deflated solve vs full width, the solver-noise experiment,
# first full bench line with the rccl_world1 object
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -x -q > gpurun_out/r5b_contract.log 2>&1; echo "contract rc=$?"
tail -3 gpurun_out/r5b_contract.log
timeout 900 python tools/small_m_probe.py --out gpurun_out/r05_small_m_probe.json > gpurun_out/r5b_small_m.log 2>&1; echo "small_m rc=$?"
grep -E "mvf_solve_minnorm_lr" gpurun_out/r5b_small_m.log | head -12
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_small_m_probe.json'))
for k,v in d.items():
    print(k, {m:(v[m]['steady_step_ms'], v[m]['steady_solve_ms'], v[m]['rank'][-1], v[m]['block'][-1]) for m in ('full','deflated')}, v['field_maxrel_between_methods'])
PY
timeout 900 python tools/solver_noise_probe.py --out gpurun_out/r05_solver_noise.json > gpurun_out/r5b_noise.log 2>&1; echo "noise rc=$?"
grep "^M=" gpurun_out/r5b_noise.log
timeout 900 python bench.py > gpurun_out/r5b_bench.json 2> gpurun_out/r5b_bench.err; echo "bench rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5b_bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['f64']['ms_per_step'], d['f64']['roofline']['frac'])
print(json.dumps(d.get('rccl_world1'))[:3000])
PY
