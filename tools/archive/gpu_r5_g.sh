#!/bin/bash
# round 5, GPU call G: the full bench line with every new object (driver-visible), then the contract test
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
S=$(date +%s)
timeout 1500 python bench.py > gpurun_out/r5g_bench.json 2> gpurun_out/r5g_bench.err; echo "bench rc=$? in $(( $(date +%s) - S )) s"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r5g_bench.json'))
print("headline", d['value'], d['ms_per_step'], d['roofline']['frac'], "f64", d['f64']['ms_per_step'], d['f64']['roofline']['frac'])
r=d['roofline']; print("traffic", r['traffic'], r['traffic_from_committed_profile'], r.get('traffic_over_algorithmic_bytes'), r.get('traffic_detail'))
print("small", json.dumps(d.get('small_configs')))
print("whole", json.dumps(d.get('whole_fit')))
print("parity", json.dumps(d.get('parity'))[:1500])
print("rccl", {k:(v.get('ms_per_step') if isinstance(v,dict) else v) for k,v in d['rccl_world1'].items()})
PY
tail -5 gpurun_out/r5g_bench.err
timeout 900 python -m pytest tests/test_gpu_bench_contract.py -x -q > gpurun_out/r5g_contract.log 2>&1; echo "contract rc=$?"
tail -5 gpurun_out/r5g_contract.log
