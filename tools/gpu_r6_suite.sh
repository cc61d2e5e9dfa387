#!/bin/bash
# round 6: the whole GPU suite, smoke, and the driver-style bench line
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out
T=${1:-s1}
S=$(date +%s)
timeout 2400 python -m pytest tests -q -m gpu -x -s > gpurun_out/r6${T}_gpu_tests.txt 2>&1; echo "gpu suite rc=$? in $(( $(date +%s) - S )) s"; grep -E "passed|failed|^FAILED|^E  " gpurun_out/r6${T}_gpu_tests.txt | tail -8
timeout 600 python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r6${T}_smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/r6${T}_smoke.log
S=$(date +%s)
timeout 1500 python bench.py ${2:-} > gpurun_out/r6${T}_bench.json 2> gpurun_out/r6${T}_bench.err; echo "bench rc=$? in $(( $(date +%s) - S )) s"
python - <<PY
import json
d=json.load(open('gpurun_out/r6${T}_bench.json'))
print("headline", d['value'], d['ms_per_step'], d['roofline']['frac'])
print("config", json.dumps(d['config']))
print("small", json.dumps(d.get('small_configs'))[:1200])
print("solve", d['solve']['avg_ms'], d['solve'].get('block'))
print("eval", json.dumps(d.get('eval'))[:800])
PY
tail -5 gpurun_out/r6${T}_bench.err
