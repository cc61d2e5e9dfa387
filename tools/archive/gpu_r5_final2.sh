#!/bin/bash
# round 5, end of round: the whole GPU suite, smoke(), the driver-style bench line - all on the final tree
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5final2; mkdir -p $OUT
cd $R; export TMPDIR=/tmp
rm -f gpurun_out/headline_solve_parity.jsonl
S=$(date +%s)
timeout 2400 python -m pytest tests -q -m gpu -s > $OUT/tests.log 2>&1; echo "tests rc=$? in $(( $(date +%s) - S )) s"
grep -E "passed|failed" $OUT/tests.log | tail -2
grep -E "^FAILED|^ERROR" $OUT/tests.log | head
python tools/parity_table.py $OUT/tests.log > $OUT/parity_table.md
cp gpurun_out/headline_solve_parity.jsonl $OUT/ 2>/dev/null
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v amdgpu.ids > $OUT/smoke.log; tail -2 $OUT/smoke.log
S=$(date +%s)
timeout 1200 python bench.py --steps 20 --warmup 5 > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc=$? in $(( $(date +%s) - S )) s"
python - <<PY
import json
d=json.load(open('$OUT/bench.json'))
print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['solve']['avg_ms'], d['f64']['value'], d['f64']['roofline']['frac'])
print({k:(round(v['ms_per_em_step'],2), round(v['solve_ms'],2)) for k,v in d['small_configs'].items() if isinstance(v,dict) and 'ms_per_em_step' in v})
PY
