#!/bin/bash
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
for extra in "" "--no-conk" "--no-rccl-world1" "--no-rccl-world1 --no-conk --no-pivot --no-f64"; do
  python bench.py --cells 1000000 --cpu-cells 0 --no-measure-traffic --steps 2 --warmup 1 $extra 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read())
s=d['small_configs']
print('$extra', {k:(round(v['ms_per_em_step'],2), round(v['solve_ms'],2)) for k,v in s.items() if isinstance(v,dict)})
"
done
