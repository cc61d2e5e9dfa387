#!/bin/bash
# where does the deflated solve overtake the full-width warm-started eigensolve below M = 1024?
mkdir -p gpurun_out/lrd
for cfg in "640 12800" "768 15360" "896 17920" "1000 20000"; do
  set -- $cfg
  timeout 300 python tools/minnorm_probe.py $1 $2 8 0.02 deflated,full > gpurun_out/lrd/cross_$1.json 2> gpurun_out/lrd/cross_$1.err
  python - $1 <<'P'
import json,sys
d=json.load(open(f"gpurun_out/lrd/cross_{sys.argv[1]}.json"))
print(d["M"], "deflated", d["deflated"]["solve_ms"][3:], "r", d["deflated"]["factor_rank"][-1], "kept", d["deflated"]["rank"][-1], "| full", d["full"]["solve_ms"][3:], "| field diff", d.get("field_maxrel_between_methods"))
P
done
