#!/bin/bash
# round 3, first GPU call: the GPU suite (minus the cases whose fixtures are still being generated), the bench line with
# its parity object, the config sweep incl. M = 100, a kernel trace of the small configurations, the con_K A/B
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3a; mkdir -p $OUT
cd $R
timeout 1500 python -m pytest tests -q -m gpu -s -k "not ten_step and not c4_generator" > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -3 $OUT/tests.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['solve']['avg_ms'], d['f64']['value'], d['con_k']['GBps'], d['cpu_baseline']['value'], d['speedup_vs_cpu_baseline']); print(json.dumps(d['parity']))"
timeout 300 python tools/conk_ab.py $OUT/conk_ab.json > $OUT/conk_ab.log 2>&1; cat $OUT/conk_ab.log
timeout 600 python tools/config_sweep.py --skip-c3 --out $OUT/config_sweep.json > $OUT/config_sweep.log 2>&1; grep -E "^(C2|default|C5)" $OUT/config_sweep.log | cut -c1-400
cd /tmp; export TMPDIR=/tmp
for cfg in "50000 500 float32" "50000 100 float32"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_$2 -o p -- python $R/tools/small_step_profile.py $1 $2 $3 100 > $OUT/small_$2.log 2>&1
  python $R/tools/rocpd_summary.py $OUT/prof_$2/p_results.db > $OUT/small_$2_kernel_stats.md 2>> $OUT/small_$2.log
  rm -rf $OUT/prof_$2
  tail -1 $OUT/small_$2.log; head -14 $OUT/small_$2_kernel_stats.md | cut -c1-160
done
