#!/bin/bash
# round 3, second GPU call: the whole GPU suite with the complete fixtures, con_K 3-way A/B, the config sweep (incl. the
# single-launch solve at M = 100), the Ozaki inner-loop probe, kernel traces of the small configurations and of bench.py
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3b; mkdir -p $OUT
cd $R
timeout 1800 python -m pytest tests -q -m gpu -s > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -3 $OUT/tests.log
timeout 300 python tools/conk_ab.py $OUT/conk_ab.json > $OUT/conk_ab.log 2>&1; grep -v amdgpu.ids $OUT/conk_ab.log
timeout 60 tools/ozaki_probe > $OUT/ozaki_probe.jsonl 2>&1; cat $OUT/ozaki_probe.jsonl
timeout 900 python tools/config_sweep.py --out $OUT/config_sweep.json > $OUT/config_sweep.log 2>&1; grep -E "^(C2|default|C3|C5)" $OUT/config_sweep.log | cut -c1-330
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['solve']['avg_ms'], d['f64']['value'], d['f64']['roofline']['frac'], d['con_k']['GBps'], d['speedup_vs_cpu_baseline']); print(json.dumps(d['parity']))"
cd /tmp; export TMPDIR=/tmp
for cfg in "50000 500 float32" "50000 100 float32"; do
  set -- $cfg
  timeout 300 rocprofv3 --kernel-trace -d $OUT/prof_$2 -o p -- python $R/tools/small_step_profile.py $1 $2 $3 100 > $OUT/small_$2.log 2>&1
  python $R/tools/rocpd_summary.py $OUT/prof_$2/p_results.db > $OUT/small_$2_kernel_stats.md 2>> $OUT/small_$2.log
  rm -rf $OUT/prof_$2
  grep "ms/step" $OUT/small_$2.log; head -8 $OUT/small_$2_kernel_stats.md | cut -c1-150
done
timeout 900 rocprofv3 --kernel-trace -d $OUT/prof -o p -- python $R/bench.py --no-conk --cpu-cells 0 > $OUT/bench_under_rocprof.json 2> $OUT/prof.log
python $R/tools/rocpd_summary.py $OUT/prof/p_results.db > $OUT/bench_kernel_stats.md 2> $OUT/bench_kernel_stats.err
rm -rf $OUT/prof
head -10 $OUT/bench_kernel_stats.md | cut -c1-170
