#!/bin/bash
# end-of-round evidence: whole GPU suite, the bench line, the same command under rocprofv3 --kernel-trace, the other
# BASELINE configurations, the solver probe
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final; mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu -s > $OUT/tests.log 2>&1; echo "tests rc $?"; tail -2 $OUT/tests.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['solve']['avg_ms'], d['f64']['value'], d['f64']['roofline']['achieved'])"
timeout 600 python tools/config_sweep.py --out $OUT/config_sweep.json > $OUT/config_sweep.log 2>&1; tail -3 $OUT/config_sweep.log
for cfg in "3000 60000 6" "2000 40000 6" "1500 30000 6" "1000 30000 6" "500 50000 8"; do
  set -- $cfg
  timeout 400 python tools/minnorm_probe.py $1 $2 $3 0.02 > $OUT/probe$1.json 2> $OUT/probe$1.err
done
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/prof -o p -- python $R/bench.py --no-conk --cpu-cells 0 > $OUT/bench_under_rocprof.json 2> $OUT/prof.log
python $R/tools/rocpd_summary.py $OUT/prof/p_results.db > $OUT/bench_kernel_stats.md 2> $OUT/bench_kernel_stats.err
rm -rf $OUT/prof
head -12 $OUT/bench_kernel_stats.md
