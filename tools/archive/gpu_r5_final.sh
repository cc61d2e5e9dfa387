#!/bin/bash
# round 5 evidence: the two repaired tests, the bench line (driver-style), the same command under rocprofv3 --kernel-trace,
# the other BASELINE configurations
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r5final; mkdir -p $OUT
cd $R
timeout 600 python -m pytest tests/test_gpu_em.py -q -s -k "default_lambda_within_reference_noise_floor or em_matches" > $OUT/em_tests.log 2>&1; echo "em tests rc $?"; tail -2 $OUT/em_tests.log
S=$(date +%s)
timeout 1200 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $? in $(( $(date +%s) - S )) s"
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['solve']['avg_ms'], d['f64']['value'], d['f64']['roofline']['frac']);print(json.dumps(d['small_configs']));print(json.dumps(d['whole_fit']['c2'])[:400])"
timeout 900 python tools/config_sweep.py --out $OUT/config_sweep.json > $OUT/config_sweep.log 2>&1; tail -3 $OUT/config_sweep.log | cut -c1-600
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/prof -o p -- python $R/bench.py --no-conk --cpu-cells 0 --no-measure-traffic --no-whole-fit --no-rccl-world1 --no-pivot > $OUT/bench_under_rocprof.json 2> $OUT/prof.log
python $R/tools/rocpd_summary.py $(find $OUT/prof -name "*.db" | head -1) > $OUT/bench_kernel_stats.md 2> $OUT/bench_kernel_stats.err
rm -rf $OUT/prof
head -12 $OUT/bench_kernel_stats.md | cut -c1-200
