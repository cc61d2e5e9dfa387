#!/bin/bash
# the bench line and the same command under rocprofv3 --kernel-trace (summary only travels back)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final; mkdir -p $OUT
cd $R
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
python -c "import json;d=json.load(open('$OUT/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['achieved'], d['roofline']['avg_kernel_ms'], d['solve']['avg_ms'], d['f64']['value'], d['f64']['roofline']['achieved'], d['f64']['roofline']['avg_kernel_ms'])"
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/prof -o p -- python $R/bench.py --no-conk --cpu-cells 0 > $OUT/bench_under_rocprof.json 2> $OUT/prof.log
python $R/tools/rocpd_summary.py $OUT/prof/p_results.db > $OUT/bench_kernel_stats.md 2> $OUT/bench_kernel_stats.err
rm -rf $OUT/prof
head -5 $OUT/bench_kernel_stats.md | cut -c1-160
python -c "import json;d=json.load(open('$OUT/bench_under_rocprof.json'));print('under rocprof', d['ms_per_step'], d['roofline']['avg_kernel_ms'], d['f64']['roofline']['avg_kernel_ms'])"
