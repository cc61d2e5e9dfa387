#!/bin/bash
# round 4: the default bench line (headline + float64 + pivot sub-run + con_K + CPU baseline + 10-step parity), the same
# command under rocprofv3 --kernel-trace (summary only), and the configuration sweep (C2, C3, C5, M = 100)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r4h; mkdir -p $OUT
cd $R
timeout 1500 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"; 
python -c "
import json;d=json.load(open('$OUT/bench.json'));print(d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['traffic'], d['solve']['avg_ms'], d['f64']['value'], d['f64']['roofline']['frac'], d['con_k']['GBps'], d['speedup_vs_cpu_baseline'], d['cpu_baseline']['cores']); print(json.dumps(d['parity'])); print(json.dumps({k:v for k,v in d['pivot_subset'].items() if k not in ('roofline','solve','note')})); print(d['pivot_subset']['roofline']['frac'], d['pivot_subset']['roofline']['traffic'], d['pivot_subset']['solve']['avg_ms'])"
cd /tmp; export TMPDIR=/tmp
timeout 900 rocprofv3 --kernel-trace -d $OUT/prof -o p -- python $R/bench.py --no-conk --cpu-cells 0 > $OUT/bench_under_rocprof.json 2> $OUT/prof.log
python $R/tools/rocpd_summary.py $OUT/prof/p_results.db > $OUT/bench_kernel_stats.md 2> $OUT/bench_kernel_stats.err
rm -rf $OUT/prof
head -8 $OUT/bench_kernel_stats.md | cut -c1-170
cd $R
timeout 1200 python tools/config_sweep.py --out $OUT/config_sweep.json > $OUT/config_sweep.log 2>&1; echo "sweep rc $?"; grep -E "^C2|^C3|^C5|default_M100_50k_float32" $OUT/config_sweep.log | cut -c1-400
