#!/bin/bash
# round 6, GPU call C: + Jacobi round skipping / sweep pre-test, pivoted-panel kernels (potrf64-style factor, quad substitution)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T=${1:-c}
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -s -k "solve or minnorm or deflated or pinv or direct" > gpurun_out/r6${T}_kernels.log 2>&1; echo "kernels rc=$?"
grep -E "^m=|direct form|passed|failed|^FAILED" gpurun_out/r6${T}_kernels.log | tail -30
timeout 900 python tools/small_m_probe.py --out gpurun_out/r06_small_m_probe_${T}.json > gpurun_out/r6${T}_small_m.log 2>&1; echo "small_m rc=$?"
python - <<PY
import json
d=json.load(open('gpurun_out/r06_small_m_probe_${T}.json'))
for k,v in d.items():
    try:
        print(k, {m:(round(v[m]['steady_step_ms'],3), round(v[m]['steady_solve_ms'],3), v[m]['rank'][-1], (v[m]['block'] or [None])[-1]) for m in ('full','deflated')}, v['field_maxrel_between_methods'])
    except Exception as e:
        print(k, "??", e)
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/r6${T}_prof" -o p -- python "$R/tools/small_m_trace.py" deflated > "$R/gpurun_out/r6${T}_trace.log" 2>&1); echo "trace rc=$?"
DB=$(find gpurun_out/r6${T}_prof -name "*.db" | head -1)
python tools/rocpd_timeline.py "$DB" assemble_kernel 400 > gpurun_out/r6${T}_small_m_timeline.md
tail -3 gpurun_out/r6${T}_small_m_timeline.md
rm -rf gpurun_out/r6${T}_prof
timeout 600 python tools/minnorm_probe.py 3000 60000 10 0.02 deflated > gpurun_out/r6${T}_minnorm3000.log 2>&1; echo "minnorm_probe rc=$?"; tail -1 gpurun_out/r6${T}_minnorm3000.log | cut -c1-400
timeout 600 python tools/lr_phase_probe.py 3000 60000 10 > gpurun_out/r6${T}_phase3000.log 2>&1; grep -E "mvf_solve|^\[" gpurun_out/r6${T}_phase3000.log | tail -5
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/r6${T}_prof3" -o p -- python "$R/tools/lr_phase_probe.py" 3000 60000 8 > "$R/gpurun_out/r6${T}_trace3.log" 2>&1); echo "trace3 rc=$?"
DB=$(find gpurun_out/r6${T}_prof3 -name "*.db" | head -1)
python tools/rocpd_timeline.py "$DB" assemble_kernel 600 > gpurun_out/r6${T}_m3000_timeline.md
tail -2 gpurun_out/r6${T}_m3000_timeline.md
rm -rf gpurun_out/r6${T}_prof3
