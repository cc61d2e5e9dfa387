#!/bin/bash
# round 6, GPU call E: the asynchronous direct form + speculative field update; then the whole GPU suite
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T=${1:-e}
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -s -x -k "solve or minnorm or deflated or pinv or direct" > gpurun_out/r6${T}_kernels.log 2>&1; echo "kernels rc=$?"
grep -E "^m=|direct form|passed|failed|^FAILED|^E  " gpurun_out/r6${T}_kernels.log | tail -30
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
python tools/rocpd_timeline.py "$DB" assemble_kernel 400 > gpurun_out/r6${T}_small_m_timeline.md; python tools/rocpd_timeline.py "$DB" estep_min_kernel 120 > gpurun_out/r6${T}_small_m_step_timeline.md
tail -3 gpurun_out/r6${T}_small_m_timeline.md
tail -2 gpurun_out/r6${T}_trace.log
rm -rf gpurun_out/r6${T}_prof
if [ "${2:-}" = "full" ]; then
timeout 2400 python -m pytest tests -q -m gpu -x > gpurun_out/r6${T}_gpu_tests.txt 2>&1; echo "gpu suite rc=$?"; tail -5 gpurun_out/r6${T}_gpu_tests.txt
fi
