#!/bin/bash
# round 5, GPU call E: the direct form of the small deflated solve
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -s -k "minnorm or deflated" > gpurun_out/r5e_kernels.log 2>&1; echo "kernels rc=$?"
grep -E "^m=|direct form|passed|failed|Error|assert" gpurun_out/r5e_kernels.log | tail -24
timeout 900 python tools/small_m_probe.py --out gpurun_out/r05_small_m_probe.json > gpurun_out/r5e_small_m.log 2>&1; echo "small_m rc=$?"
grep -E "mvf_solve_minnorm_lr" gpurun_out/r5e_small_m.log | head -12
python - <<'PY'
import json
d=json.load(open('gpurun_out/r05_small_m_probe.json'))
for k,v in d.items():
    print(k, {m:(round(v[m]['steady_step_ms'],3), round(v[m]['steady_solve_ms'],3), v[m]['rank'][-1], (v[m]['block'] or [None])[-1]) for m in ('full','deflated')}, v['field_maxrel_between_methods'])
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/r5e_prof" -o p -- python "$R/tools/small_m_trace.py" deflated > "$R/gpurun_out/r5e_trace.log" 2>&1); echo "trace rc=$?"
DB=$(find gpurun_out/r5e_prof -name "*.db" | head -1)
python tools/rocpd_timeline.py "$DB" assemble_kernel 400 > gpurun_out/r5e_small_m_timeline.md
tail -3 gpurun_out/r5e_small_m_timeline.md
rm -rf gpurun_out/r5e_prof
