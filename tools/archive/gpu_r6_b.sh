#!/bin/bash
# round 6, GPU call B: + skinny GEMM, corrected quotient in potrf64_store
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -q -s -k "solve or minnorm or deflated or pinv or direct" > gpurun_out/r6b_kernels.log 2>&1; echo "kernels rc=$?"
grep -E "^m=|direct form|passed|failed|^FAILED" gpurun_out/r6b_kernels.log | tail -30
timeout 900 python tools/small_m_probe.py --out gpurun_out/r06_small_m_probe_b.json > gpurun_out/r6b_small_m.log 2>&1; echo "small_m rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_small_m_probe_b.json'))
for k,v in d.items():
    print(k, {m:(round(v[m]['steady_step_ms'],3), round(v[m]['steady_solve_ms'],3), v[m]['rank'][-1], (v[m]['block'] or [None])[-1]) for m in ('full','deflated')}, v['field_maxrel_between_methods'])
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/r6b_prof" -o p -- python "$R/tools/small_m_trace.py" deflated > "$R/gpurun_out/r6b_trace.log" 2>&1); echo "trace rc=$?"
DB=$(find gpurun_out/r6b_prof -name "*.db" | head -1)
python tools/rocpd_timeline.py "$DB" assemble_kernel 400 > gpurun_out/r6b_small_m_timeline.md
tail -3 gpurun_out/r6b_small_m_timeline.md
rm -rf gpurun_out/r6b_prof
timeout 600 python tools/minnorm_probe.py 3000 60000 10 0.02 deflated > gpurun_out/r6b_minnorm3000.log 2>&1; echo "minnorm_probe rc=$?"; tail -3 gpurun_out/r6b_minnorm3000.log | cut -c1-600
