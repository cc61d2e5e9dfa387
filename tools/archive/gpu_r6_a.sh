#!/bin/bash
# round 6, GPU call A: the new block-column kernels of the blocked Cholesky (potrf64 / interleaved trsm / syrk + potrf)
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
R="$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
timeout 1200 python -m pytest tests/test_gpu_kernels.py -x -q -k "solve or minnorm or deflated or pinv or direct" > gpurun_out/r6a_kernels.log 2>&1; echo "kernels rc=$?"
tail -5 gpurun_out/r6a_kernels.log
timeout 900 python tools/small_m_probe.py --out gpurun_out/r06_small_m_probe_a.json > gpurun_out/r6a_small_m.log 2>&1; echo "small_m rc=$?"
python - <<'PY'
import json
d=json.load(open('gpurun_out/r06_small_m_probe_a.json'))
for k,v in d.items():
    print(k, {m:(round(v[m]['steady_step_ms'],3), round(v[m]['steady_solve_ms'],3), v[m]['rank'][-1], (v[m]['block'] or [None])[-1]) for m in ('full','deflated')}, v['field_maxrel_between_methods'])
PY
(cd /tmp && timeout 300 rocprofv3 --kernel-trace -d "$R/gpurun_out/r6a_prof" -o p -- python "$R/tools/small_m_trace.py" deflated > "$R/gpurun_out/r6a_trace.log" 2>&1); echo "trace rc=$?"
DB=$(find gpurun_out/r6a_prof -name "*.db" | head -1)
python tools/rocpd_timeline.py "$DB" assemble_kernel 400 > gpurun_out/r6a_small_m_timeline.md
tail -3 gpurun_out/r6a_small_m_timeline.md
rm -rf gpurun_out/r6a_prof
timeout 600 python tools/minnorm_probe.py 3000 60000 10 0.02 deflated,lowrank > gpurun_out/r6a_minnorm3000.log 2>&1; echo "minnorm_probe rc=$?"; tail -15 gpurun_out/r6a_minnorm3000.log
