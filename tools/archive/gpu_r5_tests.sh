#!/bin/bash
# round 5: the whole GPU suite with its printed parity lines
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/r5t
rm -f gpurun_out/headline_solve_parity.jsonl
S=$(date +%s)
timeout 2400 python -m pytest tests -q -m gpu -s > gpurun_out/r5t/tests.log 2>&1; echo "tests rc=$? in $(( $(date +%s) - S )) s"
tail -4 gpurun_out/r5t/tests.log
grep -E "FAILED|ERROR" gpurun_out/r5t/tests.log | head -20
python tools/parity_table.py gpurun_out/r5t/tests.log > gpurun_out/r5t/parity_table.md; wc -l gpurun_out/r5t/parity_table.md
