#!/bin/bash
# the whole GPU suite, then the bench line (what the driver runs at round end)
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/full; mkdir -p $OUT
cd $R
timeout 2400 python -m pytest tests -x -q -m gpu -s > $OUT/tests.log 2>&1; echo "tests rc $?"
tail -4 $OUT/tests.log
timeout 900 python bench.py > $OUT/bench.json 2> $OUT/bench.err; echo "bench rc $?"
cat $OUT/bench.json
