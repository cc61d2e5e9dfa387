#!/bin/bash
# first GPU contact of the deflated solve: field agreement with the Jacobi path + phase times
mkdir -p gpurun_out/lrd
export MVF_DEV_KNOBS=1 MVF_LR_TIMING=1
for cfg in "2000 40000" "3000 60000"; do
  set -- $cfg
  timeout 300 python tools/minnorm_probe.py $1 $2 6 0.02 deflated,lowrank > gpurun_out/lrd/probe_$1.json 2> gpurun_out/lrd/probe_$1.err
  echo "rc $?"; tail -c 1500 gpurun_out/lrd/probe_$1.json; grep -c lrd gpurun_out/lrd/probe_$1.err; grep "lrd\|Error\|error" gpurun_out/lrd/probe_$1.err | head -8
done
