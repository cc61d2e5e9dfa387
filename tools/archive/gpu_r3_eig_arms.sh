#!/bin/bash
# measurement arms of jac_eig_kernel (device clocks inside the kernel): base / no J accumulation / no S update / neither
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/r3g; mkdir -p $OUT
cd $R
export MVF_LR_TIMING=1
for v in base noj nos nojs; do
  export MVF_LIB_PATH=$R/tools/ab/libmvf_$v.so
  timeout 300 python tools/minnorm_probe.py 3000 60000 3 0.02 lowrank > $OUT/probe_$v.json 2> $OUT/probe_$v.err
  echo "== $v"; grep -A1 "mvf_solve_minnorm_lr" $OUT/probe_$v.err | tail -4
done
