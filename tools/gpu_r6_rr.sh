#!/bin/bash
# (record of a measurement: the "new basis" variants' code was not kept - profiles/r06_rr_ab.md; the libraries under tools/ab/ were
# built from the tree of that moment with -DMVF_RR_OLD_BASIS / -DMVF_RR_TOL=1.7763568394002505e-15)
# round 6, re-entry: the Rayleigh-Ritz problems of the deflated solves - block continued in the basis of the previous call's Ritz
# vectors (new basis) or not (old), stopping criterion sqrt(b) eps (tight) or 2^-36 (relaxed): sweeps of the one-launch 64 x 64
# problem in the EM's steady state, and the EM step of C2 / one C5 organ / M = 300 / 640 / C3-sized M = 2000, same box
cd "$GRAFT_REPO_ROOT"; export TMPDIR=/tmp
mkdir -p gpurun_out/rr
lib() { if [ "$1" = shipped ]; then env "${@:2}"; else env MVF_DEV_KNOBS=1 MVF_LIB_PATH=$PWD/tools/ab/libmvf_$1.so "${@:2}"; fi; }
for v in shipped oldbasis_tight oldbasis_relaxed newbasis_tight; do
  lib $v timeout 200 python tools/rr_sweeps_probe.py 50000 12 2>&1 | grep -v "^$" | cut -c1-240 > gpurun_out/rr/sweeps_$v.log
  echo "$v: $(tail -1 gpurun_out/rr/sweeps_$v.log)"
done
for rep in 1 2; do
 for case in "50000 500" "250000 500" "20000 300" "60000 640" "200000 2000"; do
  for v in shipped oldbasis_tight oldbasis_relaxed newbasis_tight; do
   echo "$v $case: $(lib $v timeout 300 python tools/small_step_profile.py $case float32 200 2>&1 | tail -1)"
  done
 done
done | tee gpurun_out/rr/ab.log
