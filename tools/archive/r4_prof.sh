#!/bin/bash
# phase-cycle profiles (-DGUSTO_PROFILE builds) of the four GuSTO kernels; the in-tree library is REPLACED by the last dev build
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out/r04e
for m in "2 1024" "3 512" "1 16384" "0 4096"; do
  set -- $m
  bash tools/build_dev.sh $1 -DGUSTO_PROFILE > gpurun_out/r04e/build_m$1.log 2>&1
  timeout 600 python tools/gpu_prof.py $2 $1 > gpurun_out/r04e/prof_m$1.log 2>&1
  head -3 gpurun_out/r04e/prof_m$1.log
done
