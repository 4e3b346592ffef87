#!/bin/bash
# GPU box: phase profile (GUSTO_PROFILE build of model 0) and timing of the normal build.  tools/r3_prof.sh <tag>
TAG=${1:-p}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
cp gusto.jl_amd/libgusto_hip.so /tmp/lib_normal.so
python tools/gpu_time.py 0 4096 50 > $OUT/time.log 2>&1
tools/build_dev.sh 0 -DGUSTO_PROFILE > $OUT/build.log 2>&1
python tools/gpu_prof.py 4096 0 > $OUT/prof.log 2>&1
python tools/gpu_prof2.py 4096 0 >> $OUT/prof.log 2>&1
cp /tmp/lib_normal.so gusto.jl_amd/libgusto_hip.so
cat $OUT/time.log $OUT/prof.log
