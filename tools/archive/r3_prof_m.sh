#!/bin/bash
# GPU box: phase profile (GUSTO_PROFILE build) of model $1 at batch $2.  tools/r3_prof_m.sh <model> <B> [extra flags]
M=${1:-2}; B=${2:-8192}; shift; shift
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out
tools/build_dev.sh $M -DGUSTO_PROFILE "$@" > gpurun_out/build_m$M.log 2>&1 || { cat gpurun_out/build_m$M.log; exit 1; }
grep -E "VGPRs|Scratch|Occupancy" gpurun_out/build_m$M.log
python tools/gpu_prof.py $B $M
python tools/gpu_prof2.py $B $M
