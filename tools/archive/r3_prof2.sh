#!/bin/bash
# GPU box: phase profile of the CURRENT sources (builds the GUSTO_PROFILE dev lib of model $1, restores the lib after).
M=${1:-0}; shift
cd $GRAFT_REPO_ROOT
cp gusto.jl_amd/libgusto_hip.so /tmp/lib_keep.so
tools/build_dev.sh $M -DGUSTO_PROFILE "$@" > /dev/null 2>&1
python tools/gpu_prof.py 4096 $M 2>&1 | cat
cp /tmp/lib_keep.so gusto.jl_amd/libgusto_hip.so
