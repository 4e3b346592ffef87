#!/bin/bash
# on the GPU box: phase profile of the one-wave and the wave-per-chain kernel (variant w2p_m$1, -DGUSTO_PROFILE -DGUSTO_PROFILE_COARSE)
#   tools/w2_prof.sh [model] [B] [modes]
cd $GRAFT_REPO_ROOT
m=${1:-3}; B=${2:-256}; MODES=${3:-"0 2 4"}
cp gusto.jl_amd/libgusto_hip.so /tmp/libgusto_hip.keep
cp gusto.jl_amd/variants/w2p_m$m.so gusto.jl_amd/libgusto_hip.so
for w in $MODES; do echo "== profile m$m B=$B W2=$w"; GUSTO_DEV_W2=$w timeout 200 python tools/gpu_prof.py $B $m 2>&1 | grep -v " 0.0%"; done
cp /tmp/libgusto_hip.keep gusto.jl_amd/libgusto_hip.so
