#!/bin/bash
# on the GPU box: the wave-per-chain kernel (scp_kernel_w2, 2 or 4 waves per problem) of the variants w2_m2 / w2_m3 (-DGUSTO_DEV_KNOBS)
# against the one-wave kernel of the same build (GUSTO_DEV_W2 = 0 / 2 / 4), parity first.
#   tools/w2_ab.sh [batches of model 3] [batches of model 2] [modes]
cd $GRAFT_REPO_ROOT
B3=${1:-"256 512 1024 2048"}; B2=${2:-"256 512 1024"}; MODES=${3:-"0 2 4"}
cp gusto.jl_amd/libgusto_hip.so /tmp/libgusto_hip.keep
for m in 3 2; do
cp gusto.jl_amd/variants/w2_m$m.so gusto.jl_amd/libgusto_hip.so
for w in $MODES; do [ $w != 0 ] && { echo "== parity model $m B=256 W2=$w"; GUSTO_DEV_W2=$w timeout 300 python tools/parity_sweep.py 256 $m 2>&1 | tail -8; }; done
done
cp gusto.jl_amd/variants/w2_m3.so gusto.jl_amd/libgusto_hip.so
for B in $B3; do for w in $MODES; do echo "-- m3 B=$B W2=$w"; GUSTO_DEV_W2=$w timeout 200 python tools/gpu_time.py 3 $B 50 2>&1 | tail -1; done; done
cp gusto.jl_amd/variants/w2_m2.so gusto.jl_amd/libgusto_hip.so
for B in $B2; do for w in $MODES; do echo "-- m2 B=$B W2=$w"; GUSTO_DEV_W2=$w timeout 200 python tools/gpu_time.py 2 $B 50 2>&1 | tail -1; done; done
cp /tmp/libgusto_hip.keep gusto.jl_amd/libgusto_hip.so
