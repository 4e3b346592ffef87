#!/bin/bash
# on the GPU box: (1) bitwise regression of the paths the wave-per-chain kernel must not touch (devdata/bits_*: models 0, 1 with the shipped
# library, models 2, 3 one wave per problem through the dev variants), (2) where the two-wave kernel stops paying (launch.hpp's policy)
cd $GRAFT_REPO_ROOT
echo "== bits (shipped library)"
timeout 300 python tools/gpu_bits.py 0 1024 check 2>&1 | tail -1
timeout 300 python tools/gpu_bits.py 1 4096 check 2>&1 | tail -1
cp gusto.jl_amd/libgusto_hip.so /tmp/libgusto_hip.keep
cp gusto.jl_amd/variants/w2_m2.so gusto.jl_amd/libgusto_hip.so
echo "== bits (one wave, variant)"; GUSTO_DEV_W2=0 timeout 300 python tools/gpu_bits.py 2 512 check 2>&1 | tail -1
for B in 2048 4096 8192; do for w in 0 2; do echo "-- m2 B=$B W2=$w"; GUSTO_DEV_W2=$w timeout 200 python tools/gpu_time.py 2 $B 50 2>&1 | tail -1; done; done
for B in 128 512; do for w in 2 4; do echo "-- m2 B=$B W2=$w"; GUSTO_DEV_W2=$w timeout 200 python tools/gpu_time.py 2 $B 50 2>&1 | tail -1; done; done
cp gusto.jl_amd/variants/w2_m3.so gusto.jl_amd/libgusto_hip.so
echo "== bits (one wave, variant)"; GUSTO_DEV_W2=0 timeout 300 python tools/gpu_bits.py 3 256 check 2>&1 | tail -1
for B in 2048 4096; do for w in 0 2; do echo "-- m3 B=$B W2=$w"; GUSTO_DEV_W2=$w timeout 200 python tools/gpu_time.py 3 $B 50 2>&1 | tail -1; done; done
for B in 128 512; do for w in 2 4; do echo "-- m3 B=$B W2=$w"; GUSTO_DEV_W2=$w timeout 200 python tools/gpu_time.py 3 $B 50 2>&1 | tail -1; done; done
cp /tmp/libgusto_hip.keep gusto.jl_amd/libgusto_hip.so
