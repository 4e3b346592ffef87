#!/bin/bash
# on the GPU box: scp_kernel_w2 against the one-wave kernel at larger batches; phase profile of both (variant w2p_m3)
cd $GRAFT_REPO_ROOT
cp gusto.jl_amd/libgusto_hip.so /tmp/libgusto_hip.keep
cp gusto.jl_amd/variants/w2_m3.so gusto.jl_amd/libgusto_hip.so
for B in 4096; do for w in 0 1; do echo "-- m3 B=$B W2=$w"; GUSTO_DEV_W2=$w timeout 200 python tools/gpu_time.py 3 $B 50 2>&1 | tail -1; done; done
cp gusto.jl_amd/variants/w2_m2.so gusto.jl_amd/libgusto_hip.so
for B in 2048 4096 8192; do for w in 0 1; do echo "-- m2 B=$B W2=$w"; GUSTO_DEV_W2=$w timeout 200 python tools/gpu_time.py 2 $B 50 2>&1 | tail -1; done; done
cp gusto.jl_amd/variants/w2p_m3.so gusto.jl_amd/libgusto_hip.so
for w in 0 1; do echo "== profile m3 B=256 W2=$w"; GUSTO_DEV_W2=$w timeout 200 python tools/gpu_prof.py 256 3 2>&1 | tail -40; done
cp /tmp/libgusto_hip.keep gusto.jl_amd/libgusto_hip.so
