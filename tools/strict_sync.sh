#!/bin/bash
# on the GPU box: the -DGUSTO_STRICT_SYNC builds (gusto.jl_amd/variants/strict{0,1,2,3}.so: every ordering point of a one-wave problem
# is `s_waitcnt vmcnt(0) lgkmcnt(0); s_barrier` on top of the compiler fence) against the saved bit patterns of the default build
cd $GRAFT_REPO_ROOT
cp gusto.jl_amd/libgusto_hip.so /tmp/libgusto_hip.keep
for job in "0 1024" "1 4096" "2 512" "3 256"; do
  set -- $job
  cp gusto.jl_amd/variants/strict$1.so gusto.jl_amd/libgusto_hip.so || continue
  python tools/gpu_bits.py $1 $2 check
done
cp /tmp/libgusto_hip.keep gusto.jl_amd/libgusto_hip.so
