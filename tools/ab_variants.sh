#!/bin/bash
# on the GPU box: time every gusto.jl_amd/variants/*.so with tools/gpu_time.py ARGS (e.g. tools/ab_variants.sh 1 65536 30)
cd $GRAFT_REPO_ROOT
cp gusto.jl_amd/libgusto_hip.so /tmp/libgusto_hip.keep
for v in gusto.jl_amd/variants/*.so; do
  cp $v gusto.jl_amd/libgusto_hip.so
  echo "== $(basename $v)"
  for r in 1 2; do timeout 300 python tools/gpu_time.py "$@" 2>&1 | tail -1; done
done
cp /tmp/libgusto_hip.keep gusto.jl_amd/libgusto_hip.so
