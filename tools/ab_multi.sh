#!/bin/bash
# on the GPU box: tools/ab_multi.sh "<variant> <script.py> <args...>" ...   -- runs tools/<script.py> <args> twice against gusto.jl_amd/variants/<variant>.so
cd $GRAFT_REPO_ROOT
cp gusto.jl_amd/libgusto_hip.so /tmp/libgusto_hip.keep
for job in "$@"; do
  set -- $job
  v=$1; sc=$2; shift 2
  cp gusto.jl_amd/variants/$v.so gusto.jl_amd/libgusto_hip.so || continue
  echo "== $job"
  if [ $sc = gpu_time.py ]; then for r in 1 2; do timeout 300 python tools/$sc "$@" 2>&1 | tail -1; done
  else timeout 300 python tools/$sc "$@" 2>&1; fi
done
cp /tmp/libgusto_hip.keep gusto.jl_amd/libgusto_hip.so
