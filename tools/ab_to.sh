#!/bin/bash
# on the GPU box: tools/ab_to.sh <model 0|2|3> <B> <variant>... -- TrajOpt kernel time + checksum per variant library
cd $GRAFT_REPO_ROOT
M=$1; B=$2; shift 2
cp gusto.jl_amd/libgusto_hip.so /tmp/keep.so
for v in "$@"; do
  cp gusto.jl_amd/variants/$v.so gusto.jl_amd/libgusto_hip.so
  echo "== $v"
  python tools/to_time.py $M $B 2>&1 | tail -1
  python tools/to_time.py $M $B 2>&1 | tail -1
done
cp /tmp/keep.so gusto.jl_amd/libgusto_hip.so
