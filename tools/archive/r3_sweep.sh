#!/bin/bash
# GPU box: A/B of build flags for one model.  tools/r3_sweep.sh <model> <B> "<flags A>" "<flags B>" ...
cd $GRAFT_REPO_ROOT
M=$1; B=$2; shift; shift
N=50; [ $M = 1 ] && N=30
for fl in "$@"; do
  tools/build_dev.sh $M $fl > gpurun_out/sw.log 2>&1 || { echo "build [$fl] failed"; tail -5 gpurun_out/sw.log; continue; }
  echo "== flags [$fl]  $(grep -E 'Scratch' gpurun_out/sw.log | head -1 | sed 's/.*remark: *//')"
  python tools/gpu_bits.py $M 512 check
  for i in 1 2 3; do python tools/gpu_time.py $M $B $N | sed 's/ipm total.*//'; done
done
