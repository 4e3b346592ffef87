#!/bin/bash
# GPU box: finite slices of level-0 problems (GUSTO_SLICE_Q) for the dev build of model $1.  tools/r3_slice.sh <model> <q...>
cd $GRAFT_REPO_ROOT
M=$1; shift
B=4096; N=50; [ $M = 1 ] && B=65536 && N=30; [ $M = 2 ] && B=8192; [ $M = 3 ] && B=2048
tools/build_dev.sh $M > gpurun_out/bs.log 2>&1 || { tail -5 gpurun_out/bs.log; exit 1; }
for q in "$@"; do
  echo "q $q"; GUSTO_SLICE_Q=$q timeout 300 python tools/gpu_bits.py $M 512 check | sed 's/; max.*kernel/ kernel/'
  for i in 1 2; do GUSTO_SLICE_Q=$q timeout 300 python tools/gpu_time.py $M $B $N | sed 's/ipm total.*//'; done
done
