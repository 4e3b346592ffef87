#!/bin/bash
# GPU box: dev build, bitwise check against devdata/ and timing for each model given (default 2 3).  tools/r3_models.sh [models...] [-- flags]
cd $GRAFT_REPO_ROOT
MODELS=""; FLAGS=""
while [ $# -gt 0 ]; do if [ "$1" = "--" ]; then shift; FLAGS="$@"; break; fi; MODELS="$MODELS $1"; shift; done
[ -z "$MODELS" ] && MODELS="2 3"
for M in $MODELS; do
  tools/build_dev.sh $M $FLAGS > gpurun_out/b$M.log 2>&1 || { tail -5 gpurun_out/b$M.log; continue; }
  grep -E "Scratch" gpurun_out/b$M.log | head -1 | sed 's/.*remark: *//'
  python tools/gpu_bits.py $M 512 check
  B=8192; N=50; [ $M = 3 ] && B=2048; [ $M = 1 ] && B=65536 && N=30; [ $M = 0 ] && B=4096
  python tools/gpu_time.py $M $B $N | sed 's/ipm total.*//'; python tools/gpu_time.py $M $B $N | sed 's/ipm total.*//'
done
