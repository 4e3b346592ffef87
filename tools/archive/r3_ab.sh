#!/bin/bash
# GPU box: bitwise check against devdata/ baselines and timing of the dev build of model $1 (default 0).  tools/r3_ab.sh <model> [B]
M=${1:-0}; B=${2:-4096}
cd $GRAFT_REPO_ROOT
python tools/gpu_bits.py $M 512 check
N=50; [ $M = 1 ] && N=30
for i in 1 2 3; do python tools/gpu_time.py $M $B $N; done
