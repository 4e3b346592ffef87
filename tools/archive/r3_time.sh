#!/bin/bash
# GPU box: timing of config 2 with and without the hardest-first order, then the quick parity subset.  tools/r3_time.sh <tag>
TAG=${1:-t}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
cd $GRAFT_REPO_ROOT
for i in 1 2 3; do python tools/gpu_time.py 0 4096 50; done > $OUT/time.log 2>&1
for i in 1 2; do GUSTO_DEV_NO_ORDER=1 python tools/gpu_time.py 0 4096 50; done > $OUT/time_noorder.log 2>&1
python tools/gpu_time.py 2 8192 50 >> $OUT/time.log 2>&1
GUSTO_DEV_NO_ORDER=1 python tools/gpu_time.py 2 8192 50 >> $OUT/time_noorder.log 2>&1
python tools/gpu_time.py 3 2048 50 >> $OUT/time.log 2>&1
GUSTO_DEV_NO_ORDER=1 python tools/gpu_time.py 3 2048 50 >> $OUT/time_noorder.log 2>&1
echo "--- ordered"; cat $OUT/time.log; echo "--- index order"; cat $OUT/time_noorder.log
python -m pytest tests/test_gpu_parity.py -q -x -k "bit_identical or determin or order" 2>&1 | tail -5
