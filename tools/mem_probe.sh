#!/bin/bash
# L1-miss latency counters of one launch (tools/pmc_probe.py); outputs under gpurun_out/<tag>/mem.
# (Only this counter set is known to be safe here: a TA_BUSY_avr / TA_TA_BUSY_sum pass hung rocprofv3 on this pool.)
TAG=${1:-mem}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 300 rocprofv3 --pmc TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum \
  --kernel-trace --output-format csv -d $OUT/mem -o set1 -- python tools/pmc_probe.py > "$OUT/set1.log" 2>&1
