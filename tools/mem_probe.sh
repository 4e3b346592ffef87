#!/bin/bash
# memory-system counters of one launch (tools/pmc_probe.py); outputs under gpurun_out/<tag>/mem
TAG=${1:-mem}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for set in "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_LATENCY_sum TCP_TCC_WRITE_REQ_sum" \
           "TA_BUSY_avr TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum TA_ADDR_STALLED_BY_TC_CYCLES_sum" \
           "TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_WRREQ_STALL_sum TCC_EA0_WRREQ_sum" \
           "TCC_BUSY_avr TCC_TAG_STALL_sum TCC_CYCLE_sum TCC_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_TOTAL_READ_sum TCP_TOTAL_WRITE_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" \
           "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum TA_DATA_STALLED_BY_TC_CYCLES_sum TA_ADDR_STALLED_BY_TD_CYCLES_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/mem -o "set$i" -- python tools/pmc_probe.py > "$OUT/set$i.log" 2>&1
done
python3 - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/mem/*counter_collection.csv")):
    acc=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'scp_kernel' in r['Kernel_Name']: acc[r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items(): print(f"{k:40s} {v:.4g}")
PY
