#!/bin/bash
# SQ counter passes over one launch of the bench workload (tools/pmc_probe.py); outputs under gpurun_out/<tag>/sq
TAG=${1:-sq}
OUT=$GRAFT_REPO_ROOT/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
i=0
for set in "SQ_IFETCH SQ_IFETCH_LEVEL SQ_INSTS SQ_WAVE_CYCLES" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INST_CYCLES_VMEM_RD" \
           "SQ_INST_LEVEL_LDS SQ_INSTS_LDS SQ_INST_LEVEL_SMEM SQ_INSTS_SMEM" \
           "SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_FLAT SQ_ACTIVE_INST_MISC SQ_INSTS_BRANCH" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_VMEM_TA_CMD_FIFO_FULL SQ_VMEM_WR_TA_DATA_FIFO_FULL SQ_INSTS_FLAT" \
           "TCP_TCC_READ_REQ_sum TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_WRITE_REQ_sum" \
           "TCC_HIT_sum TCC_MISS_sum TCC_REQ_sum"; do
  i=$((i+1))
  rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/sq -o "set$i" -- python tools/pmc_probe.py > "$OUT/set$i.log" 2>&1
done
python3 - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/sq/*counter_collection.csv")):
    acc=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'scp_kernel' in r['Kernel_Name']: acc[r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items(): print(f"{k:32s} {v:.4g}")
PY
