#!/bin/bash
# instruction-cache counters of one launch (tools/pmc_probe.py); every rocprofv3 run under its own timeout
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-icache}
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -oE "SQC_[A-Z_0-9a-z]+" | sort -u | tr "\n" " " > $OUT/sqc_list.txt
i=0
for set in "SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE" \
           "SQC_DCACHE_REQ SQC_DCACHE_HITS SQC_DCACHE_MISSES SQ_INSTS_SMEM" \
           "SQC_TC_REQ SQC_TC_INST_REQ SQC_TC_DATA_READ_REQ SQC_TC_STALL"; do
  i=$((i+1))
  timeout 180 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/ic -o "set$i" -- python tools/pmc_probe.py > "$OUT/set$i.log" 2>&1
done
python3 - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/ic/*counter_collection.csv")):
    acc=collections.defaultdict(float)
    for r in csv.DictReader(open(f)):
        if 'scp_kernel' in r['Kernel_Name']: acc[r['Counter_Name']]+=float(r['Counter_Value'])
    for k,v in acc.items(): print(f"{k:32s} {v:.4g}")
PY
cat $OUT/sqc_list.txt
