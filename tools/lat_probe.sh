#!/bin/bash
# fetch / memory latency counters of one launch (tools/pmc_probe.py); every rocprofv3 run under its own timeout
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-lat}
mkdir -p $OUT
export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
rocprofv3 -L 2>/dev/null | grep -E "Counter_Name|Expression" | grep -B1 -i "LEVEL" | grep Counter_Name | tr -s "\t " " " > $OUT/derived_list.txt
i=0
for set in "InstrFetchLatency" "SQ_IFETCH SQ_WAIT_INST_ANY SQ_WAVE_CYCLES" "VmemLatency" "LdsLatency" "SmemLatency"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/l -o "set$i" -- python tools/pmc_probe.py > "$OUT/set$i.log" 2>&1
done
python3 - <<PY
import csv,collections,glob
for f in sorted(glob.glob("$OUT/l/*counter_collection.csv")):
    acc=collections.defaultdict(list)
    for r in csv.DictReader(open(f)):
        if 'scp_kernel' in r['Kernel_Name']: acc[r['Counter_Name']].append(float(r['Counter_Value']))
    for k,v in acc.items(): print(f"{k:32s} sum {sum(v):.5g}  n {len(v)}")
PY
cat $OUT/derived_list.txt
