#!/bin/bash
# TA / TCP (vector memory pipeline) counters of ONE solve of the bench workload; separate rocprofv3 --pmc passes.
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-ta}
mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
i=0
for set in "TA_TA_BUSY_sum TA_TOTAL_WAVEFRONTS_sum" "TA_FLAT_READ_WAVEFRONTS_sum TA_FLAT_WRITE_WAVEFRONTS_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_GATE_EN1_sum" "TCP_TCC_READ_REQ_sum TCP_TCC_WRITE_REQ_sum" \
           "TCP_PENDING_STALL_CYCLES_sum TCP_TCP_TA_DATA_STALL_CYCLES_sum" "GRBM_GUI_ACTIVE GRBM_TA_BUSY" \
           "SQ_VMEM_TA_ADDR_FIFO_FULL SQ_INST_LEVEL_VMEM SQ_WAVE_CYCLES SQ_BUSY_CYCLES" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum"; do
  i=$((i+1))
  timeout 150 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o c -- python tools/pmc_probe.py > $OUT/p$i.log 2>&1
done
python - <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
for p in sorted(glob.glob("$OUT/p*/*counter_collection.csv")):
    for r in csv.DictReader(open(p)):
        k = "scp" if "scp_kernel" in r["Kernel_Name"] else ("cal" if "elementwise" in r["Kernel_Name"] else None)
        if k: acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
for k in acc:
    print(k, {c: "%.4g" % v for c, v in acc[k].items()})
PY
grep kernel_ms $OUT/p1.log
