OUT=$GRAFT_REPO_ROOT/gpurun_out/traf
mkdir -p $OUT; export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 200 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/pmc -o $c -- python tools/pmc_probe.py > $OUT/pmc_$c.log 2>&1
done
python3 - <<PY
import csv,collections
for c in ("FETCH_SIZE","WRITE_SIZE"):
    acc=collections.defaultdict(float)
    for r in csv.DictReader(open("$OUT/pmc/%s_counter_collection.csv"%c)):
        if 'scp_kernel' in r['Kernel_Name']: acc[r['Counter_Name']]+=float(r['Counter_Value'])
    print(dict(acc))
PY
grep kernel_ms $OUT/pmc_FETCH_SIZE.log
