#!/bin/bash
# on the GPU box: tools/r5_ab.sh <config 2..5> <variant> [<variant> ...] -- per variant library (gusto.jl_amd/variants/<v>.so):
# kernel time of the config's batch (two solves), bit check against devdata (if saved), and the memory-side traffic of one
# launch from two rocprofv3 --pmc passes (FETCH_SIZE x 2 on gfx950, WRITE_SIZE).  One line per variant in gpurun_out/r5_ab.txt.
cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
CFG=$1; shift
case $CFG in 2) M=0; B=4096; N=50;; 3) M=1; B=65536; N=30;; 4) M=2; B=8192; N=50;; 5) M=3; B=2048; N=50;; esac
cp gusto.jl_amd/libgusto_hip.so /tmp/libgusto_hip.keep
OUT=gpurun_out/r5_ab.txt
for v in "$@"; do
  cp gusto.jl_amd/variants/$v.so gusto.jl_amd/libgusto_hip.so || continue
  t=$(timeout 300 python tools/gpu_time.py $M $B $N 2>&1 | tail -1)
  D=/tmp/pmc_$v; rm -rf $D; mkdir -p $D
  for c in FETCH_SIZE WRITE_SIZE; do
    timeout 400 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $D -o $c -- python tools/pmc_probe.py $CFG gusto > $D/$c.log 2>&1
  done
  python3 - "$v" "$D" "$t" >> $OUT <<'PY'
import csv, glob, sys, collections
v, D, t = sys.argv[1:4]
tot = {}
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    f = glob.glob(f"{D}/**/{c}_counter_collection.csv", recursive=True)
    acc = 0.0
    for r in csv.DictReader(open(f[0])) if f else []:
        if "scp_kernel" in r["Kernel_Name"] and r["Counter_Name"] == c: acc += float(r["Counter_Value"])
    tot[c] = acc
ipm = 1
for ln in open(f"{D}/FETCH_SIZE.log"):
    if ln.startswith("kernel_ms"): ipm = int(ln.split()[3])
fb, wb = 2 * 1024 * tot["FETCH_SIZE"], 1024 * tot["WRITE_SIZE"]
print(f"{v:24s} | {t} | fetch {fb/1e9:.1f} GB write {wb/1e9:.1f} GB per KKT {(fb+wb)/ipm/1e3:.0f} KB (fetch {fb/ipm/1e3:.0f} write {wb/ipm/1e3:.0f})")
PY
  tail -1 $OUT
done
cp /tmp/libgusto_hip.keep gusto.jl_amd/libgusto_hip.so
