#!/bin/bash
# on the GPU box: FETCH_SIZE / WRITE_SIZE of the calibration kernels (tools/ub/fetch_calib.hip) -> gpurun_out/fetch_calib.txt
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
D=/tmp/fc; rm -rf $D; mkdir -p $D
for c in FETCH_SIZE WRITE_SIZE; do
  [ -x tools/ub/fetch_calib ] || /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -o tools/ub/fetch_calib tools/ub/fetch_calib.hip
  rocprofv3 --pmc $c --kernel-trace --output-format csv -d $D -o $c -- tools/ub/fetch_calib > $D/$c.log 2>&1
done
python3 - <<'PY' | tee gpurun_out/fetch_calib.txt
import csv, glob, collections
GiB = float(1 << 30)
exp = {"read8_coalesced": ("FETCH_SIZE", 1.0), "read16_coalesced": ("FETCH_SIZE", 1.0), "read8_records": ("FETCH_SIZE", 0.75),
       "write8_coalesced": ("WRITE_SIZE", 1.0), "write8_records": ("WRITE_SIZE", 0.75)}
acc = collections.defaultdict(float)
for c in ("FETCH_SIZE", "WRITE_SIZE"):
    for f in glob.glob(f"/tmp/fc/**/{c}_counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c: acc[(r["Kernel_Name"].split("(")[0], c)] += float(r["Counter_Value"])
print("# kernel: counter (KB) -> bytes reported / bytes touched by the kernel")
for k, (c, frac) in exp.items():
    v = [val for (name, cc), val in acc.items() if k in name and cc == c]
    if v: print(f"{k:20s} {c} {v[0]:.0f} KB = {v[0] * 1024 / (frac * GiB):.3f} x the {frac:.2f} GiB it touches" + (f" ({v[0] * 1024 / GiB:.3f} x the 1 GiB it spans)" if frac < 1 else ""))
PY
