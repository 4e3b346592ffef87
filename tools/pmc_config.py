"""HBM traffic summary of ONE workload other than the headline config from its two rocprofv3 --pmc passes
(FETCH_SIZE, WRITE_SIZE; tools/profile_round.sh):  python tools/pmc_config.py <dir with FETCH_SIZE/WRITE_SIZE csv + logs> <out.json>
Same corrections as tools/pmc_summarize.py (MI355X_MICROARCH.md, HBM section): FETCH_SIZE x2 on gfx950, calibrated in the
same run against a 1 GiB elementwise read."""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench
src, dst = sys.argv[1], sys.argv[2]


def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc


out = {}
for ln in open(os.path.join(src, "FETCH_SIZE.log")).read().splitlines():
    if ln.startswith("kernel_ms"):
        p = ln.split()
        out["kernel_ms"], ipm, scp, cfg, algo, B = float(p[1]), int(p[3]), int(p[5]), int(p[7]), p[9], int(p[11])
f = per_kernel(os.path.join(src, "FETCH_SIZE_counter_collection.csv"))
w = per_kernel(os.path.join(src, "WRITE_SIZE_counter_collection.csv"))
kk = [k for k in f if ("trajopt_kernel" if algo == "trajopt" else "scp_kernel") in k][0]
cal = [k for k in f if "vectorized_elementwise" in k]
cal_k = ([k for k in cal if "CUDAFunctorOnSelf_add" in k] or [max(cal, key=lambda k: f[k]["FETCH_SIZE"])])[0]
c = bench.CONFIGS[cfg]
import gusto_jl_amd as g
n, m = g.MODEL_DIMS[getattr(g, c["model"])]
b_kkt, b_lin = bench.algorithmic_bytes(n, m + (n if algo == "trajopt" else 0), c["N"])
out.update(workload=c["name"].replace(f"batch={c['B']}", f"batch={B}") + (", TrajOpt" if algo == "trajopt" else "") + ", ONE launch (tools/pmc_probe.py)",
           kernel=kk.replace("void ", "").split("(")[0], FETCH_SIZE_kb_raw=f[kk]["FETCH_SIZE"], WRITE_SIZE_kb_raw=w[kk]["WRITE_SIZE"],
           calibration={"FETCH_SIZE_kb": f[cal_k]["FETCH_SIZE"], "expected_read_kb": 1048576,
                        "note": "FETCH_SIZE reports 1/2 of the bytes read on gfx950 (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE is exact"})
out["fetch_bytes"] = 2.0 * 1024 * out["FETCH_SIZE_kb_raw"]
out["write_bytes"] = 1024 * out["WRITE_SIZE_kb_raw"]
out["traffic_bytes_per_launch"] = out["fetch_bytes"] + out["write_bytes"]
out["hbm_gbs"] = out["traffic_bytes_per_launch"] / (out["kernel_ms"] * 1e-3) / 1e9
out["kkt_solves"], out["scp_iters"] = ipm, scp
out["algorithmic_bytes_per_launch"] = b_kkt * ipm + b_lin * scp
out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
out["traffic_bytes_per_kkt_solve"] = out["traffic_bytes_per_launch"] / max(1, ipm)
json.dump(out, open(dst, "w"), indent=1)
print(json.dumps({k: v for k, v in out.items() if k != "calibration"}))
