"""Turns the rocprofv3 CSVs of tools/profile_round.sh into the committed summaries under profiles/.
   python tools/pmc_summarize.py gpurun_out/<tag> <round>"""
import collections, csv, glob, json, os, shutil, sys
src, rnd = sys.argv[1], int(sys.argv[2])
dst = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "profiles")
tag = f"r{rnd:02d}"


def per_kernel(path):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        acc[r["Kernel_Name"]][r["Counter_Name"]] += float(r["Counter_Value"])
    return acc


out = {"round": rnd, "workload": "freeflyerSE2 batch=4096 N=50, ONE gusto_solve launch (tools/pmc_probe.py)"}
log = open(os.path.join(src, "pmc_FETCH_SIZE.log")).read()
for ln in log.splitlines():
    if ln.startswith("kernel_ms"):
        p = ln.split()
        out["kernel_ms"], ipm, scp = float(p[1]), int(p[3]), int(p[5])
f = per_kernel(os.path.join(src, "pmc", "FETCH_SIZE_counter_collection.csv"))
w = per_kernel(os.path.join(src, "pmc", "WRITE_SIZE_counter_collection.csv"))
scp_k = [k for k in f if "scp_kernel" in k][0]
# calibration: the READ is the (a + 1.0) kernel (CUDAFunctorOnSelf_add: reads 1 GiB, writes 1 GiB); the zeros fill only writes
cal = [k for k in f if "vectorized_elementwise" in k]
cal_k = ([k for k in cal if "CUDAFunctorOnSelf_add" in k] or [max(cal, key=lambda k: f[k]["FETCH_SIZE"])])[0]
out["kernel"] = scp_k.replace("void ", "").split("(")[0]
out["FETCH_SIZE_kb_raw"], out["WRITE_SIZE_kb_raw"] = f[scp_k]["FETCH_SIZE"], w[scp_k]["WRITE_SIZE"]
out["calibration"] = {
    "pattern": "torch float64 elementwise over 1 GiB: zeros fill + (a + 1.0)",
    "FETCH_SIZE_kb": f[cal_k]["FETCH_SIZE"], "expected_read_kb": 1048576,
    "kernel": cal_k.split("(")[0][:120],
    "WRITE_SIZE_kb": sum(w[k]["WRITE_SIZE"] for k in cal), "expected_write_kb": 2097152,
    "note": "FETCH_SIZE reports 1/2 of the bytes read on gfx950 (MI355X_MICROARCH.md, HBM section) -> x2; WRITE_SIZE is exact"}
out["fetch_bytes"] = 2.0 * 1024 * out["FETCH_SIZE_kb_raw"]
out["write_bytes"] = 1024 * out["WRITE_SIZE_kb_raw"]
out["traffic_bytes_per_launch"] = out["fetch_bytes"] + out["write_bytes"]
out["hbm_gbs"] = out["traffic_bytes_per_launch"] / (out["kernel_ms"] * 1e-3) / 1e9
out["kkt_solves"], out["scp_iters"] = ipm, scp
out["algorithmic_bytes_per_launch"] = 51600 * ipm + 24000 * scp
out["traffic_over_algorithmic"] = out["traffic_bytes_per_launch"] / out["algorithmic_bytes_per_launch"]
sq = {}
for p in sorted(glob.glob(os.path.join(src, "sq", "*counter_collection.csv"))):
    for k, v in per_kernel(p).items():
        if "scp_kernel" in k:
            sq.update(v)
out["sq_counters_per_launch"] = sq
mf = {}
for m, cfg in ((2, "config4_astrobee_se3"), (3, "config5_manifold")):
    f_ = os.path.join(src, f"mfma_m{m}", "c_counter_collection.csv")
    if os.path.exists(f_):
        for k, v in per_kernel(f_).items():
            if "scp_kernel" in k:
                mf[cfg] = dict(v)
                if v.get("SQ_BUSY_CYCLES"):
                    # SQ_VALU_MFMA_BUSY_CYCLES counts cycles, SQ_BUSY_CYCLES quad-cycles per SE... report the raw ratio to wave cycles
                    mf[cfg]["mfma_busy_over_wave_cycles"] = v.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (4.0 * v.get("SQ_WAVE_CYCLES", 1.0))
out["mfma_counters_per_solve"] = mf
json.dump(out, open(os.path.join(dst, f"{tag}_pmc.json"), "w"), indent=1)
shutil.copy(os.path.join(src, "pmc", "FETCH_SIZE_counter_collection.csv"), os.path.join(dst, f"{tag}_pmc_FETCH_SIZE.csv"))
shutil.copy(os.path.join(src, "pmc", "WRITE_SIZE_counter_collection.csv"), os.path.join(dst, f"{tag}_pmc_WRITE_SIZE.csv"))
shutil.copy(os.path.join(src, "stats", "stats_kernel_stats.csv"), os.path.join(dst, f"{tag}_kernel_stats.csv"))
shutil.copy(os.path.join(src, "stats", "stats_kernel_trace.csv"), os.path.join(dst, f"{tag}_kernel_trace.csv"))
shutil.copy(os.path.join(src, "bench.json"), os.path.join(dst, f"{tag}_bench.json"))
for extra, name in (("bench_overlap2.json", "bench_overlap2.json"), ("configs.jsonl", "configs.jsonl")):
    if os.path.exists(os.path.join(src, extra)):
        shutil.copy(os.path.join(src, extra), os.path.join(dst, f"{tag}_{name}"))
for m, cfg in ((2, "config4_astrobee_se3"), (3, "config5_manifold")):
    f = os.path.join(src, f"stats_m{m}", "stats_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, f"{tag}_kernel_stats_{cfg}.csv"))
for name in ("config3", "config4", "config5", "trajopt_config2"):       # tools/pmc_config.py summaries (profile_round.sh)
    f = os.path.join(src, f"pmc_{name}.json")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, f"{tag}_pmc_{name}.json"))
for d_, name in (("stats_c3", "kernel_stats_config3_dubins.csv"), ("stats_trajopt", "kernel_stats_trajopt_freeflyer.csv")):
    f = os.path.join(src, d_, "stats_kernel_stats.csv")
    if os.path.exists(f):
        shutil.copy(f, os.path.join(dst, f"{tag}_{name}"))
if os.path.exists(os.path.join(src, "bench_trajopt.json")):
    shutil.copy(os.path.join(src, "bench_trajopt.json"), os.path.join(dst, f"{tag}_bench_trajopt.json"))
print(json.dumps({k: v for k, v in out.items() if k != "calibration"}, indent=1))
