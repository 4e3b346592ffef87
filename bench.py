#!/usr/bin/env python3
"""bench.py -- converged trajectories/sec of the batched GuSTO SCP hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W [--config {2,3,4,5}] [--scaling {weak,strong}]
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over one batch: straight-line initialisation of every problem,
then gusto_solve (the whole GuSTO loop: linearisation, convex subproblems, trust-region/penalty updates) until
every problem of the batch has stopped; with N > 1 ranks the step ends with the final gather of every rank's
trajectories to rank 0 over RCCL, straight from HBM (north_star: "RCCL only for the batch split and final gather").

Workload (`--config`, numbered as the lines of BASELINE.json `configs`, 1-based; default 2 = configs[1], the
configuration the metric is quoted on):
  2  freeflyerSE2 batch=4096 random initial states, N=50
  3  dubins_car batch=65536, N=30
  4  astrobeeSE3 batch=8192, N=50, ISS corner obstacle set
  5  astrobeeSE3manifold batch=2048, N=50
`--scaling weak` (default): every rank solves its own batch of that size.  `--scaling strong`: the batch of the config
is split over the ranks in contiguous blocks (host.shard_bounds) -- how BASELINE.json words configs 4 and 5 ("8xMI355X
batch-sharded").  No data-path collective either way: the problems are independent.
Inputs are resident in HBM before the timed region (`value`; the PCIe-inclusive rate of SURVEY.md 8(d) is the named
extra `pcie_inclusive_traj_per_s`).  Rank 0 prints ONE JSON line.

`value` = strictly serial steps (one batch in flight), so that the kernel time behind `roofline` is that of a lone
solve.  Named extras in the same line: `overlapped_traj_per_s` (consecutive batches on two handles/streams, the
steady-state serving rate), `pcie_inclusive_traj_per_s`, `yield` (converged / problems) and, for N > 1,
`gather_ms_per_step` (host-side time of the final gather inside a step).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

MAX_ITER = 30
HBM_PEAK_GBS = 8000.0

# BASELINE.json configs (1-based line numbers), the oracle sample per usable host core of the cpu_baseline leg
# (sized for 10-30 s of CPU work) and the workload names
CONFIGS = {
    2: dict(model="FREEFLYER_SE2", N=50, B=4096, cpu_per_core=2048,
            name="freeflyerSE2 batch=4096 random initial states, N=50, fp64 (BASELINE.json configs[1])"),
    3: dict(model="DUBINS_CAR", N=30, B=65536, cpu_per_core=4096,
            name="dubins_car batch=65536, N=30, fp64 (BASELINE.json configs[2])"),
    4: dict(model="ASTROBEE_SE3", N=50, B=8192, cpu_per_core=512,
            name="astrobeeSE3 batch=8192, N=50, ISS corner obstacle set, fp64 (BASELINE.json configs[3])"),
    5: dict(model="ASTROBEE_SE3_MANIFOLD", N=50, B=2048, cpu_per_core=192,
            name="astrobeeSE3manifold batch=2048, N=50, ISS corner obstacle set, fp64 (BASELINE.json configs[4])"),
}


def algorithmic_bytes(n, m, N):
    """SURVEY.md 8(d), restated in DESIGN.md: one KKT solve streams 8*N*(nz(nz+1)/2 + n*nz + 2(nz+n)) bytes, one
    linearisation 8*N*(2(n+m) + n + n^2) bytes (freeflyerSE2 N=50: 51 600 and 24 000)."""
    nz = n + m
    return 8 * N * (nz * (nz + 1) // 2 + n * nz + 2 * (nz + n)), 8 * N * (2 * (n + m) + n + n * n)


def workload(P, g, cfg, B, first):
    """(model id, boxes, spheres, (x_init, goal_lo, goal_hi, tf)) of `B` problems of config `cfg` starting at `first`."""
    c = CONFIGS[cfg]
    model = getattr(g, c["model"])
    if cfg == 2:
        return model, P.freeflyer_env(), None, P.freeflyer_batch(B, first)
    if cfg == 3:
        return model, None, None, P.dubins_batch(B, first)
    bx, sp = P.iss_corner_env(True)
    if cfg == 4:
        return model, bx, sp, P.astrobee_se3_batch(B, first)
    return model, bx, sp, P.astrobee_manifold_batch(B, first)


def usable_cores():
    """Host cores this process may actually use: min(affinity, cgroup cpu.max quota)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(float(quota) / float(period))))
    except Exception:
        pass
    return n


def cpu_baseline(P, g, cfg, n_sample, threads, B_gpu, algo="gusto"):
    """The oracle (a C port, NOT the Julia reference) on a bounded sample of the same workload: all usable host cores
    (OpenMP over problems) and, as SURVEY.md 8(d) asks, ONE core beside it."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gusto_oracle as go
    go.lib()
    N = CONFIGS[cfg]["N"]

    def what(k):   # the sample in words: problems 0..k-1 of the config's seeded generator; the GPU batch is problems 0..B-1 of it
        rel = "the first %d problems of the GPU batch" % k if k <= B_gpu else \
              "problems 0..%d of the config's generator (the GPU batch is its first %d)" % (k - 1, B_gpu)
        return rel

    if algo == "trajopt":      # no OpenMP entry point for TrajOpt in the oracle: one core, problem by problem
        model, boxes, spheres, (x0, glo, ghi, tf) = workload(P, g, cfg, n_sample, 0)
        o = go.OracleTrajOpt(model, N, boxes=boxes, spheres=spheres)
        t0 = time.perf_counter()
        conv = 0
        for b in range(n_sample):
            o.set_problem(x0[b], glo[b], ghi[b], tf[b])
            conv += int(o.solve_trajopt(125)["converged"])
        dt = time.perf_counter() - t0
        return {"value": n_sample / dt, "unit": "problems/s", "converged": conv, "cores": 1, "kind": "port",
                "sample": f"{what(n_sample)}, oracle/libgusto_oracle.so go_solve_trajopt, {dt:.1f} s wall"}
    model, boxes, spheres, (x0, glo, ghi, tf) = workload(P, g, cfg, n_sample, 0)
    t0 = time.perf_counter()
    r = go.solve_batch(model, N, boxes, spheres, x0, glo, ghi, tf, MAX_ITER, threads)
    dt = time.perf_counter() - t0
    out = {"value": float(r["converged"].sum() / dt), "unit": "converged trajectories/s", "cores": threads,
           "kind": "port", "sample": f"{what(n_sample)}, oracle/libgusto_oracle.so (OpenMP over problems), {dt:.1f} s wall"}
    n1 = max(8, n_sample // (4 * max(1, threads)))      # about a quarter of the all-core wall time on one core
    t0 = time.perf_counter()
    r1 = go.solve_batch(model, N, boxes, spheres, x0[:n1], glo[:n1], ghi[:n1], tf[:n1], MAX_ITER, 1)
    dt1 = time.perf_counter() - t0
    out["one_core"] = {"value": float(r1["converged"].sum() / dt1), "unit": "converged trajectories/s", "cores": 1,
                       "sample": f"{what(n1)}, {dt1:.1f} s wall"}
    return out


FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X fp64 vector peak (MI355X_MICROARCH.md)
LDS_PEAK_TBS = 150.0             # aggregate ds_read_b64 / b128 rate with every CU streaming (MI355X_MICROARCH.md, LDS section)
N_CU, CLOCK_GHZ = 256, 2.4
INFINITY_CACHE_BYTES = 256 << 20
# what the FETCH_SIZE / WRITE_SIZE counters behind `roofline.traffic` are (MI355X_MICROARCH.md, HBM section: FETCH_SIZE derives from
# TCC_EA0_RDREQ, "Infinity-Cache hits appear to be counted, not excluded")
TRAFFIC_KIND = ("L2 <-> fabric bytes of one launch (FETCH_SIZE x 2 + WRITE_SIZE: every request that leaves an L2, Infinity-Cache hits "
                "included) -- an upper bound of the HBM traffic, not HBM alone")


def lds_level(pmc, kkt_now, avg_ms):
    """The third bound ("an LDS-resident design additionally reports an LDS-level fraction"), from the committed SQ
    counters of one launch, scaled by this run's KKT solves: `busy` = LDS-array cycles (SQ_LDS_IDX_ACTIVE, bank-conflict cycles
    included) over CU-cycles of the launch; `frac` = bytes moved by the DS wave-instructions -- SQ_INSTS_LDS x 64 lanes x 8 B, the
    kernels' DS traffic is fp64 words; ds_read_b128 counts double -- over time over the 150 TB/s the guide gives for ds_read_b64."""
    sq = (pmc or {}).get("sq_counters_per_launch") or {}
    if "SQ_INSTS_LDS" not in sq or not avg_ms:
        return None
    sc = (kkt_now / float(pmc["kkt_solves"])) if pmc.get("kkt_solves") else 1.0
    t = avg_ms * 1e-3
    out = {"ds_wave_instructions_per_launch": sq["SQ_INSTS_LDS"] * sc,
           "bytes_per_launch_at_8B_per_lane": 512.0 * sq["SQ_INSTS_LDS"] * sc,
           "achieved_tbs": 512.0 * sq["SQ_INSTS_LDS"] * sc / t / 1e12, "peak_tbs": LDS_PEAK_TBS,
           "frac": 512.0 * sq["SQ_INSTS_LDS"] * sc / t / 1e12 / LDS_PEAK_TBS}
    if "SQ_LDS_IDX_ACTIVE" in sq:
        out["lds_array_busy"] = sq["SQ_LDS_IDX_ACTIVE"] * sc / (N_CU * CLOCK_GHZ * 1e9 * t)
        out["bank_conflict_share"] = sq.get("SQ_LDS_BANK_CONFLICT", 0.0) / sq["SQ_LDS_IDX_ACTIVE"]
    return out


def committed_pmc(cfg):
    """The latest committed PMC summary of a config (profiles/r*_pmc.json for config 2, r*_pmc_config{c}.json otherwise;
    separate rocprofv3 --pmc passes, tools/profile_round.sh) or None."""
    import glob
    pat = "r*_pmc_trajopt_config2.json" if cfg == "trajopt" else ("r*_pmc.json" if cfg == 2 else f"r*_pmc_config{cfg}.json")
    pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", pat)))
    if not pmc:
        return None, None
    try:
        return json.load(open(pmc[-1])), "profiles/" + os.path.basename(pmc[-1])
    except Exception:
        return None, None


def live_pmc(cfg, timeout_s=150, algo="gusto"):
    """HBM traffic of ONE launch of the config's batch measured in THIS run: two child `rocprofv3 --pmc` passes (FETCH_SIZE,
    WRITE_SIZE -- each its own run, --kernel-trace only, as MI355X_MICROARCH.md's HBM section prescribes) of
    tools/pmc_probe.py, which solves the same seeded batch once and then reads + writes 1 GiB through a float64 elementwise
    kernel as the calibration.  FETCH_SIZE x 2 on gfx950 (the calibration of this very run is reported next to it), both in
    KiB.  None when rocprofv3 is absent or a pass fails (the caller falls back to the committed summary)."""
    import collections, csv, glob, shutil, subprocess, tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    # (a run that is itself being profiled does not start profilers of its own)
    if any(k.startswith(("ROCPROF", "ROCP_")) for k in os.environ) or "rocprof" in os.environ.get("LD_PRELOAD", ""):
        return None
    res = {}
    try:
        with tempfile.TemporaryDirectory(dir="/tmp") as d:
            env = dict(os.environ, TMPDIR="/tmp")
            for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
                env.pop(k, None)
            for c in ("FETCH_SIZE", "WRITE_SIZE"):
                r = subprocess.run([exe, "--pmc", c, "--kernel-trace", "--output-format", "csv", "-d", d, "-o", c, "--",
                                    sys.executable, os.path.join(ROOT, "tools", "pmc_probe.py"), str(cfg), algo],
                                   cwd="/tmp", env=env, capture_output=True, text=True, timeout=timeout_s)
                for ln in r.stdout.splitlines():
                    if ln.startswith("kernel_ms"):
                        p = ln.split()
                        res["kernel_ms"], res["kkt_solves"], res["scp_iters"] = float(p[1]), int(p[3]), int(p[5])
                f = glob.glob(os.path.join(d, "**", f"{c}_counter_collection.csv"), recursive=True)
                if not f:
                    return None
                acc = collections.defaultdict(float)
                for row in csv.DictReader(open(f[0])):
                    if row["Counter_Name"] == c:
                        acc[row["Kernel_Name"]] += float(row["Counter_Value"])
                k_scp = [k for k in acc if ("trajopt_kernel" if algo == "trajopt" else "scp_kernel") in k]
                k_cal = [k for k in acc if "vectorized_elementwise" in k]
                if not k_scp:
                    return None
                res[c] = acc[k_scp[0]]
                if k_cal:   # the (a + 1.0) kernel reads 1 GiB and writes 1 GiB; the zeros fill only writes
                    res["cal_" + c] = max(acc[k] for k in k_cal) if c == "FETCH_SIZE" else sum(acc[k] for k in k_cal)
    except Exception:
        return None
    if "FETCH_SIZE" not in res or "WRITE_SIZE" not in res:
        return None
    res["fetch_bytes"], res["write_bytes"] = 2.0 * 1024 * res["FETCH_SIZE"], 1024.0 * res["WRITE_SIZE"]
    res["traffic_bytes_per_launch"] = res["fetch_bytes"] + res["write_bytes"]
    if res.get("cal_FETCH_SIZE"):
        res["fetch_size_over_bytes_read"] = 1024.0 * res["cal_FETCH_SIZE"] / float(1 << 30)      # (0.5 on gfx950: the x 2 above)
    if res.get("cal_WRITE_SIZE"):
        res["write_size_over_bytes_written"] = 1024.0 * res["cal_WRITE_SIZE"] / float(2 << 30)
    return res


def valu_fp64(pmc, kkt_now):
    """fp64 VALU + matrix-core flops of one launch from the committed SQ counters (SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 count
    wave instructions: x 64 lanes, an FMA = 2 flops; SQ_INSTS_VALU_MFMA_MOPS_F64 counts 512-flop units), scaled by the
    KKT solves of this run over those of the profiled launch.  None when the summary holds no such counters."""
    sq = (pmc or {}).get("sq_counters_per_launch") or {}
    if "SQ_INSTS_VALU_FMA_F64" not in sq:
        return None
    flops = 64.0 * (2.0 * sq["SQ_INSTS_VALU_FMA_F64"] + sq.get("SQ_INSTS_VALU_MUL_F64", 0.0) + sq.get("SQ_INSTS_VALU_ADD_F64", 0.0))
    flops += 512.0 * sq.get("SQ_INSTS_VALU_MFMA_MOPS_F64", 0.0)
    kkt_prof = pmc.get("kkt_solves")     # (absent in summaries before round 5: the same seeded workload, the same count)
    if kkt_prof:
        flops *= kkt_now / float(kkt_prof)
    return flops


def other_configs(P, g, torch, dev_ord, steps=3, warmup=1, live=True):
    """BASELINE.json configs 3 / 4 / 5 measured in this process after the timed region of the headline config: `steps` serial
    steps each (inputs resident in HBM, HIP-event kernel time of the one launch per step), the roofline fraction priced like
    the headline's, the traffic ratio from the committed PMC summary of the same workload."""
    out = {}
    for cfg in (3, 4, 5, "5_notebook_tf10"):
        c = CONFIGS[5 if cfg == "5_notebook_tf10" else cfg]
        t0 = time.perf_counter()
        if cfg == "5_notebook_tf10":
            # "the notebook's own problem (tf = 10) reported separately" -- the config-5 generator at the
            # notebook's horizon, problem 0 = examples/astrobeeSE3manifold.ipynb cell 1 itself
            model = g.ASTROBEE_SE3_MANIFOLD
            (boxes, spheres), (x0, glo, ghi, tf) = P.iss_corner_env(True), P.astrobee_manifold_batch_tf10(c["B"])
        else:
            model, boxes, spheres, (x0, glo, ghi, tf) = workload(P, g, cfg, c["B"], 0)
        n, m = g.MODEL_DIMS[model]
        s = g.BatchSolver(model, c["N"], c["B"], hist_cap=MAX_ITER + 34, device=dev_ord, boxes=boxes, spheres=spheres)
        dv = [torch.from_numpy(a).to(torch.device("cuda", dev_ord)) for a in (x0, glo, ghi, tf)]
        torch.cuda.synchronize()
        ms, wall = [], []
        for i in range(warmup + steps):
            t1 = time.perf_counter()
            s.set_problems_dev(c["B"], *[d.data_ptr() for d in dv])
            s.solve_async(MAX_ITER)
            s.wait()
            torch.cuda.synchronize()
            if i >= warmup:
                wall.append(time.perf_counter() - t1)
                ms.append(s.last_solve_ms())
        st = s.status()
        s.close()
        b_kkt, b_lin = algorithmic_bytes(n, m, c["N"])
        kkt, scp, conv = int(st["ipm_iters"].sum()), int(st["iterations"].sum()), int(st["converged"].sum())
        alg = b_kkt * kkt + b_lin * scp
        avg = float(np.mean(ms))
        pmc, src = committed_pmc(cfg) if cfg != "5_notebook_tf10" else (None, None)
        ratio = (pmc or {}).get("traffic_over_algorithmic")
        if live and cfg != "5_notebook_tf10":      # this run's own counter passes (two child rocprofv3 --pmc runs of the same seeded batch)
            lp = live_pmc(cfg)
            if lp:
                ratio, src = lp["traffic_bytes_per_launch"] / float(alg), "live: two child rocprofv3 --pmc passes in this run"
        out[str(cfg)] = {
            "workload": c["name"] if cfg != "5_notebook_tf10" else
                        "astrobeeSE3manifold batch=2048, N=50 at the NOTEBOOK's horizon tf = 10, problem 0 = examples/astrobeeSE3manifold.ipynb cell 1 "
                        "(SURVEY.md 8(d): reported separately from BASELINE.json configs[4], which runs tf = 40)",
            "steps": steps, "ms_per_step": 1e3 * float(np.mean(wall)), "avg_launch_ms": avg,
            "value": conv / float(np.mean(wall)), "unit": "converged trajectories/s", "converged": conv, "problems": c["B"],
            "kkt_solves_per_launch": kkt, "max_scp_iters": int(st["iterations"].max()), "max_kkt_solves_of_a_problem": int(st["ipm_iters"].max()),
            "frac": alg / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic_over_algorithmic": ratio, "traffic_source": src, "traffic_kind": TRAFFIC_KIND,
            "setup_s": time.perf_counter() - t0 - float(np.sum(wall)),
        }
    # ... and the second SCP algorithm behind the same seam: TrajOpt on the freeflyerSE2 batch of `bench.py --algo trajopt`
    # (1024 problems through the whole penalty / convex / trust-region schedule; value = problems per second)
    try:
        t0 = time.perf_counter()
        c, B = CONFIGS[2], 1024
        model, boxes, spheres, (x0, glo, ghi, tf) = workload(P, g, 2, B, 0)
        n, m = g.MODEL_DIMS[model]
        tp = g.default_trajopt_params(model)
        cap = tp.max_penalty_iteration * tp.max_convex_iteration * tp.max_trust_iteration
        s = g.TrajOptSolver(model, c["N"], B, hist_cap=2 * cap + 16, device=dev_ord, boxes=boxes, spheres=spheres)
        ms, wall = [], []
        for i in range(warmup + steps):
            t1 = time.perf_counter()
            s.set_problems(x0, glo, ghi, tf)
            s.solve(cap)
            if i >= warmup:
                wall.append(time.perf_counter() - t1)
                ms.append(s.last_solve_ms())
        st = s.status()
        s.close()
        b_kkt, b_lin = algorithmic_bytes(n, m + n, c["N"])     # (the n defect variables of a knot are controls)
        kkt, scp = int(st["ipm_iters"].sum()), int(st["iterations"].sum())
        avg = float(np.mean(ms))
        pmc, src = committed_pmc("trajopt")
        out["trajopt"] = {
            "workload": f"TrajOpt (gusto_solve_trajopt), freeflyerSE2 batch={B}, N={c['N']}, fp64", "steps": steps,
            "ms_per_step_host_to_host": 1e3 * float(np.mean(wall)), "avg_launch_ms": avg, "value": B / (avg * 1e-3),
            "unit": "problems/s through the whole schedule (kernel time)", "problems": B, "kkt_solves_per_launch": kkt, "subproblems_per_launch": scp,
            "frac": (b_kkt * kkt + b_lin * scp) / (avg * 1e-3) / 1e9 / HBM_PEAK_GBS,
            "traffic_over_algorithmic": (pmc or {}).get("traffic_over_algorithmic"), "traffic_source": src,
            "setup_s": time.perf_counter() - t0 - float(np.sum(wall)),
        }
    except Exception as e:
        out["trajopt"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    # ... and the shards ONE GPU holds of the 8-GPU configurations (BASELINE.json configs[3] / [4]: 8192 / 8 and 2048 / 8 problems),
    # which leave SIMDs without a wave: kernel time with one, two and four waves per problem (gusto_set_decomposition,
    # csrc/segw.hpp) and as the library chooses
    try:
        for cfg, B in ((4, 1024), (5, 256)):
            c = CONFIGS[cfg]
            model, boxes, spheres, (x0, glo, ghi, tf) = workload(P, g, cfg, B, 0)
            dv = [torch.from_numpy(a).to(torch.device("cuda", dev_ord)) for a in (x0, glo, ghi, tf)]
            e = {"workload": f"{c['name'].split(' batch=')[0]} batch={B} (one GPU's shard of the 8-GPU configuration), N={c['N']}", "problems": B,
                 "unit": "ms, HIP-event kernel time of one launch"}
            for name, dec in (("one_wave", 1), ("two_waves", 3), ("four_waves", 4), ("auto", 0)):
                s = g.BatchSolver(model, c["N"], B, hist_cap=MAX_ITER + 34, device=dev_ord, boxes=boxes, spheres=spheres)
                s.set_decomposition(dec)
                ms = []
                for i in range(warmup + steps):
                    s.set_problems_dev(B, *[d.data_ptr() for d in dv])
                    s.solve_async(MAX_ITER)
                    s.wait()
                    if i >= warmup:
                        ms.append(s.last_solve_ms())
                st = s.status()
                e[name + "_ms"] = float(np.mean(ms))
                if dec == 0:
                    e["converged"] = int(st["converged"].sum())
                    e["value"] = e["converged"] / (e["auto_ms"] * 1e-3)
                    e["value_unit"] = "converged trajectories/s (kernel time, auto)"
                    e["workgroup_lds_bytes_auto"] = int(s.launch_info()[1])
                s.close()
            out[f"{cfg}_shard"] = e
    except Exception as e:
        out["shards"] = {"error": f"{type(e).__name__}: {e}"[:200]}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=0, help="0 = about 10 s of GPU time for the config (160 for config 2)")
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", type=int, default=2, choices=sorted(CONFIGS), help="BASELINE.json configs line (1-based)")
    ap.add_argument("--scaling", default="weak", choices=("weak", "strong"),
                    help="weak: every rank its own batch; strong: the config's batch split over the ranks")
    ap.add_argument("--batch", type=int, default=0, help="override the config's batch size")
    ap.add_argument("--cpu-sample", type=int, default=0, help="0 = the config's sample per usable host core (10-30 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-other-configs", action="store_true", help="skip the in-process lines of configs 3 / 4 / 5 (other_configs)")
    ap.add_argument("--no-live-traffic", action="store_true", help="skip the two child rocprofv3 --pmc passes (roofline.traffic then "
                    "comes from the committed summary under profiles/)")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive and overlapped extras (profiling runs: "
                    "every launch of the kernel is then one serial step, so rocprofv3's average is the step's)")
    ap.add_argument("--overlap", type=int, default=1,
                    help="batches in flight: consecutive steps alternate between this many handles/streams, so the "
                         "slowest problems of one batch overlap the start of the next (1 = strictly serial steps)")
    ap.add_argument("--algo", default="gusto", choices=("gusto", "trajopt"),
                    help="gusto: solve_gusto_hip! (the metric of BASELINE.json); trajopt: gusto_solve_trajopt, the second SCP "
                         "algorithm behind the same seam (configs 2 and 4: FreeflyerSE2 / AstrobeeSE3; default batch 1024 / 256)")
    ap.add_argument("--dist-backend", default="nccl", choices=("nccl", "gloo"),
                    help="torch.distributed backend of a multi-rank run (nccl = RCCL; gloo: ranks sharing one GPU in tests)")
    args = ap.parse_args()
    cfg = CONFIGS[args.config]
    trajopt = args.algo == "trajopt"
    if trajopt:
        if args.config not in (2, 4):
            raise SystemExit("--algo trajopt: FreeflyerSE2 (--config 2) and AstrobeeSE3 (--config 4) have a TrajOpt variant")
        args.overlap = 1
        args.batch = args.batch or {2: 1024, 4: 256}[args.config]
    if not args.steps:
        args.steps = ({2: 40, 4: 6} if trajopt else {2: 160, 3: 40, 4: 50, 5: 50})[args.config]

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    dev_ord = local_rank % torch.cuda.device_count()
    torch.cuda.set_device(dev_ord)
    dev = torch.device("cuda", dev_ord)
    dist = None
    if "WORLD_SIZE" in os.environ:      # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend="gloo")
    cdev = dev if (dist is None or args.dist_backend == "nccl") else torch.device("cpu")   # where the collectives run

    import gusto_jl_amd as g
    P = g.problems
    N_KNOTS = cfg["N"]
    Btot = args.batch or cfg["B"]
    if args.scaling == "weak":      # rank r solves problems [r*B, (r+1)*B): independent shards, no exchange step
        first, B = rank * Btot, Btot
    else:                           # the batch of the config in contiguous blocks over the ranks
        first, hi = g.host.shard_bounds(Btot, world, rank)
        B = hi - first
    if B <= 0:
        raise SystemExit(f"rank {rank}: empty shard (batch {Btot} over {world} ranks)")
    model, boxes, spheres, (x0, glo, ghi, tf) = workload(P, g, args.config, B, first)
    n, m = g.MODEL_DIMS[model]
    D = max(1, args.overlap)
    if trajopt:
        tp = g.default_trajopt_params(model)
        TO_MAX = tp.max_penalty_iteration * tp.max_convex_iteration * tp.max_trust_iteration
        mk = lambda: g.TrajOptSolver(model, N_KNOTS, B, hist_cap=2 * TO_MAX + 16, device=dev_ord, boxes=boxes, spheres=spheres)
    else:
        mk = lambda: g.BatchSolver(model, N_KNOTS, B, hist_cap=MAX_ITER + 34, device=dev_ord, boxes=boxes, spheres=spheres)
    solvers = [mk() for _ in range(D)]
    solver = solvers[0]
    # inputs resident in HBM before the timed region
    d_x0, d_glo, d_ghi, d_tf = (torch.from_numpy(a).to(dev) for a in (x0, glo, ghi, tf))
    torch.cuda.synchronize()

    kernel_ms = []
    timed_in_flight = [False] * D
    gathered, gather_ok, gather_err, gather_s = [0], [True], [None], [0.0]

    def collect(j, had_step):
        solvers[j].wait()
        if dist is not None and had_step:   # final gather of this step's trajectories to rank 0, device tensors -> RCCL
            tg = time.perf_counter()
            # every rank takes the same branch: the ranks agree (MIN over ranks) on whether the gather is still on
            flag = torch.tensor([1 if gather_ok[0] else 0], dtype=torch.int32, device=cdev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()) == 0:
                gather_ok[0] = False
            else:
                try:
                    Xd, Ud = solvers[j].traj_dev()
                    if cdev.type == "cpu":
                        Xd, Ud = Xd.cpu().numpy(), Ud.cpu().numpy()
                    out = g.host.gather_batch_results(dict(X=Xd, U=Ud), world, rank)
                    if rank == 0:
                        gathered[0] = int(out["X"].shape[0])
                except Exception as e:      # say so in the JSON line; the other ranks learn it at the next step's all_reduce
                    gather_ok[0] = False
                    gather_err[0] = f"{type(e).__name__}: {e}"[:200]
                    print(f"[bench] rank {rank}: final gather failed, continuing without it: {gather_err[0]}", file=sys.stderr)
            if timed_in_flight[j]:
                gather_s[0] += time.perf_counter() - tg
        if timed_in_flight[j]:
            kernel_ms.append(solvers[j].last_solve_ms())
            timed_in_flight[j] = False

    in_flight = [False] * D

    def step(i, timed):
        # one step = straight-line initialisation of the batch + the whole GuSTO solve of every problem in it.
        # Step i runs on handle i % D; re-using a handle first completes the step it still has in flight.
        j = i % D
        collect(j, in_flight[j])
        solvers[j].set_problems_dev(B, d_x0.data_ptr(), d_glo.data_ptr(), d_ghi.data_ptr(), d_tf.data_ptr())
        if trajopt:
            solvers[j].solve(TO_MAX)        # (gusto_solve_trajopt is synchronous)
        else:
            solvers[j].solve_async(MAX_ITER)
        timed_in_flight[j] = timed
        in_flight[j] = True

    def drain():
        for j in range(D):
            collect(j, in_flight[j])
            in_flight[j] = False

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for i in range(args.warmup):
        step(i, False)
    drain()
    barrier()
    t0 = time.perf_counter()
    for i in range(args.steps):
        step(i, True)
    drain()
    barrier()
    elapsed = time.perf_counter() - t0
    assert len(kernel_ms) == args.steps
    solver = solvers[(args.steps - 1) % D]

    st = solver.status()
    n_conv, n_succ = int(st["converged"].sum()), int(st["successful"].sum())
    scp_iters, ipm_iters = int(st["iterations"].sum()), int(st["ipm_iters"].sum())
    tot = torch.tensor([float(n_conv), float(n_succ), float(scp_iters), float(ipm_iters), float(B)], device=cdev,
                       dtype=torch.float64)
    tmax = torch.tensor([elapsed, gather_s[0]], device=cdev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tot = tot.cpu().numpy()
    elapsed, gather_total = float(tmax[0].item()), float(tmax[1].item())

    # named extras, outside the timed region, rank 0's GPU only:
    # (a) SURVEY.md 8(d): host buffers in, trajectories out (H2D + solve + D2H), median of 5
    pcie = []
    for _ in range(0 if args.no_extras else 5):
        t1 = time.perf_counter()
        solver.set_problems(x0, glo, ghi, tf)
        solver.solve(TO_MAX if trajopt else MAX_ITER)
        solver.traj()
        pcie.append(time.perf_counter() - t1)
    pcie_s = float(np.median(pcie)) if pcie else float("nan")
    # (b) two batches in flight on two handles/streams (the tail of one batch overlaps the head of the next)
    overlapped = None
    if D == 1 and dist is None and not args.no_extras and not trajopt:
        s2 = [solver, mk()]
        K2 = 12
        for i in range(2):
            s2[i].set_problems_dev(B, d_x0.data_ptr(), d_glo.data_ptr(), d_ghi.data_ptr(), d_tf.data_ptr())
            s2[i].solve_async(MAX_ITER)
        for q in s2:
            q.wait()
        torch.cuda.synchronize()
        t1 = time.perf_counter()
        for i in range(K2):
            q = s2[i % 2]
            q.wait()
            q.set_problems_dev(B, d_x0.data_ptr(), d_glo.data_ptr(), d_ghi.data_ptr(), d_tf.data_ptr())
            q.solve_async(MAX_ITER)
        for q in s2:
            q.wait()
        torch.cuda.synchronize()
        overlapped = n_conv * K2 / (time.perf_counter() - t1)

    if rank == 0:
        # TrajOpt marks a problem `converged` only when evaluate_ctol < ctol (scp_trajopt.jl:150-154), which the default
        # SCPParam_TrajOpt rarely reaches before its loops end: the rate of that line is problems run through the whole
        # three-loop schedule per second, the converged count is reported beside it
        value = (tot[4] if trajopt else tot[0]) * args.steps / elapsed
        problems = int(tot[4])
        avg_ms = float(np.mean(kernel_ms))
        b_kkt, b_lin = algorithmic_bytes(n, m + (n if trajopt else 0), N_KNOTS)   # (TrajOpt: the n defect variables of a knot are controls)
        alg_bytes = b_kkt * ipm_iters + b_lin * scp_iters      # this rank, one launch
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        # HBM traffic of one solve.  PMC counters need their own rocprofv3 --pmc passes: the default single-GPU run makes
        # them itself after the timed region (two child passes of tools/pmc_probe.py on the same seeded batch: live_pmc);
        # otherwise, or if that fails, the figure is read from the latest committed summary of the same workload
        traffic, traffic_src, valu_flops, traffic_live = None, None, None, None
        pmc, pmc_src = committed_pmc(args.config)
        full_batch = not trajopt and B == CONFIGS[args.config]["B"]
        if pmc and full_batch:
            traffic = pmc.get("traffic_bytes_per_launch")
            traffic_src = pmc_src + " (separate rocprofv3 --pmc passes, not this run)"
            valu_flops = valu_fp64(pmc, ipm_iters)
        if trajopt and args.config == 2 and B == 1024:     # (tools/pmc_probe.py's TrajOpt batch)
            pmc_t, src_t = committed_pmc("trajopt")
            if pmc_t:
                traffic, traffic_src = pmc_t.get("traffic_bytes_per_launch"), src_t + " (separate rocprofv3 --pmc passes, not this run)"
        if (full_batch or (trajopt and args.config == 2 and B == 1024)) and dist is None and not args.no_extras and not args.no_live_traffic:
            traffic_live = live_pmc(args.config, algo="trajopt" if trajopt else "gusto")
            if traffic_live:
                traffic_live["committed_traffic_bytes_per_launch"] = traffic
                traffic = traffic_live["traffic_bytes_per_launch"]
                traffic_src = ("live: two child rocprofv3 --pmc passes (FETCH_SIZE x 2, WRITE_SIZE; KiB) of tools/pmc_probe.py in this run, "
                               "one launch of the same seeded batch")
        others = None
        if args.config == 2 and not args.batch and dist is None and not args.no_extras and not trajopt and not args.no_other_configs:
            try:
                others = other_configs(P, g, torch, dev_ord, live=not args.no_live_traffic)
            except Exception as e:      # the headline line must come out whatever happens to the side measurements
                others = {"error": f"{type(e).__name__}: {e}"[:200]}
        out = {
            "metric": (f"problems/sec through the whole TrajOpt schedule (batched SCP, solve_trajopt_hip!), {cfg['model'].lower()} N={N_KNOTS}, inputs resident in HBM"
                       if trajopt else f"converged trajectories/sec (batched SCP), {cfg['model'].lower()} N={N_KNOTS}, inputs resident in HBM"),
            # the run contract: `value` = whole-job rate with the inputs resident in HBM when the timed region starts; the
            # PCIe-inclusive median-of-5 rate of SURVEY.md 8(d) is `pcie_inclusive_traj_per_s` below
            "value_definition": ("problems" if trajopt else "converged problems") + " x steps / wall time of the timed steps, inputs resident in HBM (serial steps)",
            "value": value, "unit": "problems/s" if trajopt else "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": args.scaling,
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": (cfg["name"] if not args.batch else cfg["name"].replace(f"batch={cfg['B']}", f"batch={args.batch} (--batch)")) +
                                   (" per GPU" if args.scaling == "weak" else " split over the ranks"),
                       "baseline_config": args.config, "batch_total": problems, "batch_rank0": B, "N": N_KNOTS,
                       "algorithm": "TrajOpt (gusto_solve_trajopt)" if trajopt else "GuSTO (gusto_solve)",
                       "max_iter": TO_MAX if trajopt else MAX_ITER, "sharding": "independent problems per rank; final gather of X,U to rank 0 "
                                                         "over RCCL inside every step (N > 1)",
                       "batches_in_flight": D},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src, "traffic_kind": TRAFFIC_KIND,
                         # (how much of it can be Infinity-Cache hits: the launch's interior point workspace against the 256 MiB)
                         "workspace_bytes": solver.workspace_bytes(), "infinity_cache_bytes": INFINITY_CACHE_BYTES,
                         "lds": lds_level(pmc, ipm_iters, avg_ms) if full_batch else None,
                         "traffic_over_algorithmic": (traffic / alg_bytes) if traffic else None, "traffic_live": traffic_live,
                         # the issue-side bound next to the memory-side one (the kernels are register / LDS resident): fp64
                         # VALU (+ matrix core) flops of the launch from the committed SQ counters / this run's kernel time
                         "valu_fp64_tflops": (valu_flops / (avg_ms * 1e-3) / 1e12) if valu_flops else None,
                         "valu_fp64_frac": (valu_flops / (avg_ms * 1e-3) / 1e12 / FP64_VALU_PEAK_TFLOPS) if valu_flops else None,
                         "valu_fp64_peak_tflops": FP64_VALU_PEAK_TFLOPS, "valu_fp64_source": (pmc_src + ": SQ_INSTS_VALU_{FMA,MUL,ADD}_F64 x 64 lanes, committed counters x this run's time") if valu_flops else None,
                         # one gusto_solve = ONE launch of the persistent kernel (device-side longest-first scheduler,
                         # gusto_set_schedule); avg_launch_ms from HIP events on the handle's stream
                         "kernel": f"gusto::trajopt_kernel<{4 if args.config == 2 else 5}>" if trajopt else f"gusto::scp_kernel<{model}>", "launches_per_solve": 1, "avg_launch_ms": avg_ms,
                         "kkt_solves_per_launch": ipm_iters, "scp_iters_per_launch": scp_iters,
                         "bytes_per_kkt_solve": b_kkt, "bytes_per_linearisation": b_lin,
                         # with D > 1 launches overlap, so a launch's duration spans the batches it shares the GPU
                         # with; the job-level rate is the algorithmic bytes of all launches over the timed region
                         "launches_in_flight": D,
                         "aggregate_achieved": alg_bytes * args.steps / elapsed / 1e9},
            "converged": int(tot[0]), "successful": int(tot[1]), "problems": problems,
            "yield": float(tot[0]) / max(1, problems),
            "mean_scp_iters": tot[2] / max(1, problems), "mean_ipm_iters": tot[3] / max(1, problems),
            "max_scp_iters": int(st["iterations"].max()), "max_kkt_solves_of_a_problem": int(st["ipm_iters"].max()),
            "pcie_inclusive_traj_per_s": (n_conv / pcie_s) if pcie else None, "pcie_inclusive_note": "SURVEY.md 8(d): set_problems (host) + "
            "solve + get_traj (host), median of 5, one GPU", "overlapped_traj_per_s": overlapped,
            "gathered_problems_per_step": gathered[0] if dist is not None else None, "gather_error": gather_err[0],
            "gather_ms_per_step": (1e3 * gather_total / args.steps) if dist is not None else None,
            # BASELINE.json configs 3 / 4 / 5 in the same run (3 serial steps each, after the timed region)
            "other_configs": others,
        }
        if not args.no_cpu_baseline:
            threads = usable_cores()
            n_sample = args.cpu_sample or (({2: 768, 4: 16}[args.config]) if trajopt else cfg["cpu_per_core"] * max(2, threads))
            out["cpu_baseline"] = cpu_baseline(P, g, args.config, n_sample, threads, B, args.algo)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
