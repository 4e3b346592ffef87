#!/usr/bin/env python3
"""bench.py -- converged trajectories/sec of the batched GuSTO SCP hot path (BASELINE.json metric).

  python bench.py --gpus N --steps K --warmup W
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N --steps K --warmup W

One "step" = one full pass of the hot path over one batch: straight-line initialisation of every problem,
then gusto_solve (the whole GuSTO loop: linearisation, convex subproblems, trust-region/penalty updates) until
every problem of the batch has stopped; with N > 1 ranks the step ends with the final gather of every rank's
trajectories to rank 0 over RCCL, straight from HBM (north_star: "RCCL only for the batch split and final gather").
Workload = BASELINE.json configs[1]: freeflyerSE2, batch 4096 random initial states, N = 50, fp64, per GPU (weak
scaling: every rank solves its own 4096 problems; no data-path collective -- the problems are independent).
Inputs are resident in HBM before the timed region.  Rank 0 prints ONE JSON line.

`value` = strictly serial steps (one batch in flight), so that the kernel time behind `roofline` is that of a lone
solve.  Named extras in the same line: `overlapped_traj_per_s` (consecutive batches on two handles/streams, the
steady-state serving rate) and `pcie_inclusive_traj_per_s` (gusto_set_problems from host buffers +
gusto_solve + gusto_get_traj to host, median of 5).
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

N_KNOTS = 50
BATCH = 4096
MAX_ITER = 30
# algorithmic bytes (restated in DESIGN.md): one KKT solve of freeflyerSE2 N=50 streams
# 8*N*(nz(nz+1)/2 + n*nz + 2(nz+n)) bytes; one linearisation 8*N*(2(n+m) + n + n^2) bytes
BYTES_PER_KKT = 8 * N_KNOTS * (9 * 10 // 2 + 6 * 9 + 2 * (9 + 6))       # 51 600
BYTES_PER_LINEARIZE = 8 * N_KNOTS * (2 * 9 + 6 + 36)                     # 24 000
HBM_PEAK_GBS = 8000.0


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


def cpu_baseline(problems, env, n_sample, threads):
    """The oracle (a C port, NOT the Julia reference) on a bounded sample of the same workload."""
    sys.path.insert(0, os.path.join(ROOT, "oracle"))
    import gusto_oracle as go
    x0, glo, ghi, tf = problems.freeflyer_batch(n_sample)
    go.lib()
    t0 = time.perf_counter()
    r = go.solve_batch(go.FREEFLYER_SE2, N_KNOTS, env, None, x0, glo, ghi, tf, MAX_ITER, threads)
    dt = time.perf_counter() - t0
    return {"value": float(r["converged"].sum() / dt), "unit": "converged trajectories/s", "cores": threads,
            "kind": "port", "sample": f"first {n_sample} problems of the batch, oracle/libgusto_oracle.so "
            f"(OpenMP over problems), {dt:.1f} s wall"}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=160)     # ~10 s of GPU time in the timed region
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=BATCH)
    ap.add_argument("--cpu-sample", type=int, default=0, help="0 = 2048 problems per usable host core (about 15-20 s)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true", help="skip the PCIe-inclusive and overlapped extras (profiling runs: "
                    "every launch of the kernel is then one serial step, so rocprofv3's average is the step's)")
    ap.add_argument("--overlap", type=int, default=1,
                    help="batches in flight: consecutive steps alternate between this many handles/streams, so the "
                         "slowest problems of one batch overlap the start of the next (1 = strictly serial steps)")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    import torch
    if not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU: the HIP path has no CPU fallback")
    torch.cuda.set_device(local_rank)
    dist = None
    if "WORLD_SIZE" in os.environ:      # launched by torch.distributed.run: one rank per GPU over RCCL
        import torch.distributed as dist
        dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))

    import gusto_jl_amd as g
    P = g.problems
    env = P.freeflyer_env()
    B = args.batch
    # rank r solves problems [r*B, (r+1)*B): independent shards, no exchange step
    x0, glo, ghi, tf = P.freeflyer_batch(B, first=rank * B)
    D = max(1, args.overlap)
    solvers = [g.BatchSolver(g.FREEFLYER_SE2, N_KNOTS, B, hist_cap=MAX_ITER + 34, device=local_rank, boxes=env)
               for _ in range(D)]
    solver = solvers[0]
    # inputs resident in HBM before the timed region
    dev = torch.device("cuda", local_rank)
    d_x0, d_glo, d_ghi, d_tf = (torch.from_numpy(a).to(dev) for a in (x0, glo, ghi, tf))
    torch.cuda.synchronize()

    import ctypes as C

    kernel_ms = []
    timed_in_flight = [False] * D

    gathered, gather_ok, gather_err = [0], [True], [None]

    def collect(j):
        solvers[j].wait()
        if dist is not None and gather_ok[0]:   # final gather of this step's trajectories to rank 0, device tensors -> RCCL
            try:
                Xd, Ud = solvers[j].traj_dev()
                out = g.host.gather_batch_results(dict(X=Xd, U=Ud), world, rank)
                if rank == 0:
                    gathered[0] = int(out["X"].shape[0])
            except Exception as e:      # never lose the scaling measurement to the gather: say so in the JSON line instead
                gather_ok[0] = False
                gather_err[0] = f"{type(e).__name__}: {e}"[:200]
                print(f"[bench] rank {rank}: final gather failed, continuing without it: {gather_err[0]}", file=sys.stderr)
        if timed_in_flight[j]:
            kernel_ms.append(solvers[j].last_solve_ms())
            timed_in_flight[j] = False

    def step(i, timed):
        # one step = straight-line initialisation of the batch + the whole GuSTO solve of every problem in it.
        # Step i runs on handle i % D; re-using a handle first completes the step it still has in flight.
        j = i % D
        collect(j)
        solvers[j].set_problems_dev(B, d_x0.data_ptr(), d_glo.data_ptr(), d_ghi.data_ptr(), d_tf.data_ptr())
        solvers[j].solve_async(MAX_ITER)
        timed_in_flight[j] = timed

    def drain():
        for j in range(D):
            collect(j)

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
    tot = torch.tensor([float(n_conv), float(n_succ), float(scp_iters), float(ipm_iters)], device=dev,
                       dtype=torch.float64)
    tmax = torch.tensor([elapsed], device=dev, dtype=torch.float64)
    if dist is not None:
        dist.all_reduce(tot, op=dist.ReduceOp.SUM)
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
    tot = tot.cpu().numpy()
    elapsed = float(tmax.item())

    # named extras, outside the timed region, rank 0's GPU only:
    # (a) SURVEY.md 8(d): host buffers in, trajectories out (H2D + solve + D2H), median of 5
    pcie = []
    for _ in range(0 if args.no_extras else 5):
        t1 = time.perf_counter()
        solver.set_problems(x0, glo, ghi, tf)
        solver.solve(MAX_ITER)
        solver.traj()
        pcie.append(time.perf_counter() - t1)
    pcie_s = float(np.median(pcie)) if pcie else float("nan")
    # (b) two batches in flight on two handles/streams (the tail of one batch overlaps the head of the next)
    overlapped = None
    if D == 1 and dist is None and not args.no_extras:
        s2 = [solver, g.BatchSolver(g.FREEFLYER_SE2, N_KNOTS, B, hist_cap=MAX_ITER + 34, device=local_rank, boxes=env)]
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
        value = tot[0] * args.steps / elapsed
        avg_ms = float(np.mean(kernel_ms))
        alg_bytes = BYTES_PER_KKT * ipm_iters + BYTES_PER_LINEARIZE * scp_iters      # this rank, one launch
        achieved = alg_bytes / (avg_ms * 1e-3) / 1e9
        # HBM traffic of one solve: NOT measured in this run -- PMC counters need their own rocprofv3 --pmc passes
        # (tools/profile_round.sh); the figure is read from the latest committed summary of the same workload
        traffic, traffic_src = None, None
        try:
            import glob
            pmc = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc.json")))
            if pmc and B == BATCH:
                traffic = json.load(open(pmc[-1]))["traffic_bytes_per_launch"]
                traffic_src = "profiles/" + os.path.basename(pmc[-1]) + " (separate rocprofv3 --pmc passes, not this run)"
        except Exception:
            traffic = None
        out = {
            "metric": "converged trajectories/sec (batched SCP), freeflyerSE2 N=50",
            "value": value, "unit": "trajectories/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": 1e3 * elapsed / args.steps, "higher_is_better": True, "scaling": "weak",
            "vs_baseline": None, "dtype": "f64", "data": "synthetic",
            "config": {"workload": "freeflyerSE2 batch=4096 random initial states per GPU, N=50, fp64 "
                                   "(BASELINE.json configs[1])", "batch_per_gpu": B, "N": N_KNOTS,
                       "max_iter": MAX_ITER, "sharding": "independent problems per rank; final gather of X,U to rank 0 "
                                                         "over RCCL inside every step (N > 1)",
                       "batches_in_flight": D},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": achieved / HBM_PEAK_GBS, "traffic": traffic, "traffic_source": traffic_src,
                         # one gusto_solve = ONE launch of the persistent kernel (device-side longest-first scheduler,
                         # gusto_set_schedule); avg_launch_ms from HIP events on the handle's stream
                         "kernel": "gusto::scp_kernel<0>", "launches_per_solve": 1, "avg_launch_ms": avg_ms,
                         "kkt_solves_per_launch": ipm_iters, "scp_iters_per_launch": scp_iters,
                         # with D > 1 launches overlap, so a launch's duration spans the batches it shares the GPU
                         # with; the job-level rate is the algorithmic bytes of all launches over the timed region
                         "launches_in_flight": D,
                         "aggregate_achieved": alg_bytes * args.steps / elapsed / 1e9},
            "converged": int(tot[0]), "successful": int(tot[1]), "problems": B * world,
            "mean_scp_iters": tot[2] / (B * world), "mean_ipm_iters": tot[3] / (B * world),
            "pcie_inclusive_traj_per_s": (n_conv / pcie_s) if pcie else None, "pcie_inclusive_note": "SURVEY.md 8(d): set_problems (host) + "
            "solve + get_traj (host), median of 5, one GPU", "overlapped_traj_per_s": overlapped,
            "gathered_problems_per_step": gathered[0] if dist is not None else None, "gather_error": gather_err[0],
        }
        if not args.no_cpu_baseline:
            threads = usable_cores()
            n_sample = args.cpu_sample or max(4096, 2048 * threads)
            out["cpu_baseline"] = cpu_baseline(P, env, n_sample, threads)
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
