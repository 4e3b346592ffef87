"""Multi-process path on CPU (gloo, world_size 2): the batch is sharded into contiguous blocks, every rank works
on its own block with NO data-path collective, and the only communication is the final gather to rank 0
(SURVEY.md 8(e)).  The per-rank 'solve' here is a deterministic stand-in computed from the inputs, so the test
checks exactly the sharding / gather plumbing that bench.py and solve_SCP_batch use on GPUs."""
import os
import socket
import sys

import numpy as np
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import gusto_jl_amd as g
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    x0, glo, ghi, tf = g.problems.freeflyer_batch(B)           # every rank can regenerate the seeded inputs
    lo, hi = g.host.shard_bounds(B, world, rank)
    # stand-in for the per-rank solve: something only the owner of a problem can produce
    local = {"X": (x0[lo:hi] * (1 + np.arange(lo, hi))[:, None]), "owner": np.full(hi - lo, rank, dtype=np.int64),
             "index": np.arange(lo, hi, dtype=np.int64)}
    out = g.host.gather_batch_results(local, world, rank)
    import torch
    cnt = torch.tensor([hi - lo], dtype=torch.int64)
    dist.all_reduce(cnt)                                       # the metric reduction bench.py performs
    if rank == 0:
        q.put((out["X"], out["owner"], out["index"], int(cnt.item())))
    else:
        assert out is None
    dist.destroy_process_group()


def test_two_rank_shard_and_gather():
    B, world = 37, 2                       # ragged: 19 + 18
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    X, owner, index, total = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    import gusto_jl_amd as g
    x0, _, _, _ = g.problems.freeflyer_batch(B)
    assert total == B
    np.testing.assert_array_equal(index, np.arange(B))
    np.testing.assert_array_equal(owner, np.array([0] * 19 + [1] * 18))
    np.testing.assert_array_equal(X, x0 * (1 + np.arange(B))[:, None])


def _gpu_worker(rank, world, port, B, q):
    sys.path.insert(0, ROOT)
    import torch.distributed as dist
    import gusto_jl_amd as g
    os.environ["MASTER_ADDR"], os.environ["MASTER_PORT"] = "127.0.0.1", str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    P = g.problems
    x0, glo, ghi, tf = P.freeflyer_batch(B)
    out = g.host.solve_batch_sharded(g.FREEFLYER_SE2, 50, x0, glo, ghi, tf, world, rank, device=0, boxes=P.freeflyer_env())
    if rank == 0:
        q.put({k: (v.cpu().numpy() if hasattr(v, "cpu") else v) for k, v in out.items()})
    else:
        assert out is None
    dist.destroy_process_group()


import pytest


@pytest.mark.gpu
def test_two_rank_real_solve_and_gather():
    """The real thing with world_size 2: each rank runs the HIP solver on its shard (both on GPU 0 of the one-GPU test
    box, gloo carrying the gather) and rank 0 receives exactly what one process solving the whole batch produces."""
    B, world = 75, 2                       # ragged: 38 + 37
    port = _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_gpu_worker, args=(r, world, port, B, q)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=600)
    for p in procs:
        p.join(timeout=120)
        assert p.exitcode == 0
    sys.path.insert(0, ROOT)
    import gusto_jl_amd as g
    P = g.problems
    x0, glo, ghi, tf = P.freeflyer_batch(B)
    s = g.BatchSolver(g.FREEFLYER_SE2, 50, B, hist_cap=136, boxes=P.freeflyer_env())
    s.set_problems(x0, glo, ghi, tf)
    s.solve(30)
    X, U = s.traj()
    st = s.status()
    np.testing.assert_array_equal(out["X"], X)          # per-problem results do not depend on the batch they ran in
    np.testing.assert_array_equal(out["U"], U)
    np.testing.assert_array_equal(out["iterations"], st["iterations"])
    np.testing.assert_array_equal(out["converged"].astype(bool), st["converged"])


def _rccl_worker(port, q):
    sys.path.insert(0, ROOT)
    os.environ["GUSTO_FORCE_GATHER"] = "1"        # a one-rank group still runs all_gather + gather
    import torch
    import torch.distributed as dist
    import gusto_jl_amd as g
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda", 0))
    P = g.problems
    x0, glo, ghi, tf = P.freeflyer_batch(24)
    s = g.BatchSolver(g.FREEFLYER_SE2, 50, 24, hist_cap=136, boxes=P.freeflyer_env())
    s.set_problems(x0, glo, ghi, tf)
    s.solve(30)
    Xd, Ud = s.traj_dev()
    out = g.host.gather_batch_results(dict(X=Xd, U=Ud, it=s.status()["iterations"].astype(np.int64)), 1, 0)
    X, U = s.traj()
    q.put((bool(out["X"].is_cuda), np.array_equal(out["X"].cpu().numpy(), X), np.array_equal(out["U"].cpu().numpy(), U),
           np.array_equal(out["it"], s.status()["iterations"])))
    dist.destroy_process_group()


@pytest.mark.gpu
def test_rccl_gathers_the_handles_own_device_buffers():
    """The nccl (= RCCL) backend sends straight from the handle's HBM buffers (zero-copy torch views of hipMalloc'd
    memory): all_gather of the shard sizes + gather of X, U in a one-rank group on the test box's single GPU."""
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    p = ctx.Process(target=_rccl_worker, args=(_free_port(), q))
    p.start()
    on_gpu, okx, oku, okit = q.get(timeout=600)
    p.join(timeout=120)
    assert p.exitcode == 0 and on_gpu and okx and oku and okit


def _run_bench(extra, nproc=0, timeout=900):
    import json
    import subprocess
    cmd = [sys.executable]
    if nproc:
        cmd += ["-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={nproc}", "--master-addr", "127.0.0.1",
                "--master-port", str(_free_port())]
    cmd += [os.path.join(ROOT, "bench.py")] + extra
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=timeout, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]
    return json.loads(lines[0])


@pytest.mark.gpu
def test_bench_runs_config_4_strong_scaling_on_two_ranks():
    """BASELINE.json configs[3] ("astrobeeSE3 ... batch-sharded") through bench.py itself: the batch is split over two
    ranks (both on the one GPU of the test box, gloo carrying the collectives), the final gather sits inside every step
    and rank 0 reports the whole job."""
    out = _run_bench(["--gpus", "2", "--config", "4", "--scaling", "strong", "--batch", "96", "--steps", "2", "--warmup", "1",
                      "--no-cpu-baseline", "--no-extras", "--dist-backend", "gloo"], nproc=2)
    assert out["scaling"] == "strong" and out["n_gpus"] == 2 and out["config"]["baseline_config"] == 4
    assert out["problems"] == 96 and out["config"]["batch_rank0"] == 48
    assert out["gathered_problems_per_step"] == 96 and out["gather_error"] is None and out["gather_ms_per_step"] > 0
    assert 0.5 < out["yield"] <= 1.0 and out["value"] > 0
    assert out["roofline"]["bytes_per_kkt_solve"] == 8 * 50 * (18 * 19 // 2 + 12 * 18 + 2 * (18 + 12))


@pytest.mark.gpu
@pytest.mark.parametrize("cfg,n,m,N", [(3, 3, 1, 30), (5, 13, 6, 50)])
def test_bench_runs_the_other_configs(cfg, n, m, N):
    out = _run_bench(["--config", str(cfg), "--batch", "128", "--steps", "2", "--warmup", "1", "--no-cpu-baseline", "--no-extras"])
    nz = n + m
    assert out["config"]["baseline_config"] == cfg and out["problems"] == 128 and out["scaling"] == "weak"
    assert out["roofline"]["bytes_per_kkt_solve"] == 8 * N * (nz * (nz + 1) // 2 + n * nz + 2 * (nz + n))
    assert 0 < out["yield"] <= 1.0 and out["roofline"]["frac"] > 0


def test_bench_config_table_matches_baseline_json():
    """CPU: the config table of bench.py names the batch sizes / horizons BASELINE.json states."""
    import json
    sys.path.insert(0, ROOT)
    import bench
    base = json.load(open(os.path.join(ROOT, "BASELINE.json")))["configs"]
    for i, c in bench.CONFIGS.items():
        line = base[i - 1]
        assert f"batch={c['B']}" in line.replace(" ", "").replace("batch=", "batch=") and f"N={c['N']}" in line, (i, line)
    assert bench.algorithmic_bytes(6, 3, 50) == (51600, 24000)
