"""Timing of the batched indirect shooting (gusto_shoot) at the BASELINE batch sizes of the two models that have a shooting
ODE: python tools/shoot_bench.py   (one JSON line per model; run under rocprofv3 --kernel-trace --stats for the kernel time)"""
import json, os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems


def run(name, model, N, B, batch, boxes=None, spheres=None, scp_iters=3):
    s = g.BatchSolver(model, N, B, hist_cap=64, boxes=boxes, spheres=spheres)
    s.set_problems(*batch)
    s.solve(scp_iters)                       # the SCP dual that seeds Newton (solve_SCPshooting!: a few SCP iterations first)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter()
        r = s.shoot()
        ts.append(time.perf_counter() - t0)
    ms = 1e3 * min(ts)
    opt = int((r["status"] == 1).sum())
    print(json.dumps({"shooting": name, "B": B, "N": N, "scp_iters_before": scp_iters, "gusto_shoot_ms": ms, "optimal": opt,
                      "optimal_per_s": opt / (ms * 1e-3), "mean_newton_iters": float(r["newton_iters"].mean()),
                      "max_newton_iters": int(r["newton_iters"].max())}), flush=True)


run("dubins_car", g.DUBINS_CAR, 30, 65536, P.dubins_batch(65536))
bx, sp = P.iss_corner_env(True)
run("astrobeeSE3manifold", g.ASTROBEE_SE3_MANIFOLD, 50, 2048, P.astrobee_manifold_batch(2048), boxes=bx, spheres=sp)
