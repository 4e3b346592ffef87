"""All five BASELINE.json configs, one gusto_solve each (after one warm-up), as JSON lines: python tools/bench_configs.py
(bench.py is the contract benchmark of configs[1]; this is the side table of DESIGN.md section 5)."""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems


def run(name, model, N, B, batch, boxes=None, spheres=None, max_iter=30):
    x0, glo, ghi, tf = batch
    s = g.BatchSolver(model, N, B, hist_cap=max_iter + 34, boxes=boxes, spheres=spheres)
    for _ in range(2):
        s.set_problems(x0, glo, ghi, tf)
        s.solve(max_iter)
    st, ms = s.status(), s.last_solve_ms()
    out = {"config": name, "B": B, "N": N, "kernel_ms": ms, "converged": int(st["converged"].sum()),
           "successful": int(st["successful"].sum()), "converged_traj_per_s": float(st["converged"].sum() / (ms * 1e-3)),
           "mean_scp_iters": float(st["iterations"].mean()), "kkt_solves": int(st["ipm_iters"].sum()),
           "us_per_kkt_solve": float(1e3 * ms / max(1, st["ipm_iters"].sum()))}
    print(json.dumps(out), flush=True)


only = set(int(a) for a in sys.argv[1:])      # optional: model ids to run (default all five configs)
env = P.freeflyer_env()
one = (P.FREEFLYER_X_INIT[None], P.FREEFLYER_X_GOAL[None], P.FREEFLYER_X_GOAL[None], np.array([P.FREEFLYER_TF]))
if not only or 0 in only:
    run("1: freeflyerSE2 notebook problem", g.FREEFLYER_SE2, 50, 1, one, boxes=env)
    run("2: freeflyerSE2 random initial states", g.FREEFLYER_SE2, 50, 4096, P.freeflyer_batch(4096), boxes=env)
if not only or 1 in only:
    run("3: dubins_car", g.DUBINS_CAR, 30, 65536, P.dubins_batch(65536))
bx, sp = P.iss_corner_env(True)
if not only or 2 in only:
    run("4: astrobeeSE3 ISS corner", g.ASTROBEE_SE3, 50, 8192, P.astrobee_se3_batch(8192), boxes=bx, spheres=sp)
if not only or 3 in only:
    run("5: astrobeeSE3manifold ISS corner (tf=40)", g.ASTROBEE_SE3_MANIFOLD, 50, 2048, P.astrobee_manifold_batch(2048), boxes=bx, spheres=sp)
