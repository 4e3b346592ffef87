"""Edge-horizon robustness probe: python tools/n5_debug.py <model> -- one cold subproblem per problem at small N, statuses and iteration counts"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, gusto_jl_amd as g
P = g.problems
model = int(sys.argv[1])
B = 24
NS = [int(v) for v in sys.argv[2:]] or [5, 7, 12, 33, 50]
for N in NS:
    boxes = spheres = None
    if model == 0: batch = P.freeflyer_batch(B); boxes = P.freeflyer_env(); D, cl = 3.0, 0.05
    elif model == 2: batch = P.astrobee_se3_batch(B); boxes, spheres = P.iss_corner_env(True); D, cl = 10.0, 0.03
    else: batch = P.astrobee_manifold_batch(B); boxes, spheres = P.iss_corner_env(True); D, cl = 1000.0, 0.03
    s = g.BatchSolver(model, N, B, hist_cap=40, boxes=boxes, spheres=spheres)
    s.set_problems(*batch)
    X0, U0 = s.traj()
    for om in (1.0, 1e4):
        sub = s.subproblem(X0, U0, D, om, D / 8 + cl)
        print(model, N, om, "status", np.bincount(sub["status"], minlength=4), "iters sum", sub["iters"].sum(), "max", sub["iters"].max())
    s.set_problems(*batch); s.solve(30); st = s.status()
    print(model, N, "solve: conv", st["converged"].sum(), "ipm", st["ipm_iters"].sum(), "stops", np.bincount(st["stop_reason"], minlength=5))
