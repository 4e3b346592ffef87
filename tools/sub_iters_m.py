"""IPM iteration counts of one subproblem per problem for any model, device only: python tools/sub_iters_m.py <model> <omega> <Delta> [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import gusto_jl_amd as g
P = g.problems
model, omega, Delta = int(sys.argv[1]), float(sys.argv[2]), float(sys.argv[3])
B = int(sys.argv[4]) if len(sys.argv) > 4 else 96
N = 30 if model == 1 else 50
boxes = spheres = None
if model == 0: batch = P.freeflyer_batch(B); boxes = P.freeflyer_env(); cl = 0.05
elif model == 1: batch = P.dubins_batch(B); cl = 0.01
elif model == 2: batch = P.astrobee_se3_batch(B); boxes, spheres = P.iss_corner_env(True); cl = 0.03
else: batch = P.astrobee_manifold_batch(B); boxes, spheres = P.iss_corner_env(True); cl = 0.03
s = g.BatchSolver(model, N, B, hist_cap=40, boxes=boxes, spheres=spheres)
s.set_problems(*batch)
s.solve(3)                      # a few trips: a linearisation point that is not the straight line
Xp, Up = s.traj()
r = s.subproblem(Xp, Up, Delta, omega, Delta / 8 + cl)
it = r["iters"]
print(f"model {model} omega {omega} Delta {Delta}: iters sum {it.sum()} max {it.max()} status {np.bincount(r['status'], minlength=4)} obj checksum {np.nansum(r['obj']):.10g}")
