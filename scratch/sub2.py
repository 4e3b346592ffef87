import sys, os, time
sys.path.insert(0, "/root/repo")
import numpy as np
import gusto_jl_amd as g
P = g.problems
boxes, sph = P.iss_corner_env(True)
x0, glo, ghi, tf = P.astrobee_se3_batch(8)
s = g.BatchSolver(g.ASTROBEE_SE3, 50, 8, hist_cap=8, boxes=boxes, spheres=sph)
s.set_problems(x0, glo, ghi, tf)
Xp, Up = s.traj()
t = time.time()
print("launching", flush=True)
r = s.subproblem(Xp, Up, 10.0, 1.0, 10.0 / 8 + 0.03)
print("done", time.time() - t, r["status"], r["iters"])
