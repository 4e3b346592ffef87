"""TrajOpt timing: python tools/to_time.py <model 0|2> <B>   (two runs; prints kernel ms and a checksum of the result)"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems
model, B = int(sys.argv[1]), int(sys.argv[2])
if model == 0:
    batch = P.freeflyer_batch(B); boxes, spheres = P.freeflyer_env(), None
else:
    batch = P.astrobee_se3_batch(B); boxes, spheres = P.iss_corner_env(True)
s = g.TrajOptSolver(model, 50, B, boxes=boxes, spheres=spheres)
for _ in range(2):
    s.set_problems(*batch); s.solve(125)
X, U = s.traj(); st = s.status()
print(f"trajopt model {model} B={B}: kernel {s.last_solve_ms():.1f} ms, converged {int(st['converged'].sum())}, solves {int(st['iterations'].sum())}, "
      f"checksum {float(np.nansum(X)):.12e} {float(np.nansum(U)):.12e}")
