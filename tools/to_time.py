"""TrajOpt timing + checksum of one model: python tools/to_time.py <model 0|2|3> <B>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems
model, B = int(sys.argv[1]), int(sys.argv[2])
if model == 0: batch, boxes, spheres = P.freeflyer_batch(B), P.freeflyer_env(), None
elif model == 2: batch = P.astrobee_se3_batch(B); boxes, spheres = P.iss_corner_env(True)
else: batch = P.astrobee_manifold_batch(B); boxes, spheres = P.iss_corner_env(True)
s = g.TrajOptSolver(model, 50, B, boxes=boxes, spheres=spheres)
for _ in range(2):
    s.set_problems(*batch); s.solve(125)
X, U = s.traj(); st = s.status()
print(f"trajopt model {model} B={B}: kernel {s.last_solve_ms():.1f} ms, solves {int(st['iterations'].sum())}, ipm {int(st['ipm_iters'].sum())}, "
      f"converged {int(st['converged'].sum())}, checksum {float(np.nansum(X)):.12e} {float(np.nansum(U)):.12e}")
