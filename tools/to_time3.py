"""TrajOpt timing of the manifold model: python tools/to_time3.py <B>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems
B = int(sys.argv[1])
batch = P.astrobee_manifold_batch(B); boxes, spheres = P.iss_corner_env(True)
s = g.TrajOptSolver(3, 50, B, boxes=boxes, spheres=spheres)
for _ in range(2):
    s.set_problems(*batch); s.solve(125)
X, U = s.traj(); st = s.status()
print(f"trajopt model 3 B={B}: kernel {s.last_solve_ms():.1f} ms, solves {int(st['iterations'].sum())}, checksum {float(np.nansum(X)):.12e} {float(np.nansum(U)):.12e}")
