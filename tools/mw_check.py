"""multi-wave kernel (N > 64) of a model against the oracle: python tools/mw_check.py <model> <N>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import gusto_jl_amd as g, gusto_oracle as go
P = g.problems
model, N, B = int(sys.argv[1]), int(sys.argv[2]), 4
if model == 0:
    x0, glo, ghi, tf = P.freeflyer_batch(B); boxes, spheres = P.freeflyer_env(), None
elif model == 2:
    x0, glo, ghi, tf = P.astrobee_se3_batch(B); boxes, spheres = P.iss_corner_env(True)
else:
    x0, glo, ghi, tf = P.astrobee_manifold_batch(B); boxes, spheres = P.iss_corner_env(True)
s = g.BatchSolver(model, N, B, boxes=boxes, spheres=spheres)
s.set_problems(x0, glo, ghi, tf)
X0, U0 = s.traj()
sp, mp = g.default_params(model)
r = s.subproblem(X0, U0, sp.Delta0, 1.0, sp.Delta0 / 8 + mp.clearance)
o = go.Oracle(model, N, boxes=boxes, spheres=spheres)
for b in range(B):
    o.set_problem(x0[b], glo[b], ghi[b], tf[b])
    ro = o.subproblem(X0[b], U0[b], sp.Delta0, 1.0, sp.Delta0 / 8 + mp.clearance)
    print(b, 'status', r['status'][b], ro['status'], 'iters', r['iters'][b], ro['iters'], 'dX', np.abs(r['X'][b] - ro['X']).max(), 'dU', np.abs(r['U'][b] - ro['U']).max())
