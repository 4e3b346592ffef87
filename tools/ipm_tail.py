"""Per-problem interior point work of a batch: python tools/ipm_tail.py <model> <B> -- the longest problems and their per-trip iteration counts"""
import os, sys
sys.path.insert(0, os.getcwd())
import numpy as np, gusto_jl_amd as g
P = g.problems
model, B = int(sys.argv[1]), int(sys.argv[2])
N = 30 if model == 1 else 50
boxes = spheres = None
if model == 0: batch = P.freeflyer_batch(B); boxes = P.freeflyer_env()
elif model == 1: batch = P.dubins_batch(B)
elif model == 2: batch = P.astrobee_se3_batch(B); boxes, spheres = P.iss_corner_env(True)
else: batch = P.astrobee_manifold_batch(B); boxes, spheres = P.iss_corner_env(True)
s = g.BatchSolver(model, N, B, hist_cap=64, boxes=boxes, spheres=spheres)
s.set_problems(*batch); s.solve(30)
st, h = s.status(), s.history()
ipm = st["ipm_iters"]
order = np.argsort(-ipm)[:6]
print(f"kernel {s.last_solve_ms():.1f} ms; ipm total {ipm.sum()} mean {ipm.mean():.1f} max {ipm.max()}; solver_status counts {np.bincount(h['solver_status'][h['solver_status'] > 0], minlength=4)}")
for b in order:
    nh = h["n_hist"][b]
    print(b, "ipm", ipm[b], "trips", st["iterations"][b], "per trip", list(h["ipm_iters"][b, 1:nh]), "status", list(h["solver_status"][b, 1:nh]), "omega", [f"{v:g}" for v in h["omega"][b, :nh]][-1])
