"""Timing of one model/config: python tools/gpu_time.py <model> <B> <N> [probe_iters]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems
model, B, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
boxes = spheres = None
if model == 0:
    x0, glo, ghi, tf = P.freeflyer_batch(B); boxes = P.freeflyer_env()
elif model == 1:
    x0, glo, ghi, tf = P.dubins_batch(B)
elif model == 2:
    x0, glo, ghi, tf = P.astrobee_se3_batch(B); boxes, spheres = P.iss_corner_env(True)
else:
    x0, glo, ghi, tf = P.astrobee_manifold_batch(B); boxes, spheres = P.iss_corner_env(True)
s = g.BatchSolver(model, N, B, hist_cap=64, boxes=boxes, spheres=spheres)
if len(sys.argv) > 4:
    s.set_schedule(int(sys.argv[4]), 1)   # probe trips of the longest-first schedule (0 = off)
for rep in range(2):
    s.set_problems(x0, glo, ghi, tf); s.solve(30)
st = s.status()
ms = s.last_solve_ms()
print(f"model {model} B={B} N={N}: kernel {ms:.1f} ms conv {st['converged'].sum()} traj/s {st['converged'].sum()/(ms/1e3):.0f} "
      f"ipm total {st['ipm_iters'].sum()} ({1e6*ms/1e3/max(1,st['ipm_iters'].sum()):.3f} us per KKT solve) stops {np.bincount(st['stop_reason'], minlength=4)}")
