"""Distribution of the work per problem of a config (interior point iterations, trips): python tools/work_dist.py <model> <B> <N>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import gusto_jl_amd as g
P = g.problems
model, B, N = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
FIRST = int(os.environ.get("FIRST", "0"))
boxes = spheres = None
if model == 0: x0, glo, ghi, tf = P.freeflyer_batch(B, FIRST); boxes = P.freeflyer_env()
elif model == 1: x0, glo, ghi, tf = P.dubins_batch(B, FIRST)
elif model == 2: x0, glo, ghi, tf = P.astrobee_se3_batch(B, FIRST); boxes, spheres = P.iss_corner_env(True)
else: x0, glo, ghi, tf = P.astrobee_manifold_batch(B, FIRST); boxes, spheres = P.iss_corner_env(True)
io = g.default_ipm_opts()
if os.environ.get("MUFLOOR"): io.mu_floor = float(os.environ["MUFLOOR"])
if os.environ.get("ACC"): io.acc_iter = int(os.environ["ACC"])
s = g.BatchSolver(model, N, B, hist_cap=64, boxes=boxes, spheres=spheres, ipm_opts=io)
s.set_problems(x0, glo, ghi, tf); s.solve(30)
s.set_problems(x0, glo, ghi, tf); s.solve(30)
st = s.status(); h = s.history(); ms = s.last_solve_ms(); slots = s.launch_info()[0]
w = st["ipm_iters"]
print(f"model {model} B {B}: kernel {ms:.1f} ms, slots {slots}, KKT total {w.sum()} = {w.sum()/slots:.0f} per slot -> balanced {ms * (w.sum()/slots) / max(1,w.max()):.1f} ms if the longest problem ({w.max()} KKT) set the time")
print(" quantiles of KKT per problem 50/90/99/99.9/max:", np.quantile(w, [0.5, 0.9, 0.99, 0.999]).astype(int), w.max())
top = np.argsort(-w)[:8]
for b in top:
    nh = h["n_hist"][b]
    print(f"  problem {b}: KKT {w[b]} trips {st['iterations'][b]} stop {st['stop_reason'][b]} conv {st['converged'][b]} omega_end {h['omega'][b, nh-1]:.0f} ipm/trip {h['ipm_iters'][b, 1:nh][:30]} solver {h['solver_status'][b, 1:nh][:30]}")
