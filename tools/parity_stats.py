"""Whole-solve agreement HIP vs oracle, per problem class: python tools/parity_stats.py <model> <B> [decomposition]
prints the distribution of |X_hip - X_oracle| and of the relative J_true history difference for converged / MaxIter runs."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import gusto_jl_amd as g
import gusto_oracle as go
P = g.problems
model, B = int(sys.argv[1]), int(sys.argv[2])
dec = int(sys.argv[3]) if len(sys.argv) > 3 else 0
boxes = spheres = None
if model == 0:
    x0, glo, ghi, tf = P.freeflyer_batch(B); boxes = P.freeflyer_env(); N = 50
elif model == 1:
    x0, glo, ghi, tf = P.dubins_batch(B); N = 30
elif model == 2:
    x0, glo, ghi, tf = P.astrobee_se3_batch(B); boxes, spheres = P.iss_corner_env(True); N = 50
else:
    x0, glo, ghi, tf = P.astrobee_manifold_batch(B); boxes, spheres = P.iss_corner_env(True); N = 50
s = g.BatchSolver(model, N, B, hist_cap=40, boxes=boxes, spheres=spheres)
s.set_decomposition(dec)
s.set_problems(x0, glo, ghi, tf); s.solve(30)
X, U = s.traj(); st = s.status(); h = s.history()
o = go.Oracle(model, N, boxes=boxes, spheres=spheres)
rows = []
for b in range(B):
    o.set_problem(x0[b], glo[b], ghi[b], tf[b]); r = o.solve(30)
    nh = int(h["n_hist"][b]); c = min(nh, len(r["omega"]))
    same = nh == len(r["omega"]) and np.array_equal(h["scp_status"][b, :c], r["scp_status"][:c]) and np.array_equal(h["omega"][b, :c], r["omega"][:c])
    nJ = min(int(h["nJ"][b]), len(r["J_true"]))
    jr = np.max(np.abs(h["J_true"][b, :nJ] - r["J_true"][:nJ]) / np.maximum(1e-9, np.abs(r["J_true"][:nJ]))) if nJ else 0.0
    rows.append((same, r["converged"], r["stop_reason"], np.abs(X[b] - r["X"]).max(), jr, r["omega"].max(), sum(r["ipm_iters"]) == int(st["ipm_iters"][b])))
rows = np.array(rows, dtype=float)
print(f"model {model} B {B}: same decisions {rows[:,0].mean():.4f}, same total ipm {rows[:,6].mean():.3f}")
for name, sel in (("converged", (rows[:,0] == 1) & (rows[:,1] == 1)), ("MaxIter  ", (rows[:,0] == 1) & (rows[:,1] == 0) & (rows[:,2] == 0)), ("other    ", (rows[:,0] == 1) & (rows[:,1] == 0) & (rows[:,2] != 0))):
    if sel.sum() == 0: continue
    d, j = rows[sel, 3], rows[sel, 4]
    print(f"  {name} n={int(sel.sum())}: |dX| median {np.median(d):.1e} 90% {np.quantile(d,0.9):.1e} max {d.max():.1e};  J_true rel median {np.median(j):.1e} max {j.max():.1e}; omega max {rows[sel,5].max():.0f}")
