"""TrajOpt: HIP against the oracle on a few problems (subproblem + whole solve).  python tools/to_check.py <model 0|2> <B>"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import gusto_jl_amd as g
import gusto_oracle as go
P = g.problems
model, B = int(sys.argv[1]), int(sys.argv[2])
if model == 0:
    x0, glo, ghi, tf = P.freeflyer_batch(B); boxes, spheres = P.freeflyer_env(), None
else:
    x0, glo, ghi, tf = P.astrobee_se3_batch(B); boxes, spheres = P.iss_corner_env(True)
N = 50
s = g.TrajOptSolver(model, N, B, boxes=boxes, spheres=spheres)
s.set_problems(x0, glo, ghi, tf)
X0, U0 = s.traj()
o = go.OracleTrajOpt(model, N, boxes=boxes, spheres=spheres)
for mu, tr in ((1.0, 1.0), (25.0, 0.0625)):
    r = s.subproblem(X0, U0, mu, tr)
    worst = 0
    for b in range(B):
        o.set_problem(x0[b], glo[b], ghi[b], tf[b])
        ro = o.subproblem(X0[b], U0[b], mu, tr)
        dx, du, dd = np.abs(r["X"][b] - ro["X"]).max(), np.abs(r["U"][b] - ro["U"]).max(), np.abs(r["D"][b] - ro["D"]).max()
        worst = max(worst, dx, du, dd)
        if b < 3 or r["status"][b] != ro["status"]:
            print(f"  sub mu={mu} s={tr} b={b}: status {r['status'][b]}/{ro['status']} iters {r['iters'][b]}/{ro['iters']} obj {r['obj'][b]:.9f}/{ro['obj']:.9f} dX {dx:.2e} dU {du:.2e} dD {dd:.2e}")
    print(f"subproblem mu={mu} s={tr}: worst |d| {worst:.3e}")
s.set_problems(x0, glo, ghi, tf)
s.solve(125)
X, U = s.traj(); st = s.status(); h = s.history()
print("kernel ms", s.last_solve_ms(), "solves", np.bincount(st["iterations"]), "converged", st["converged"].sum(), "stops", np.bincount(st["stop_reason"], minlength=5))
bad = 0
for b in range(B):
    o.set_problem(x0[b], glo[b], ghi[b], tf[b])
    R = o.solve_trajopt(125)
    same = R["solves"] == st["iterations"][b] and R["converged"] == st["converged"][b]
    dx = np.abs(X[b] - R["X"]).max(); du = np.abs(U[b] - R["U"]).max()
    ns = R["solves"]
    drho = np.abs(h["rho_vec"][b, :ns + 1] - R["rho_vec"]).max() if same else np.nan
    if b < 4 or not same:
        print(f"  b={b}: solves {st['iterations'][b]}/{R['solves']} conv {st['converged'][b]}/{R['converged']} dX {dx:.2e} dU {du:.2e} drho {drho:.2e} mu {h['mu_vec'][b, :h['n_mu'][b]]} / {R['mu_vec']} ctol {np.round(h['ctol_vec'][b, :h['n_ctol'][b]], 5)} / {np.round(R['ctol_vec'], 5)}")
    bad += not same
print("problems with a different schedule:", bad, "of", B)
