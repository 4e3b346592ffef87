"""Whole TrajOpt runs of the manifold model, HIP against the oracle: how far the histories part, per solve index (the SCP loop
amplifies the last digits of every subproblem solution once steps are rejected).  python tools/to_whole_diff.py [B]   (GPU box)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gusto_jl_amd as g
import gusto_oracle as go
P = g.problems
B = int(sys.argv[1]) if len(sys.argv) > 1 else 16
boxes, spheres = P.iss_corner_env(True)
x0, glo, ghi, tf = P.astrobee_manifold_batch(B)
s = g.TrajOptSolver(g.ASTROBEE_SE3_MANIFOLD, 50, B, boxes=boxes, spheres=spheres)
s.set_problems(x0, glo, ghi, tf); s.solve(125)
X, U = s.traj(); st, h = s.status(), s.history()
o = go.OracleTrajOpt(g.ASTROBEE_SE3_MANIFOLD, 50, boxes=boxes, spheres=spheres)
worst = {k: np.zeros(16) for k in ("rho_vec", "xtol_vec", "J_true", "J_full", "conv")}
wx = 0.0
for b in range(B):
    o.set_problem(x0[b], glo[b], ghi[b], tf[b])
    R = o.solve_trajopt(125)
    S = R["solves"]
    if st["iterations"][b] != S: print("schedule differs", b); continue
    def rel(a, r): return np.abs(np.asarray(a) - np.asarray(r)) / np.maximum(1e-9, np.abs(np.asarray(r)))
    for k, (a, r) in dict(rho_vec=(h["rho_vec"][b, :S + 1], R["rho_vec"]), xtol_vec=(h["xtol_vec"][b, :len(R["xtol_vec"])], R["xtol_vec"]),
                          J_true=(h["J_true"][b, :S + 1], R["J_true"]), J_full=(h["J_full"][b, :S], R["J_full"]),
                          conv=(h["convergence_measure"][b, 1:S + 1], R["conv"][1:S + 1])).items():
        e = rel(a, r)
        n = min(len(e), 16)
        worst[k][:n] = np.maximum(worst[k][:n], e[:n])
    wx = max(wx, np.abs(X[b] - R["X"]).max() / R["mu_vec"][-1])
for k, v in worst.items():
    print(f"{k:9s} worst relative difference by solve index:", np.array2string(v[:12], precision=1, max_line_width=200))
print("final X / mu:", wx)
