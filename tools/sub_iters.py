"""IPM iteration counts of one subproblem per problem, device vs oracle: python tools/sub_iters.py <omega> <Delta> [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gusto_jl_amd as g
import gusto_oracle as go
P = g.problems
omega, Delta = float(sys.argv[1]), float(sys.argv[2])
B = int(sys.argv[3]) if len(sys.argv) > 3 else 48
x0, glo, ghi, tf = P.freeflyer_batch(B)
x0[0] = P.FREEFLYER_X_INIT
env = P.freeflyer_env()
s = g.BatchSolver(g.FREEFLYER_SE2, 50, B, hist_cap=8, boxes=env)
s.set_problems(x0, glo, ghi, tf)
Xp, Up = s.traj()
r = s.subproblem(Xp, Up, Delta, omega, Delta / 8 + 0.05)
o = go.Oracle(g.FREEFLYER_SE2, 50, boxes=env)
oi = []
for b in range(B):
    o.set_problem(x0[b], glo[b], ghi[b], tf[b])
    ro = o.subproblem(Xp[b], Up[b], Delta, omega, Delta / 8 + 0.05)
    oi.append(ro["iters"])
oi = np.array(oi)
d = r["iters"] - oi
print(f"omega {omega} Delta {Delta}: device iters sum {r['iters'].sum()} oracle {oi.sum()}; diff histogram {dict(zip(*np.unique(d, return_counts=True)))}")
