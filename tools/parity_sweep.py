"""Wide HIP-vs-oracle sweep on a GPU box: python tools/parity_sweep.py [B] -- full GuSTO solves of B freeflyer problems,
compares per-problem SCP iterations, convergence flags, KKT-solve counts and final trajectories (test tooling)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gusto_jl_amd as g
import gusto_oracle as go
P = g.problems
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
env = P.freeflyer_env()
x0, glo, ghi, tf = P.freeflyer_batch(B, first=20000)
s = g.BatchSolver(g.FREEFLYER_SE2, 50, B, hist_cap=64, boxes=env)
s.set_schedule(2, 1)
s.set_problems(x0, glo, ghi, tf)
s.solve(30)
X, U = s.traj(); st = s.status()
r = go.solve_batch(go.FREEFLYER_SE2, 50, env, None, x0, glo, ghi, tf, 30, 0)
same_it = st["iterations"] == r["iterations"]
same_cv = st["converged"].astype(bool) == r["converged"]
dx = np.abs(X - r["X"]).reshape(B, -1).max(1)
print(f"B={B}: identical SCP iteration count {same_it.mean()*100:.2f}%  identical converged flag {same_cv.mean()*100:.2f}%")
print(f"  among identical-iteration problems: max|dX| median {np.median(dx[same_it]):.2e}  99% {np.quantile(dx[same_it], .99):.2e}  max {dx[same_it].max():.2e}")
print(f"  KKT solves gpu {st['ipm_iters'].sum()} oracle {r['ipm_iters'].sum()}  converged gpu {st['converged'].sum()} oracle {r['converged'].sum()}")
bad = np.where(~same_it)[0]
print("  problems with different iteration counts:", [(int(b), int(st['iterations'][b]), int(r['iterations'][b])) for b in bad[:12]])
