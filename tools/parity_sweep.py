"""Wide HIP-vs-oracle sweep on a GPU box: python tools/parity_sweep.py [B] [model] -- full GuSTO solves of B problems of
one model (0 freeflyerSE2, 1 dubins, 2 astrobeeSE3, 3 manifold), compares per-problem SCP iterations, convergence flags,
KKT-solve counts and final trajectories (test tooling)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gusto_jl_amd as g
import gusto_oracle as go
P = g.problems
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
model = int(sys.argv[2]) if len(sys.argv) > 2 else 0
env = sph = None
N = 50
if model == 0:
    env = P.freeflyer_env(); x0, glo, ghi, tf = P.freeflyer_batch(B, first=20000)
elif model == 1:
    N = 30; x0, glo, ghi, tf = P.dubins_batch(B, first=20000)
elif model == 2:
    env, sph = P.iss_corner_env(True); x0, glo, ghi, tf = P.astrobee_se3_batch(B)
else:
    env, sph = P.iss_corner_env(True); x0, glo, ghi, tf = P.astrobee_manifold_batch(B)
s = g.BatchSolver(model, N, B, hist_cap=64, boxes=env, spheres=sph)
s.set_schedule(2, 1)
s.set_problems(x0, glo, ghi, tf)
s.solve(30)
X, U = s.traj(); st = s.status()
r = go.solve_batch(model, N, env, sph, x0, glo, ghi, tf, 30, 0)
same_it = st["iterations"] == r["iterations"]
same_cv = st["converged"].astype(bool) == r["converged"]
dx = np.abs(X - r["X"]).reshape(B, -1).max(1)
print(f"model {model} B={B}: identical SCP iteration count {same_it.mean()*100:.2f}%  identical converged flag {same_cv.mean()*100:.2f}%")
both = same_it & st["converged"].astype(bool) & r["converged"]
print(f"  converged on both sides with identical iteration counts ({both.sum()}): max|dX| median {np.median(dx[both]):.2e}  "
      f"99% {np.quantile(dx[both], .99):.2e}  max {dx[both].max():.2e}")
rest = same_it & ~both
if rest.any():
    print(f"  not converged ({rest.sum()}, 30 trips of an SCP that does not settle): max|dX| median {np.median(dx[rest]):.2e}  max {dx[rest].max():.2e}")
print(f"  KKT solves gpu {st['ipm_iters'].sum()} oracle {r['ipm_iters'].sum()}  converged gpu {st['converged'].sum()} oracle {r['converged'].sum()}")
bad = np.where(~same_it)[0]
print("  problems with different iteration counts:", [(int(b), int(st['iterations'][b]), int(r['iterations'][b])) for b in bad[:12]])
