"""Diagnostic run on a GPU box: HIP path vs oracle on a few problems, then a timing of config 2."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle"))
import numpy as np
import gusto_jl_amd as g
import gusto_oracle as go

P = g.problems
env = P.freeflyer_env()
N = 50
B = int(sys.argv[1]) if len(sys.argv) > 1 else 8

# 1. subproblem parity
x0, glo, ghi, tf = P.freeflyer_batch(B)
x0[0] = P.FREEFLYER_X_INIT
s = g.BatchSolver(g.FREEFLYER_SE2, N, max(B, 4096), hist_cap=64, boxes=env)
s.set_problems(x0, glo, ghi, tf)
X0, U0 = s.traj()
t0 = time.time()
r = s.subproblem(X0, U0, 3.0, 1.0, 3.0 / 8 + 0.05)
print("subproblem wall", time.time() - t0, "kernel ms", s.last_solve_ms())
o = go.Oracle(go.FREEFLYER_SE2, N, boxes=env)
for b in range(min(B, 4)):
    o.set_problem(x0[b], glo[b], ghi[b], tf[b])
    Xp, Up = o.init_straightline()
    print("init diff", np.abs(Xp - X0[b]).max())
    ro = o.subproblem(Xp, Up, 3.0, 1.0, 3.0 / 8 + 0.05)
    print(b, "status", r["status"][b], ro["status"], "iters", r["iters"][b], ro["iters"], "obj", r["obj"][b], ro["obj"],
          "dX", np.abs(r["X"][b] - ro["X"]).max(), "dU", np.abs(r["U"][b] - ro["U"]).max(),
          "ddual", np.abs(r["dual"][b] - ro["dual"]).max())

# 2. full SCP
s.set_problems(x0, glo, ghi, tf)
t0 = time.time()
s.solve(30)
print("solve wall", time.time() - t0, "kernel ms", s.last_solve_ms())
X, U = s.traj()
st = s.status()
h = s.history()
for b in range(min(B, 8)):
    o.set_problem(x0[b], glo[b], ghi[b], tf[b])
    ro = o.solve(30)
    print(b, "iters", st["iterations"][b], ro["iterations"], "conv", st["converged"][b], ro["converged"], "succ",
          st["successful"][b], ro["successful"], "ipm", st["ipm_iters"][b], ro["total_ipm_iters"], "dX",
          np.abs(X[b] - ro["X"]).max(), "dU", np.abs(U[b] - ro["U"]).max(), "J", h["J_true"][b, h["nJ"][b] - 1],
          ro["J_true"][-1])
print("converged", st["converged"].sum(), "of", B, "mean iters", st["iterations"].mean(), "mean ipm", st["ipm_iters"].mean())

# 3. timing config 2
if len(sys.argv) > 2:
    Bb = int(sys.argv[2])
    x0, glo, ghi, tf = P.freeflyer_batch(Bb)
    for rep in range(2):
        s.set_problems(x0, glo, ghi, tf)
        t0 = time.time()
        s.solve(30)
        w = time.time() - t0
        st = s.status()
        print(f"B={Bb} wall {w:.3f}s kernel {s.last_solve_ms():.1f} ms converged {st['converged'].sum()} "
              f"traj/s {st['converged'].sum() / (s.last_solve_ms() / 1e3):.1f} mean iters {st['iterations'].mean():.2f} "
              f"mean ipm {st['ipm_iters'].mean():.1f} stops {np.bincount(st['stop_reason'], minlength=4)}")
