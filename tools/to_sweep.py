"""TrajOpt parity sweep, HIP against the oracle, whole runs on more problems than the tests use (no assertions: counts).
   python tools/to_sweep.py [B_freeflyer B_astrobee B_manifold]   (on a GPU box)"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "oracle"))
import numpy as np
import gusto_jl_amd as g
import gusto_oracle as go
P = g.problems
Bs = [int(a) for a in sys.argv[1:4]] if len(sys.argv) >= 4 else [1024, 256, 128]
for model, B in zip((g.FREEFLYER_SE2, g.ASTROBEE_SE3, g.ASTROBEE_SE3_MANIFOLD), Bs):
    if model == g.FREEFLYER_SE2: batch, boxes, spheres = P.freeflyer_batch(B), P.freeflyer_env(), None
    elif model == g.ASTROBEE_SE3: batch = P.astrobee_se3_batch(B); boxes, spheres = P.iss_corner_env(True)
    else: batch = P.astrobee_manifold_batch(B); boxes, spheres = P.iss_corner_env(True)
    x0, glo, ghi, tf = batch
    s = g.TrajOptSolver(model, 50, B, boxes=boxes, spheres=spheres)
    s.set_problems(x0, glo, ghi, tf); s.solve(125)
    X, U = s.traj(); st, h = s.status(), s.history()
    o = go.OracleTrajOpt(model, 50, boxes=boxes, spheres=spheres)
    t0 = time.time()
    same, div, ex, ej, eu = 0, [], [], [], []
    ipm_d = ipm_o = 0
    for b in range(B):
        o.set_problem(x0[b], glo[b], ghi[b], tf[b])
        R = o.solve_trajopt(125)
        S = R["solves"]
        ipm_d += int(st["ipm_iters"][b]); ipm_o += int(np.sum(R["ipm_iters"])) if "ipm_iters" in R else 0
        sched = (st["iterations"][b] == S and bool(st["converged"][b]) == R["converged"] and st["stop_reason"][b] == R["stop_reason"]
                 and h["n_mu"][b] == len(R["mu_vec"]) and np.array_equal(h["s_vec"][b, :S + 1], R["s_vec"])
                 and np.array_equal(h["mu_vec"][b, :h["n_mu"][b]], R["mu_vec"]))
        if not sched:
            div.append((b, int(st["iterations"][b]), S, int(st["stop_reason"][b]), int(R["stop_reason"])))
            continue
        same += 1
        ex.append(np.abs(X[b] - R["X"]).max() / max(1.0, R["mu_vec"][-1])); eu.append(np.abs(U[b] - R["U"]).max() / max(1.0, R["mu_vec"][-1]))
        jt = np.asarray(R["J_true"]); ej.append(np.max(np.abs(h["J_true"][b, :S + 1] - jt) / np.maximum(1e-12, np.abs(jt))))
    ex, eu, ej = np.array(ex), np.array(eu), np.array(ej)
    print(f"model {model} B={B}: identical schedules (solves, converged, stop, s_vec, mu_vec) {same} of {B} = {100.0 * same / B:.2f} %; "
          f"of those: |dX|/max(1,mu) max {ex.max():.2e} median {np.median(ex):.1e}, |dU| max {eu.max():.2e}, J_true rel max {ej.max():.2e} median {np.median(ej):.1e} "
          f"(> 1e-7: {(ej > 1e-7).sum()}, X > 5e-5: {(ex > 5e-5).sum()}); oracle {time.time() - t0:.0f} s", flush=True)
    print("   divergent (b, device solves, oracle solves, device stop, oracle stop):", div[:24], flush=True)
