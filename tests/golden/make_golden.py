"""Generates the golden vectors of tests/golden/*.npz with the CPU oracle (run from the repo root).

The reference cannot run in this environment (Julia, JuMP, Ipopt/Gurobi and BulletCollision are absent) and its
own test-suite holds no vectors (test/runtests.jl is a placeholder), so these fixtures pin the *oracle* -- and
through it the HIP path -- against regressions; they are NOT outputs of the reference ("parity unpinned").
Each file stores inputs (model, N, x_init, goal, tf, env) and outputs (final X, U, per-iteration histories)."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import gusto_jl_amd as g       # noqa: E402
import gusto_oracle as go      # noqa: E402

P = g.problems
OUT = os.path.dirname(os.path.abspath(__file__))


def run(name, model, N, boxes, spheres, x0, glo, ghi, tf, max_iter=30):
    o = go.Oracle(model, N, boxes=boxes, spheres=spheres)
    res = []
    for b in range(len(x0)):
        o.set_problem(x0[b], glo[b], ghi[b], tf[b])
        Xi, Ui = o.init_straightline()
        D0, clr = o.sp.Delta0, o.mp.clearance
        sub = o.subproblem(Xi, Ui, D0, 1.0, D0 / 8 + clr)
        o.set_problem(x0[b], glo[b], ghi[b], tf[b])
        r = o.solve(max_iter)
        r["sub_X"], r["sub_U"], r["sub_obj"], r["sub_iters"], r["sub_status"] = sub["X"], sub["U"], sub["obj"], sub["iters"], sub["status"]
        res.append(r)
    keys = ["X", "U", "sub_X", "sub_U"]
    d = dict(model=model, N=N, boxes=np.zeros((0, 6)) if boxes is None else boxes,
             spheres=np.zeros((0, 4)) if spheres is None else spheres, x_init=x0, goal_lo=glo, goal_hi=ghi, tf=tf,
             max_iter=max_iter)
    for k in keys:
        d[k] = np.stack([r[k] for r in res])
    for k in ("iterations", "converged", "successful", "stop_reason", "sub_obj", "sub_iters", "sub_status"):
        d[k] = np.array([r[k] for r in res])
    H = max(len(r["omega"]) for r in res) + 2
    for k in ("J_true", "J_full", "conv", "Delta", "omega", "rho", "accept", "scp_status"):
        a = np.full((len(res), H), np.nan)
        for i, r in enumerate(res):
            a[i, :len(r[k])] = r[k]
        d[k] = a
    np.savez_compressed(os.path.join(OUT, name + ".npz"), **d)
    print(name, "iterations", d["iterations"], "converged", d["converged"])


if __name__ == "__main__":
    env = P.freeflyer_env()
    x0, glo, ghi, tf = P.freeflyer_batch(6)
    x0[0] = P.FREEFLYER_X_INIT
    run("freeflyer_se2_n50", go.FREEFLYER_SE2, 50, env, None, x0, glo, ghi, tf)
    run("freeflyer_se2_n200_notebook", go.FREEFLYER_SE2, 200, env, None, x0[:1], glo[:1], ghi[:1], tf[:1], max_iter=40)
    x0, glo, ghi, tf = P.dubins_batch(6)
    x0[0] = [2.0, 2.0, 2.0]
    run("dubins_car_n30", go.DUBINS_CAR, 30, None, None, x0, glo, ghi, tf)
    boxes, sph = P.iss_corner_env(True)
    x0, glo, ghi, tf = P.astrobee_se3_batch(3)
    run("astrobee_se3_n50", go.ASTROBEE_SE3, 50, boxes, sph, x0, glo, ghi, tf, max_iter=15)
    x0, glo, ghi, tf = P.astrobee_manifold_batch(3)
    run("astrobee_se3_manifold_n50", go.ASTROBEE_SE3_MANIFOLD, 50, boxes, sph, x0, glo, ghi, tf, max_iter=15)
