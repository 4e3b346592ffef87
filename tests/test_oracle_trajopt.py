"""Pins the oracle's TrajOpt restatement (src/scp/scp_trajopt.jl) without a GPU:
  * its convex subproblem against scipy SLSQP on an INDEPENDENT formulation -- no defect variables: the dynamics penalty is
    written as mu |F_k(X, U)|_1 on the trapezoid defects themselves, rows rebuilt here from the model constants;
  * trust_region_ratio_trajopt / evaluate_ctol against numpy restatements of the formulas DESIGN.md section 4 states;
  * the three-loop schedule (penalty mu x k, convex iterations, trust region s x tau+-) on its own histories.
The reference cannot run here (no Julia) and TrajOpt does not run as written at its HEAD either: "parity unpinned"."""
import numpy as np
import pytest
import scipy.optimize as so

import gusto_oracle as go
import gusto_jl_amd as g

P = g.problems
REG = 1e-4      # GO_TRAJOPT_DEFECT_REG


def _ff(N, env, b=0):
    x0, glo, ghi, tf = P.freeflyer_batch(b + 1)
    o = go.OracleTrajOpt(go.FREEFLYER_SE2, N, boxes=env)
    o.set_problem(x0[b], glo[b], ghi[b], tf[b])
    return o, x0[b], glo[b], tf[b]


@pytest.mark.parametrize("N,mu,s_tr,with_env", [(10, 1.0, 1.0, True), (12, 5.0, 0.25, True), (16, 25.0, 0.0625, False)])
def test_subproblem_against_slsqp(N, mu, s_tr, with_env):
    env = P.freeflyer_env() if with_env else None
    o, x_init, goal, tf = _ff(N, env)
    Xp, Up = o.init_straightline()
    r = o.subproblem(Xp, Up, mu, s_tr)
    assert r["status"] == 1
    n, m = 6, 3
    mp = o.mp
    dt = tf / (N - 1)
    kappa = 1.0 / max(1.0, mu)
    w = np.full(N, dt); w[0] = w[-1] = 0.5 * dt
    A = np.kron(np.array([[0.0, 1.0], [0.0, 0.0]]), np.eye(3))
    Bm = np.zeros((6, 3)); Bm[3, 0] = Bm[4, 1] = 1 / mp.mass; Bm[5, 2] = 1 / mp.Jdiag[2]
    # active obstacle rows: linearised at Xp inside obstacle_toggle_distance = clearance + 1 (scp_trajopt.jl:64)
    obs = []
    for k in range(N):
        for i in range(len(o.boxes)):
            d, nh = o.signed_distance(0, Xp[k, :2], i)
            if d < mp.clearance + 1.0:
                obs.append((k, nh[:2].copy(), mp.clearance - d + nh[:2] @ Xp[k, :2]))
    npen = 2 * N + len(obs) + 2 * (N - 1)          # vel, omega per knot; obstacles; two control rows for k < N-1
    ndef = (N - 1) * n
    nx = N * (n + m)

    def split(z):
        Z = z[:nx].reshape(N, n + m)
        return Z[:, :n], Z[:, n:], z[nx:nx + npen], z[nx + npen:]

    def defects(X, U):
        a = X @ A.T + U @ Bm.T
        return X[1:] - X[:-1] - 0.5 * dt * (a[:-1] + a[1:])

    def pen_values(X, U):
        out = [X[:, 3] ** 2 + X[:, 4] ** 2 - mp.hard_limit_vel ** 2, X[:, 5] ** 2 - mp.hard_limit_omega ** 2,
               np.array([c0 - nh @ X[k, :2] for k, nh, c0 in obs]),
               (U[:-1, 0] ** 2 + U[:-1, 1] ** 2) / mp.mass ** 2 - mp.hard_limit_accel ** 2,
               U[:-1, 2] ** 2 / mp.Jdiag[2] ** 2 - mp.hard_limit_alpha ** 2]
        return np.concatenate(out)

    def obj(z):
        X, U, V, W = split(z)
        F = defects(X, U)
        return kappa * (np.sum(w[:, None] * U ** 2) + REG * np.sum(w[:-1, None] * F ** 2)) + V.sum() + W.sum()

    def ineq(z):      # >= 0
        X, U, V, W = split(z)
        F = (kappa * mu) * defects(X, U).ravel()
        tr = s_tr - np.sum((X - Xp) ** 2, axis=1)
        return np.concatenate([V - (kappa * mu) * pen_values(X, U), W - F, W + F, tr])

    eq = lambda z: np.concatenate([split(z)[0][0] - x_init, split(z)[0][-1] - goal])
    z0 = np.concatenate([np.hstack([Xp, Up]).ravel(), np.ones(npen), np.ones(ndef)])
    res = so.minimize(obj, z0, method="SLSQP", constraints=[{"type": "eq", "fun": eq}, {"type": "ineq", "fun": ineq}],
                      bounds=[(None, None)] * nx + [(0, None)] * (npen + ndef), options={"ftol": 1e-14, "maxiter": 800})
    Xs, Us, _, _ = split(res.x)
    assert abs(res.fun / kappa - r["obj"]) <= 2e-6 * max(1.0, abs(r["obj"])), (res.fun / kappa, r["obj"], res.message)
    assert np.abs(Xs - r["X"]).max() < 5e-4 and np.abs(Us - r["U"]).max() < 5e-4
    # the defect variables of the oracle ARE the trapezoid defects of its optimum
    assert np.abs(defects(r["X"], r["U"]) - r["D"][:-1]).max() < 1e-7 and np.abs(r["D"][-1]).max() < 1e-7


def test_ratio_and_ctol_formulas():
    """trust_region_ratio_trajopt and evaluate_ctol as DESIGN.md section 4 reads them, restated with numpy."""
    N = 20
    env = P.freeflyer_env()
    o, x_init, goal, tf = _ff(N, env, b=3)
    Xp, Up = o.init_straightline()
    r = o.subproblem(Xp, Up, 1.0, 1.0)
    X, U = r["X"], r["U"]
    mp, dt = o.mp, tf / (N - 1)
    A = np.kron(np.array([[0.0, 1.0], [0.0, 0.0]]), np.eye(3))
    Bm = np.zeros((6, 3)); Bm[3, 0] = Bm[4, 1] = 1 / mp.mass; Bm[5, 2] = 1 / mp.Jdiag[2]
    f = lambda x, u: A @ x + Bm @ u
    num = den = 0.0
    for k in range(N - 1):
        po = np.abs(f(Xp[k], Up[k]) - (Xp[k + 1] - Xp[k]) / dt).sum()
        pn = np.abs(f(X[k], U[k]) - (X[k + 1] - X[k]) / dt).sum()
        ph = np.abs(X[k + 1] - X[k] - 0.5 * dt * (f(X[k], U[k]) + f(X[k + 1], U[k + 1]))).sum()     # (the model is linear)
        num += po - pn; den += po - ph
    off = np.array([[0, 0, 0], list(mp.comp_off[1])])
    for k in range(N):
        for c in range(mp.n_robot_comp):
            for i in range(len(o.boxes)):
                d0, nh = o.signed_distance(c, Xp[k, :2], i)
                d1, _ = o.signed_distance(c, X[k, :2], i)
                lin = d0 + nh[:2] @ (X[k, :2] - Xp[k, :2])
                num += d1 - d0; den += lin - d0
    assert abs(o.ratio(X, U, Xp, Up) - num / den) < 1e-9 * max(1.0, abs(num / den))
    # evaluate_ctol: per class max |g(traj) - g(prev)| and max |g(traj)|, summed
    JN = JD = 0.0
    for g_ in (lambda Z: Z[:, 3] ** 2 + Z[:, 4] ** 2 - mp.hard_limit_vel ** 2, lambda Z: Z[:, 5] ** 2 - mp.hard_limit_omega ** 2,
               lambda Z: np.array([[mp.clearance - o.signed_distance(0, Z[k, :2], i)[0] for i in range(len(o.boxes))] for k in range(N)])):
        JN += np.abs(g_(X) - g_(Xp)).max(); JD += np.abs(g_(X)).max()
    Fd = lambda Z, V: np.array([Z[k + 1] - Z[k] - 0.5 * dt * (f(Z[k], V[k]) + f(Z[k + 1], V[k + 1])) for k in range(N - 1)])
    JN += np.linalg.norm(Fd(X, U) - Fd(Xp, Up), axis=1).max(); JD += np.linalg.norm(Fd(X, U), axis=1).max()
    assert abs(o.ctol(X, U, Xp, Up) - JN / JD) < 1e-9


@pytest.mark.parametrize("model", [go.FREEFLYER_SE2, go.ASTROBEE_SE3])
def test_three_loop_schedule(model):
    """scp_trajopt.jl:69-163 on the oracle's own histories: one rho / s entry per solve, s halves or doubles by rho > c, mu
    grows by k once per unsatisfied penalty iteration, a trust loop ends when s < xtol, every step is taken."""
    if model == go.FREEFLYER_SE2:
        x0, glo, ghi, tf = P.freeflyer_batch(6); boxes, spheres = P.freeflyer_env(), None
    else:
        x0, glo, ghi, tf = P.astrobee_se3_batch(6); boxes, spheres = P.iss_corner_env(True)
    o = go.OracleTrajOpt(model, 50, boxes=boxes, spheres=spheres)
    tp = o.tp
    for b in range(6):
        o.set_problem(x0[b], glo[b], ghi[b], tf[b])
        R = o.solve_trajopt(125)
        S = R["solves"]
        assert S == R["iterations"] >= 1 and len(R["rho_vec"]) == len(R["s_vec"]) == S + 1 and len(R["J_true"]) == S + 1
        assert len(R["J_full"]) == S and R["s_vec"][0] == tp.s0 and R["mu_vec"][0] == tp.mu0 and R["rho_vec"][0] == 0.0
        for i in range(S):
            fac = tp.tau_plus if R["rho_vec"][i + 1] > tp.c else tp.tau_minus
            assert R["s_vec"][i + 1] == fac * R["s_vec"][i]
        assert np.allclose(R["mu_vec"][1:] / R["mu_vec"][:-1], tp.k) and len(R["mu_vec"]) <= tp.max_penalty_iteration + 1
        assert all(st in (1, 2) for st in R["solver_status"][1:S + 1]) and R["stop_reason"] in (0, 1)
        assert R["converged"] == (R["ctol_vec"][-1] < tp.ctol and R["stop_reason"] == 1)
        assert np.abs(R["X"][0] - x0[b]).max() < 1e-9 and np.abs(R["X"][-1] - glo[b]).max() < 1e-7     # hard boundary rows
        assert np.abs(R["D"]).max() < 1e-6 or not R["converged"] or R["mu_vec"][-1] == tp.mu0          # defects vanish as mu grows
    # max_iter caps the subproblem solves (the reference computes iter_cap and never reads it)
    o.set_problem(x0[0], glo[0], ghi[0], tf[0])
    assert o.solve_trajopt(3)["solves"] == 3
    with pytest.raises(RuntimeError):
        go.OracleTrajOpt(go.DUBINS_CAR, 30)


def test_manifold_subproblem_rows_as_the_reference_registers_them():
    """TrajOpt for AstrobeeSE3Manifold (astrobee_se3_manifold.jl:56-70,533-608; scp_trajopt.jl:159-279): no trust region row
    (none is registered, :601), the linearised quaternion norm of every knot held hard -- inside the band the oracle states --,
    BoxGoal rows on q hard, -qw / speed / rate / obstacle / control rows and the dynamics penalised.  Checked on the optimum of
    one subproblem: the trust region argument s does not move it, the hard rows hold, the reported objective is the control
    effort plus the L1 penalties evaluated independently in numpy.  Then the three-loop schedule of whole runs."""
    boxes, spheres = P.iss_corner_env(True)
    x0, glo, ghi, tf = P.astrobee_manifold_batch(3)
    N, mu = 50, 5.0
    o = go.OracleTrajOpt(go.ASTROBEE_SE3_MANIFOLD, N, boxes=boxes, spheres=spheres)
    mp = o.mp
    for b in range(3):
        o.set_problem(x0[b], glo[b], ghi[b], tf[b])
        Xp, Up = o.init_straightline()
        r = o.subproblem(Xp, Up, mu, 10.0)
        r2 = o.subproblem(Xp, Up, mu, 1e-3)
        assert r["status"] == 1 and r2["status"] == 1
        assert np.array_equal(r["X"], r2["X"]) and np.array_equal(r["U"], r2["U"])      # no trust region row at all
        X, U, D = r["X"], r["U"], r["D"]
        qn = np.linalg.norm(Xp[:, 6:10], axis=1)
        h = qn + ((Xp[:, 6:10] / qn[:, None]) * (X[:, 6:10] - Xp[:, 6:10])).sum(1) - 1.0
        assert np.abs(h).max() <= 1e-7                                                    # cse_quaternion_norm: hard `== 0` (an equality row; round 6)
        assert np.abs(X[0] - x0[b]).max() < 1e-9
        pt = glo[b] == ghi[b]
        assert np.abs(X[-1][pt] - glo[b][pt]).max() < 1e-7
        assert (X[-1][~pt] <= ghi[b][~pt] + 1e-9).all() and (X[-1][~pt] >= glo[b][~pt] - 1e-9).all()
        dt = tf[b] / (N - 1)
        cost = sum(0.5 * dt * (U[k - 1] @ U[k - 1] + U[k] @ U[k]) for k in range(1, N))
        pen = 0.0
        for k in range(N):
            pen += max(0.0, mu * -X[k, 6]) + max(0.0, mu * (X[k, 3:6] @ X[k, 3:6] - mp.hard_limit_vel ** 2))
            pen += max(0.0, mu * (X[k, 10:13] @ X[k, 10:13] - mp.hard_limit_omega ** 2))
            for i in range(len(boxes) + len(spheres)):
                d0, nh = o.signed_distance(0, Xp[k, :3], i)
                if d0 < mp.clearance + 1.0:
                    pen += max(0.0, mu * (mp.clearance - (d0 + nh @ (X[k, :3] - Xp[k, :3]))))
            if k < N - 1:
                pen += max(0.0, mu * (U[k, :3] @ U[k, :3] / mp.mass ** 2 - mp.hard_limit_accel ** 2))
                pen += max(0.0, mu * (sum((U[k, 3 + j] / mp.Jdiag[j]) ** 2 for j in range(3)) - mp.hard_limit_alpha ** 2))
            pen += mu * np.abs(D[k]).sum()
        reg = sum(1e-4 * (0.5 * dt if k in (0, N - 1) else dt) * (D[k] @ D[k]) for k in range(N))
        assert abs(r["obj"] - (cost + pen + reg)) <= 1e-5 * max(1.0, abs(r["obj"])), (r["obj"], cost, pen, reg)
    o = go.OracleTrajOpt(go.ASTROBEE_SE3_MANIFOLD, 50, boxes=boxes, spheres=spheres)
    tp = o.tp
    assert (tp.s0, tp.xtol, tp.mu0, tp.k) == (10.0, 0.01, 1.0, 5.0)                       # astrobee_se3_manifold.jl:56-70
    for b in range(3):
        o.set_problem(x0[b], glo[b], ghi[b], tf[b])
        R = o.solve_trajopt(125)
        S = R["solves"]
        assert S >= 1 and R["stop_reason"] in (0, 1) and len(R["s_vec"]) == S + 1
        for i in range(S):
            assert R["s_vec"][i + 1] == (tp.tau_plus if R["rho_vec"][i + 1] > tp.c else tp.tau_minus) * R["s_vec"][i]
        # (|q_k| - 1 is the SECOND-order term of the last accepted step: every subproblem holds the linearised norm row exactly --
        # round 6: an equality row, see test_manifold_subproblem... above -- and this model has no trust-region row, so the steps
        # of a run that ends at its iteration limit are not small; measured 1.8e-3 .. 7e-2 on these problems)
        assert np.abs(np.linalg.norm(R["X"][:, 6:10], axis=1) - 1.0).max() < 0.15
