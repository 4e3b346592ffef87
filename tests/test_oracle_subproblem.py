"""Pins the C oracle's convex subproblem against independent solvers:
  (alpha) closed form: no obstacles, inactive rows -> equality-constrained QP = dense KKT solve with numpy
  (ii)    scipy SLSQP on the full slack formulation (N <= 20), independent algorithm
  (delta) optimality certificate: no feasible perturbation lowers the exact-penalty objective
The reference itself (Julia + JuMP + Ipopt) cannot run here: "parity unpinned" against it."""
import numpy as np
import pytest
import scipy.optimize as so

import gusto_oracle as go
import np_ref
import gusto_jl_amd as g

P = g.problems


def _freeflyer_oracle(N, env):
    o = go.Oracle(go.FREEFLYER_SE2, N, boxes=env)
    o.set_problem(P.FREEFLYER_X_INIT, P.FREEFLYER_X_GOAL, P.FREEFLYER_X_GOAL, P.FREEFLYER_TF)
    return o


def test_equality_constrained_qp_closed_form():
    """BlankEnv, Delta so large that the trust region is inactive: the subproblem is min sum w|u|^2 s.t. linear
    rows (freeflyer dynamics are exactly linear), i.e. the dense KKT system solved by numpy.linalg.solve."""
    N = 30
    o = _freeflyer_oracle(N, None)
    Xp, Up = o.init_straightline()
    dt = P.FREEFLYER_TF / (N - 1)
    r = o.subproblem(Xp, Up, 1e6, 1.0, 1e6 / 8 + 0.05)
    assert r["status"] == 1
    n, m = 6, 3
    E, e = np_ref.subproblem_matrices(o, Xp, Up, P.FREEFLYER_X_INIT, np.arange(6), P.FREEFLYER_X_GOAL, dt)
    w = np.full(N, dt); w[0] = w[-1] = 0.5 * dt
    H = np.zeros(((n + m) * N,) * 2)
    for k in range(N):
        for j in range(m):
            H[(n + m) * k + n + j, (n + m) * k + n + j] = 2 * w[k]
    K = np.block([[H, E.T], [E, np.zeros((E.shape[0],) * 2)]])
    sol = np.linalg.solve(K, np.concatenate([np.zeros(H.shape[0]), e]))
    z = sol[:H.shape[0]].reshape(N, n + m)
    # velocity rows may be active in the exact solution; check they are not before comparing
    v2 = z[:, 3] ** 2 + z[:, 4] ** 2
    assert v2.max() < 0.2 ** 2
    assert np.abs(r["X"] - z[:, :n]).max() < 1e-7
    assert np.abs(r["U"] - z[:, n:]).max() < 1e-8
    # the terminal-impulse quirk: u_N is a free variable and is not zero
    assert np.abs(z[-1, n:]).max() > 1e-3
    # the closed form of rho (a15): dynamics numerator = sum ||B (u_k - up_k)||, no obstacle terms
    mp = go.default_params(go.FREEFLYER_SE2)[1]
    Bm = np.zeros((6, 3)); Bm[3, 0] = Bm[4, 1] = 1 / mp.mass; Bm[5, 2] = 1 / mp.Jdiag[2]
    num = sum(np.linalg.norm(Bm @ (r["U"][k] - Up[k])) for k in range(N - 1))
    A = np.kron(np.array([[0, 1], [0, 0.]]), np.eye(3))
    den = sum(np.linalg.norm(Bm @ Up[k] + A @ Xp[k] + A @ (r["X"][k] - Xp[k])) for k in range(N - 1))
    assert abs(o.trust_region_ratio(r["X"], r["U"], Xp, Up) - num / den) < 1e-10


@pytest.mark.parametrize("N,omega,Delta", [(10, 1.0, 3.0), (16, 10.0, 0.5), (20, 1.0, 0.05)])
def test_against_slsqp(N, omega, Delta):
    """Full slack formulation solved by scipy SLSQP with analytic Jacobians."""
    env = P.freeflyer_env()
    o = _freeflyer_oracle(N, env)
    Xp, Up = o.init_straightline()
    dt = P.FREEFLYER_TF / (N - 1)
    toggle = Delta / 8 + 0.05
    r = o.subproblem(Xp, Up, Delta, omega, toggle)
    assert r["status"] == 1
    rows = o.rows()
    n, m = 6, 3
    nzN = (n + m) * N
    pen = [q for q in rows if q["kind"] in (1, 2, 3)]
    hard = [q for q in rows if q["kind"] in (0, 4)]
    E, e = np_ref.subproblem_matrices(o, Xp, Up, P.FREEFLYER_X_INIT, np.arange(6), P.FREEFLYER_X_GOAL, dt)
    kappa = 1.0 / max(1.0, omega)
    w = np.full(N, dt); w[0] = w[-1] = 0.5 * dt

    def split(z):
        Z = z[:nzN].reshape(N, n + m)
        return Z[:, :n], Z[:, n:], z[nzN:]

    def obj(z):
        X, U, s = split(z)
        return kappa * np.sum(w[:, None] * U ** 2) + s.sum()

    def obj_grad(z):
        X, U, s = split(z)
        gz = np.zeros((N, n + m)); gz[:, n:] = 2 * kappa * w[:, None] * U
        return np.concatenate([gz.ravel(), np.ones(len(pen))])

    def ineq(z):      # >= 0
        X, U, s = split(z)
        out = []
        for j, q in enumerate(pen):
            val, _ = np_ref.row_value_grad(q, (U if q["isu"] else X)[q["k"]])
            out.append(s[j] - (q["mul"] * val - q["off"]))
        for q in hard:
            val, _ = np_ref.row_value_grad(q, (U if q["isu"] else X)[q["k"]])
            out.append(-(q["mul"] * val - q["off"]))
        return np.array(out)

    def ineq_jac(z):
        X, U, s = split(z)
        Jm = np.zeros((len(pen) + len(hard), len(z)))
        for j, q in enumerate(pen + hard):
            _, gr = np_ref.row_value_grad(q, (U if q["isu"] else X)[q["k"]])
            off = (n + m) * q["k"] + (n if q["isu"] else 0)
            Jm[j, off:off + len(gr)] = -q["mul"] * gr
            if j < len(pen):
                Jm[j, nzN + j] = 1.0
        return Jm

    z0 = np.concatenate([np.hstack([r["X"], r["U"]]).ravel() * 0 + np.hstack([Xp, Up]).ravel(), np.ones(len(pen))])
    Epad = np.hstack([E, np.zeros((E.shape[0], len(pen)))])
    res = so.minimize(obj, z0, jac=obj_grad, method="SLSQP",
                      constraints=[{"type": "eq", "fun": lambda z: Epad @ z - e, "jac": lambda z: Epad},
                                   {"type": "ineq", "fun": ineq, "jac": ineq_jac}],
                      bounds=[(None, None)] * nzN + [(0, None)] * len(pen),
                      options={"ftol": 1e-15, "maxiter": 500})
    Xs, Us, ss = split(res.x)
    assert abs(res.fun / kappa - r["obj"]) <= 1e-6 * max(1.0, abs(r["obj"])), (res.fun / kappa, r["obj"], res.message)
    assert np.abs(Xs - r["X"]).max() < 2e-4 and np.abs(Us - r["U"]).max() < 2e-4


def test_no_feasible_descent_direction():
    """delta-certificate: projected random perturbations that keep the linear rows never lower the exact-penalty
    objective of the oracle's solution (and raise it to second order)."""
    N = 24
    env = P.freeflyer_env()
    o = _freeflyer_oracle(N, env)
    Xp, Up = o.init_straightline()
    dt = P.FREEFLYER_TF / (N - 1)
    omega, Delta = 10.0, 1.0
    r = o.subproblem(Xp, Up, Delta, omega, Delta / 8 + 0.05)
    rows = o.rows()
    kappa = 1.0 / omega
    E, e = np_ref.subproblem_matrices(o, Xp, Up, P.FREEFLYER_X_INIT, np.arange(6), P.FREEFLYER_X_GOAL, dt)
    z = np.hstack([r["X"], r["U"]]).ravel()
    assert np.abs(E @ z - e).max() < 1e-7
    J0 = np_ref.exact_penalty_objective(o, rows, r["X"], r["U"], dt, kappa)
    assert abs(J0 / kappa - r["obj"]) < 1e-6 * max(1, r["obj"])
    _, _, Vt = np.linalg.svd(E)
    Z = Vt[E.shape[0]:].T                      # null space of the linear rows
    rng = np.random.default_rng(1)
    hard = [q for q in rows if q["kind"] in (0, 4)]
    for scale in (1e-3, 1e-4):
        for _ in range(40):
            d = Z @ rng.standard_normal(Z.shape[1])
            zz = (z + scale * d / np.linalg.norm(d)).reshape(N, 9)
            X, U = zz[:, :6], zz[:, 6:]
            if any(q["mul"] * np_ref.row_value_grad(q, (U if q["isu"] else X)[q["k"]])[0] - q["off"] > 0 for q in hard):
                continue
            assert np_ref.exact_penalty_objective(o, rows, X, U, dt, kappa) >= J0 - 1e-9


@pytest.mark.parametrize("model", [go.FREEFLYER_SE2, go.DUBINS_CAR, go.ASTROBEE_SE3, go.ASTROBEE_SE3_MANIFOLD])
def test_jacobians_by_central_differences(model):
    """gamma: A = df/dx and B = df/du of every model against central differences of f (update_A!, B_dyn)."""
    n, m = go.MODEL_DIMS[model]
    o = go.Oracle(model, 10)
    rng = np.random.default_rng(model)
    for _ in range(5):
        x, u = rng.uniform(-0.5, 0.5, n), rng.uniform(-0.5, 0.5, m)
        f, A, B = o.dynamics(x, u)
        h = 1e-6
        for j in range(n):
            d = np.zeros(n); d[j] = h
            fd = (o.dynamics(x + d, u)[0] - o.dynamics(x - d, u)[0]) / (2 * h)
            assert np.abs(fd - A[:, j]).max() < 1e-8, (model, j)
        for j in range(m):
            d = np.zeros(m); d[j] = h
            fd = (o.dynamics(x, u + d)[0] - o.dynamics(x, u - d)[0]) / (2 * h)
            assert np.abs(fd - B[:, j]).max() < 1e-8, (model, j)


def test_signed_distance_properties():
    """Analytic disc-vs-AABB distance: sign, unit normals, 1-Lipschitz, linearisation exactness outside corners."""
    env = P.freeflyer_env()
    o = go.Oracle(go.FREEFLYER_SE2, 10, boxes=env)
    rng = np.random.default_rng(0)
    for _ in range(300):
        p, q = rng.uniform(-0.5, 4.0, 2), rng.uniform(-0.5, 4.0, 2)
        i = int(rng.integers(len(env)))
        d1, n1 = o.signed_distance(0, p, i)
        d2, _ = o.signed_distance(0, q, i)
        assert abs(np.linalg.norm(n1[:2]) - 1) < 1e-12
        assert abs(d1 - d2) <= np.linalg.norm(p - q) + 1e-12
        inside = (env[i, 0] <= p[0] <= env[i, 3]) and (env[i, 1] <= p[1] <= env[i, 4])
        assert (d1 < -0.157 + 1e-12) == inside or not inside
    # known value: 1 m left of the x<=0 wall slab ... body radius 0.157
    d, nh = o.signed_distance(0, np.array([1.0, 1.0]), 1)
    assert abs(d - (1.0 - 0.157)) < 1e-12 and np.allclose(nh[:2], [1.0, 0.0])
    # arm component: offset (0, 0.15)
    d_arm, _ = o.signed_distance(1, np.array([1.0, 1.0]), 3)      # y <= 0 slab
    assert abs(d_arm - (1.15 - 0.157)) < 1e-12


def test_reference_formulas_of_the_outer_loop_pieces():
    """cost_true (freeflyer_se2.jl:66-76) and convergence_metric (traj_opt.jl:74-85) against direct numpy."""
    N = 12
    o = _freeflyer_oracle(N, None)
    rng = np.random.default_rng(3)
    X, Xq, U = rng.standard_normal((N, 6)), rng.standard_normal((N, 6)), rng.standard_normal((N, 3))
    dt = P.FREEFLYER_TF / (N - 1)
    J = sum(0.5 * dt * (U[k - 1, j] ** 2 + U[k, j] ** 2) for k in range(1, N) for j in range(3))
    assert abs(o.cost_true(U) - J) < 1e-10 * J
    cm = np.linalg.norm(X - Xq, axis=1).max() / np.linalg.norm(X, axis=1).max()
    assert abs(o.convergence_metric(X, Xq) - cm) < 1e-14
    Xs, Us = o.init_straightline()
    t = np.arange(N) / (N - 1)
    assert np.abs(Xs - ((1 - t)[:, None] * P.FREEFLYER_X_INIT + t[:, None] * P.FREEFLYER_X_GOAL)).max() < 1e-15
    assert np.all(Us == 0)
