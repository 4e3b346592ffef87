"""Indirect shooting seeded by the SCP dual (src/shooting.jl, src/traj_opt.jl:4-45): DubinsCar and
AstrobeeSE3Manifold, the two models with a shooting ODE in the reference.
CPU: the oracle's restatement against scipy.  GPU: gusto_shoot against the oracle, and solve_SCPshooting! end to end."""
import numpy as np
import pytest

import gusto_oracle as go
import gusto_jl_amd as g

P = g.problems


def _scipy_shoot(x0, xg, tf, p0):
    """independent integrator (adaptive RK45 at 1e-12) and root finder (MINPACK hybr) for the same boundary value problem"""
    from scipy.integrate import solve_ivp
    from scipy.optimize import root

    def ode(t, z):
        th, px, py, pth = z[2], z[3], z[4], z[5]
        return [2.0 * np.cos(th), 2.0 * np.sin(th), 0.5 * pth, 0.0, 0.0, px * 2.0 * np.sin(th) - py * 2.0 * np.cos(th)]

    def F(p):
        return xg - solve_ivp(ode, (0, tf), np.concatenate([x0, p]), rtol=1e-12, atol=1e-14).y[:3, -1]

    return root(F, p0, method="hybr", tol=1e-12)


@pytest.mark.parametrize("b", [0, 2, 3, 9])
def test_oracle_shooting_against_scipy(b):
    x0, glo, ghi, tf = P.dubins_batch(12)
    o = go.Oracle(go.DUBINS_CAR, 30)
    o.set_problem(x0[b], glo[b], ghi[b], tf[b])
    r = o.solve(30)
    s = o.shoot(substeps=16, ftol=1e-11)             # seed: the dual of the last SCP subproblem
    assert s["status"] == 1 and s["newton_iters"] <= 6 and s["resid"] < 1e-11
    assert np.abs(s["p0"] - r["dual"]).max() < 0.05 * np.abs(r["dual"]).max()     # the SCP dual IS the initial costate, to a few per cent
    assert np.abs(s["X"][0] - x0[b]).max() == 0 and np.abs(s["X"][-1] - glo[b]).max() < 1e-10
    assert np.abs(s["U"][:, 0] - 0.5 * 1.0 * (s["U"][:, 0] * 2)).max() < 1e-15
    sol = _scipy_shoot(x0[b], glo[b], tf[b], r["dual"])
    assert sol.success and np.abs(sol.x - s["p0"]).max() < 1e-6       # RK4 at 16 substeps against RK45 at 1e-12
    assert o.cost_true(s["U"]) < r["J_true"][-1]                       # the extremal is cheaper than the SCP's trapezoid optimum


def test_oracle_shooting_diverges_from_a_useless_seed_and_refuses_other_models():
    x0, glo, ghi, tf = P.dubins_batch(12)
    o = go.Oracle(go.DUBINS_CAR, 30)
    o.set_problem(x0[1], glo[1], ghi[1], tf[1])
    o.solve(30)                                      # subproblem failed: the dual is garbage (1e6)
    s = o.shoot()
    assert s["status"] == 0 and s["resid"] > 1.0
    f = go.Oracle(go.FREEFLYER_SE2, 20, boxes=P.freeflyer_env())
    f.set_problem(P.FREEFLYER_X_INIT, P.FREEFLYER_X_GOAL, P.FREEFLYER_X_GOAL, 200.0)
    assert f.shoot()["status"] == -1


def _manifold_ode(mass, J):
    """independent restatement of dynamics_shooting! + get_control (astrobee_se3_manifold.jl:831-895) for scipy"""
    J = np.asarray(J, dtype=float)

    def ode(t, z):
        v, q, w = z[3:6], z[6:10], z[10:13]
        pr, pv, pq, pw = z[13:16], z[16:19], z[19:23], z[23:26]
        F, M = pv / (2 * mass), pw / J / 2
        qw, qx, qy, qz = q
        wx, wy, wz = w
        pqw, pqx, pqy, pqz = pq
        d = np.zeros(26)
        d[0:3] = v
        d[3:6] = F / mass
        d[6] = 0.5 * (-wx * qx - wy * qy - wz * qz)
        d[7] = 0.5 * (wx * qw - wz * qy + wy * qz)
        d[8] = 0.5 * (wy * qw + wz * qx - wx * qz)
        d[9] = 0.5 * (wz * qw - wy * qx + wx * qy)
        d[10:13] = (M - np.cross(w, J * w)) / J
        d[16:19] = -pr
        d[19] = -0.5 * (pqx * wx + pqy * wy + pqz * wz)
        d[20] = -0.5 * (-pqw * wx + pqy * wz - pqz * wy)
        d[21] = -0.5 * (-pqw * wy - pqx * wz + pqz * wx)
        d[22] = -0.5 * (-pqw * wz + pqx * wy - pqy * wx)
        d[23] = -0.5 * (-pqw * qx + pqx * qw - pqy * qz + pqz * qy)
        d[24] = -0.5 * (-pqw * qy + pqx * qz + pqy * qw - pqz * qx)
        d[25] = -0.5 * (-pqw * qz - pqx * qy + pqy * qx + pqz * qw)
        return d
    return ode


@pytest.mark.parametrize("b", [0, 1, 3])
def test_oracle_manifold_shooting_against_scipy(b):
    """The 26-dimensional state + costate ODE: the oracle's RK4 extremal from its converged costate against an adaptive
    integration (RK45 at 1e-12) of an independent restatement, knot by knot; the boundary condition at shooting.jl's ftol."""
    from scipy.integrate import solve_ivp
    N = 50
    x0, glo, ghi, tf = P.astrobee_manifold_batch(8)
    bx, sp = P.iss_corner_env(True)
    o = go.Oracle(go.ASTROBEE_SE3_MANIFOLD, N, boxes=bx, spheres=sp)
    o.set_problem(x0[b], glo[b], ghi[b], tf[b])
    r = o.solve(30)
    s = o.shoot()                                    # shooting.jl:14: ftol = 1e-3, seeded by SCPS.dual
    assert r["converged"] and s["status"] == 1 and s["newton_iters"] <= 3 and s["resid"] <= 1e-3
    mass, J = o.mp.mass, list(o.mp.Jdiag)           # gusto_default_params: astrobee_se3_manifold.jl's robot constants
    sol = solve_ivp(_manifold_ode(mass, J), (0, tf[b]), np.concatenate([x0[b], s["p0"]]), rtol=1e-12, atol=1e-14,
                    t_eval=np.linspace(0, tf[b], N))
    assert sol.success
    assert np.abs(sol.y[:13].T - s["X"]).max() < 1e-6                      # RK4, 4 substeps per knot interval
    U = np.concatenate([sol.y[16:19] / (2 * mass), sol.y[23:26] / np.asarray(J)[:, None] / 2]).T
    assert np.abs(U - s["U"]).max() < 1e-7
    assert np.abs(s["X"][0] - x0[b]).max() == 0 and np.abs(s["X"][-1] - glo[b]).max() <= 1e-3
    assert abs(np.linalg.norm(s["X"][-1, 6:10]) - 1.0) < 1e-6               # the flow keeps the quaternion on the sphere
    # the SCP dual (trapezoid rule, obstacle rows) seeds Newton within its basin -- <= 3 steps above -- and no component of
    # the costate runs away along the unobservable direction p_q || q (the step leaves it untouched)
    assert np.abs(s["p0"] - r["dual"]).max() < 1.0 * np.abs(r["dual"]).max()


@pytest.mark.gpu
def test_gpu_shooting_matches_the_oracle():
    B = 256
    x0, glo, ghi, tf = P.dubins_batch(B)
    s = g.BatchSolver(g.DUBINS_CAR, 30, B, hist_cap=40)
    s.set_problems(x0, glo, ghi, tf)
    s.solve(30)
    duals = s.dual()
    r = s.shoot()                                    # seeds = SCPS.dual on device
    o = go.Oracle(go.DUBINS_CAR, 30)
    n_opt = 0
    for b in range(B):
        o.set_problem(x0[b], glo[b], ghi[b], tf[b])
        ro = o.shoot(p0=duals[b])
        assert int(r["status"][b]) == ro["status"], b
        if ro["status"] == 1:        # (a diverging Newton run on a garbage seed need not take the same path on both sides)
            n_opt += 1
            assert int(r["newton_iters"][b]) == ro["newton_iters"], b
            # same scheme on both sides; the forward-difference Jacobian (h = 1e-6) turns 1e-16 differences of sin/cos
            # into 1e-9 relative differences of the Newton step, which an ill-conditioned Jacobian amplifies: 1e-5
            assert np.abs(r["p0"][b] - ro["p0"]).max() < 1e-5 * max(1.0, np.abs(ro["p0"]).max())
            assert np.abs(r["X"][b] - ro["X"]).max() < 1e-4 and np.abs(r["U"][b] - ro["U"]).max() < 1e-4   # (both sides stop at ftol = 1e-3 from seeds that agree to 1e-9: measured 4.7e-5)
            assert np.abs(r["X"][b, -1] - glo[b]).max() <= 1e-3          # ftol of shooting.jl:14
    assert n_opt > B // 2
    with pytest.raises(g.GustoError):                # models without a shooting ODE are refused, not approximated
        f = g.BatchSolver(g.FREEFLYER_SE2, 20, 1, boxes=P.freeflyer_env())
        f.set_problems(P.FREEFLYER_X_INIT[None], P.FREEFLYER_X_GOAL[None], P.FREEFLYER_X_GOAL[None], [200.0])
        f.shoot()


@pytest.mark.gpu
def test_group_pass_of_the_stragglers_changes_nothing(monkeypatch):
    """gusto_shoot runs the problems that are still iterating after 8 Newton steps with 16 lanes each (a lane per Jacobian
    column / line-search candidate).  Same integrations, same acceptance rule: every output equals, bit for bit, what the
    lane-per-problem pass alone produces -- for the problems that converge late and for the ones that never do."""
    B = 4096
    s = g.BatchSolver(g.DUBINS_CAR, 30, B, hist_cap=40)
    s.set_problems(*P.dubins_batch(B))
    s.solve(3)                                       # a few SCP iterations: a mediocre seed, many long Newton runs
    two = s.shoot()
    one = s.shoot(group_pass=False)
    late = two["newton_iters"] > 8
    assert late.sum() > B // 50 and (two["status"][late] == 1).any() and (two["status"][late] == 0).any()
    for key in ("status", "newton_iters", "resid", "p0"):
        assert np.array_equal(two[key], one[key], equal_nan=(key in ("resid", "p0"))), key
    ok = two["status"] == 1
    assert np.array_equal(two["X"][ok], one["X"][ok]) and np.array_equal(two["U"][ok], one["U"][ok])


@pytest.mark.gpu
def test_gpu_manifold_shooting_matches_the_oracle():
    B, N = 48, 50
    x0, glo, ghi, tf = P.astrobee_manifold_batch(B)
    bx, sp = P.iss_corner_env(True)
    s = g.BatchSolver(g.ASTROBEE_SE3_MANIFOLD, N, B, hist_cap=40, boxes=bx, spheres=sp)
    s.set_problems(x0, glo, ghi, tf)
    s.solve(30)
    duals = s.dual()
    r = s.shoot()                                    # seeds = SCPS.dual on device
    o = go.Oracle(go.ASTROBEE_SE3_MANIFOLD, N, boxes=bx, spheres=sp)
    n_opt = 0
    for b in range(B):
        o.set_problem(x0[b], glo[b], ghi[b], tf[b])
        ro = o.shoot(p0=duals[b])
        assert int(r["status"][b]) == ro["status"], b
        if ro["status"] == 1:
            n_opt += 1
            assert int(r["newton_iters"][b]) == ro["newton_iters"], b
            # (13 x 13 finite-difference Jacobian, rank deficient along the quaternion norm: see DESIGN.md 7)
            assert np.abs(r["X"][b] - ro["X"]).max() < 1e-4 and np.abs(r["U"][b] - ro["U"]).max() < 1e-4   # (both sides stop at ftol = 1e-3 from seeds that agree to 1e-9: measured 4.7e-5)
            assert np.abs(r["X"][b, -1] - glo[b]).max() <= 1e-3          # ftol of shooting.jl:14
            assert r["X"][b].shape == (N, 13) and r["U"][b].shape == (N, 6)
    assert n_opt > B // 2


@pytest.mark.gpu
def test_solve_SCPshooting_through_the_seam():
    """solve_SCPshooting!(TOS, TOP, solve_gusto_hip!, init_traj_straightline, "hip") (traj_opt.jl:4-45)."""
    H = g.host
    model = H.DubinsCar()
    x0, glo, ghi, tf = P.dubins_batch(12)
    done = 0
    for b in (0, 2, 3, 9):
        gs = H.GoalSet()
        H.add_goal(gs, H.Goal(H.PointGoal(glo[b]), tf[b], model))
        TOP = H.TrajectoryOptimizationProblem(H.ProblemDefinition(H.Robot(), model, H.BlankEnv(), x0[b], gs), 30, tf[b], True)
        TOS = H.TrajectoryOptimizationSolution(TOP)
        H.solve_SCPshooting(TOS, TOP, H.solve_gusto_hip, H.init_traj_straightline, "hip", max_iter=30)
        SS, SCPS = TOS.SS, TOS.SCPS
        assert len(SS.prob_status) == len(SS.convergence_measure) and SS.prob_status[0] == "NA"
        if SS.converged:       # two consecutive successful shooting runs agreed: the extremal is the answer
            done += 1
            assert SS.prob_status[-1] == SS.prob_status[-2] == "Optimal"
            assert SCPS.iterations < 30 and TOS.traj.X.shape == (3, 30)
            assert np.abs(TOS.traj.X[:, 0] - x0[b]).max() < 1e-12 and np.abs(TOS.traj.X[:, -1] - glo[b]).max() <= 1e-3
            assert np.isfinite(SS.J_true[-1])
        else:
            assert np.array_equal(TOS.traj.X, SCPS.traj.X)
    assert done >= 2


@pytest.mark.gpu
def test_solve_SCPshooting_manifold_through_the_seam():
    """The same driver for AstrobeeSE3Manifold in the ISS corner (26-dimensional shooting ODE, 13 x 13 Newton system)."""
    H = g.host
    model = H.AstrobeeSE3Manifold()
    x0, glo, ghi, tf = P.astrobee_manifold_batch(4)
    for b in (0, 1):
        gs = H.GoalSet()
        H.add_goal(gs, H.Goal(H.PointGoal(glo[b]), tf[b], model))
        TOP = H.TrajectoryOptimizationProblem(H.ProblemDefinition(H.Robot(), model, H.ISSCorner(True), x0[b], gs), 50, tf[b], True)
        TOS = H.TrajectoryOptimizationSolution(TOP)
        H.solve_SCPshooting(TOS, TOP, H.solve_gusto_hip, H.init_traj_straightline, "hip", max_iter=30)
        SS, SCPS = TOS.SS, TOS.SCPS
        assert len(SS.prob_status) == len(SS.convergence_measure) >= 2 and SS.prob_status[0] == "NA"
        assert "Optimal" in SS.prob_status          # the SCP dual seeds a converging Newton run
        assert TOS.traj.X.shape == (13, 50) and TOS.traj.U.shape == (6, 50)
        assert np.abs(TOS.traj.X[:, 0] - x0[b]).max() < 1e-12 and np.abs(TOS.traj.X[:, -1] - glo[b]).max() <= 1e-3
        if SS.converged:
            assert SS.prob_status[-1] == SS.prob_status[-2] == "Optimal" and np.isfinite(SS.J_true[-1])
        else:
            assert np.array_equal(TOS.traj.X, SCPS.traj.X)


def _shooting_tops(model_name, idx):
    H = g.host
    if model_name == "dubins":
        model = H.DubinsCar()
        x0, glo, ghi, tf = P.dubins_batch(max(idx) + 1)
        env, N = H.BlankEnv, 30
    else:
        model = H.AstrobeeSE3Manifold()
        x0, glo, ghi, tf = P.astrobee_manifold_batch(max(idx) + 1)
        env, N = (lambda: H.ISSCorner(True)), 50
    tops = []
    for b in idx:
        gs = H.GoalSet()
        H.add_goal(gs, H.Goal(H.PointGoal(glo[b]), tf[b], model))
        tops.append(H.TrajectoryOptimizationProblem(H.ProblemDefinition(H.Robot(), model, env(), x0[b], gs), N, tf[b], True))
    return tops


@pytest.mark.gpu
@pytest.mark.parametrize("model_name,idx", [("dubins", list(range(24))), ("manifold", [0, 1, 2, 3])])
def test_solve_SCPshooting_batch_equals_the_single_problem_driver(model_name, idx):
    """solve_SCPshooting_batch! (one handle, one gusto_shoot and one gusto_solve(1) per round over the live problems,
    gusto_set_active) against solve_SCPshooting! problem by problem (traj_opt.jl:4-45): every problem leaves its loop in
    the same round for the same reason with the same SCPSolution / ShootingSolution, bit for bit."""
    H = g.host
    tops = _shooting_tops(model_name, idx)
    ones = []
    for t in tops:
        TOS = H.TrajectoryOptimizationSolution(t)
        H.solve_SCPshooting(TOS, t, H.solve_gusto_hip, H.init_traj_straightline, "hip", max_iter=30)
        ones.append(TOS)
    TOSs = [H.TrajectoryOptimizationSolution(t) for t in tops]
    H.solve_SCPshooting_batch(TOSs, tops, H.solve_gusto_hip, H.init_traj_straightline, "hip", max_iter=30)
    exits = set()
    for a, b in zip(ones, TOSs):
        assert a.SS.converged == b.SS.converged and a.SS.prob_status == b.SS.prob_status
        assert np.array_equal(a.SS.convergence_measure, b.SS.convergence_measure, equal_nan=True)
        assert np.array_equal(a.SS.J_true, b.SS.J_true, equal_nan=True)
        assert a.SCPS.iterations == b.SCPS.iterations and a.SCPS.converged == b.SCPS.converged
        assert a.SCPS.scp_status == b.SCPS.scp_status and a.SCPS.J_true == b.SCPS.J_true
        assert a.SCPS.SCPP.Delta_vec == b.SCPS.SCPP.Delta_vec and a.SCPS.SCPP.omega_vec == b.SCPS.SCPP.omega_vec
        assert np.array_equal(a.SCPS.traj.X, b.SCPS.traj.X) and np.array_equal(a.SCPS.dual, b.SCPS.dual)
        assert np.array_equal(a.traj.X, b.traj.X) and np.array_equal(a.traj.U, b.traj.U)
        exits.add("shooting" if a.SS.converged else ("scp" if a.SCPS.converged else "other"))
    print(model_name, "exits", exits, "iterations", [t.SCPS.iterations for t in TOSs])
    if model_name == "dubins":
        assert {"shooting", "scp"} <= exits or {"shooting", "other"} <= exits      # the batch holds problems that leave differently


@pytest.mark.gpu
def test_inactive_problems_are_left_untouched():
    """gusto_set_active: a solve / shoot over a subset changes nothing of the others (trajectory, histories, counters), and
    the subset gets what it gets in a batch of its own."""
    B = 64
    x0, glo, ghi, tf = P.freeflyer_batch(B)
    s = g.BatchSolver(g.FREEFLYER_SE2, 50, B, hist_cap=80, boxes=P.freeflyer_env())
    s.set_problems(x0, glo, ghi, tf)
    s.solve(2)
    X0, U0 = s.traj()
    h0, st0 = s.history(), s.status()
    act = np.zeros(B, bool)
    act[[1, 5, 17, 40, 63]] = True
    s.set_active(act)
    s.solve(3)
    X1, U1 = s.traj()
    h1, st1 = s.history(), s.status()
    off = ~act
    assert np.array_equal(X0[off], X1[off]) and np.array_equal(U0[off], U1[off])
    assert all(np.array_equal(h0[k][off], h1[k][off]) for k in h0) and all(np.array_equal(st0[k][off], st1[k][off]) for k in st0)
    assert (st1["iterations"][act] > st0["iterations"][act]).all()
    r = g.BatchSolver(g.FREEFLYER_SE2, 50, int(act.sum()), hist_cap=80, boxes=P.freeflyer_env())
    r.set_problems(x0[act], glo[act], ghi[act], tf[act])
    r.solve(2)
    r.solve(3)
    Xr, Ur = r.traj()
    assert np.array_equal(Xr, X1[act]) and np.array_equal(Ur, U1[act])
    s.set_active(None)                       # everything again
    s.solve(1)
    assert (s.status()["iterations"] >= st1["iterations"]).all() and (s.status()["iterations"][off] > st1["iterations"][off]).any()
    s.set_active(np.zeros(B, bool))          # nothing: a launch that hands out no problem leaves EVERYTHING as it was
    st2, (X2, U2), h2 = s.status(), s.traj(), s.history()
    s.solve(5)
    st3, (X3, U3), h3 = s.status(), s.traj(), s.history()
    assert np.array_equal(X2, X3) and np.array_equal(U2, U3)
    for key in st2:
        assert np.array_equal(st2[key], st3[key]), key
    for key in h2:
        assert np.array_equal(np.asarray(h2[key]), np.asarray(h3[key]), equal_nan=True), key
    lane = g.BatchSolver(g.DUBINS_CAR, 30, 8)
    lane.set_problems(*P.dubins_batch(8))
    try:
        lane.set_decomposition(2)            # (the lane-per-problem kernel: only in -DGUSTO_WITH_LANE builds)
    except g.GustoError:
        return
    lane.set_active(np.ones(8, bool))
    with pytest.raises(g.GustoError):
        lane.solve(2)
