"""Indirect shooting seeded by the SCP dual (src/shooting.jl, src/traj_opt.jl:4-45), DubinsCar.
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
            assert np.abs(r["X"][b] - ro["X"]).max() < 1e-5 and np.abs(r["U"][b] - ro["U"]).max() < 1e-5
            assert np.abs(r["X"][b, -1] - glo[b]).max() <= 1e-3          # ftol of shooting.jl:14
    assert n_opt > B // 2
    with pytest.raises(g.GustoError):                # models without a shooting ODE are refused, not approximated
        f = g.BatchSolver(g.FREEFLYER_SE2, 20, 1, boxes=P.freeflyer_env())
        f.set_problems(P.FREEFLYER_X_INIT[None], P.FREEFLYER_X_GOAL[None], P.FREEFLYER_X_GOAL[None], [200.0])
        f.shoot()


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
