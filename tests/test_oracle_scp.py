"""The oracle's GuSTO outer loop (scp_gusto.jl:49-176) against the committed golden vectors and the soft pin the
reference offers (the one recorded notebook run, examples/freeflyerSE2.ipynb:87-97)."""
import glob
import os

import numpy as np
import pytest

import gusto_oracle as go
import gusto_jl_amd as g

P = g.problems
GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.mark.parametrize("name", ["freeflyer_se2_n50", "dubins_car_n30", "astrobee_se3_n50", "astrobee_se3_manifold_n50"])
def test_oracle_reproduces_golden(name):
    d = _load(name)
    model, N = int(d["model"]), int(d["N"])
    o = go.Oracle(model, N, boxes=d["boxes"], spheres=d["spheres"])
    for b in range(len(d["x_init"])):
        o.set_problem(d["x_init"][b], d["goal_lo"][b], d["goal_hi"][b], d["tf"][b])
        r = o.solve(int(d["max_iter"]))
        assert r["iterations"] == d["iterations"][b]
        assert r["converged"] == bool(d["converged"][b]) and r["successful"] == bool(d["successful"][b])
        assert np.abs(r["X"] - d["X"][b]).max() < 1e-9 and np.abs(r["U"] - d["U"][b]).max() < 1e-9
        nh = len(r["omega"])
        np.testing.assert_array_equal(r["scp_status"], d["scp_status"][b, :nh].astype(int))
        np.testing.assert_allclose(r["omega"], d["omega"][b, :nh])
        np.testing.assert_allclose(r["Delta"], d["Delta"][b, :nh])
        np.testing.assert_allclose(r["J_true"], d["J_true"][b, :len(r["J_true"])], rtol=1e-9, atol=1e-12)


def test_history_bookkeeping_matches_reference_conventions():
    """SCPSolution / SCPParam_GuSTO initial entries and lengths (types.jl:233, scp_gusto.jl:21-23,73-75)."""
    o = go.Oracle(go.FREEFLYER_SE2, 50, boxes=P.freeflyer_env())
    o.set_problem(P.FREEFLYER_X_INIT, P.FREEFLYER_X_GOAL, P.FREEFLYER_X_GOAL, P.FREEFLYER_TF)
    r = o.solve(30)
    it = r["iterations"]
    assert r["converged"] and it > 2            # earliest possible convergence is iteration 3 (scp_gusto.jl:170)
    assert len(r["omega"]) == len(r["Delta"]) == len(r["accept"]) == len(r["scp_status"]) == len(r["conv"]) == it + 1
    assert len(r["J_true"]) == len(r["J_full"]) == it + 1
    assert len(r["rho"]) == 2 + int(np.sum(r["tr_sat"][1:]))      # [0.] + ratio(traj,traj) + one per TR-satisfied trip
    assert r["accept"][0] == 1 and r["scp_status"][0] == 0 and r["solver_status"][0] == 0 and r["conv"][0] == 0.0
    assert r["Delta"][0] == 3.0 and r["omega"][0] == 1.0 and r["J_true"][0] == 0.0     # straight line, U = 0
    # stop rule: sum of the last two convergence measures under the threshold after an accepted step
    assert r["conv"][-1] + r["conv"][-2] <= 1e-2 and r["accept"][-1] == 1
    # accepted steps copy the trajectory, rejected ones repeat J_true (scp_gusto.jl:149-154)
    for k in range(1, it + 1):
        if not r["accept"][k]:
            assert r["J_true"][k] == r["J_true"][k - 1]
    # Delta/omega updates (scp_gusto.jl:123-147)
    for k in range(1, it + 1):
        s = r["scp_status"][k]
        if s == 2:
            assert r["Delta"][k] == 0.5 * r["Delta"][k - 1] and r["omega"][k] == r["omega"][k - 1]
        if s in (3, 4):
            assert r["omega"][k] == 10.0 * r["omega"][k - 1]
        if s == 1:
            assert r["omega"][k] == r["omega"][k - 1] and r["Delta"][k] >= r["Delta"][k - 1]


def test_resume_semantics():
    """A second solve call continues from SCPS (iter_cap = iterations + max_iter, scp_gusto.jl:67)."""
    env = P.freeflyer_env()
    a = go.Oracle(go.FREEFLYER_SE2, 50, boxes=env)
    a.set_problem(P.FREEFLYER_X_INIT, P.FREEFLYER_X_GOAL, P.FREEFLYER_X_GOAL, P.FREEFLYER_TF)
    ra = a.solve(30)
    b = go.Oracle(go.FREEFLYER_SE2, 50, boxes=env)
    b.set_problem(P.FREEFLYER_X_INIT, P.FREEFLYER_X_GOAL, P.FREEFLYER_X_GOAL, P.FREEFLYER_TF)
    r1 = b.solve(4)
    assert r1["iterations"] == 4 and not r1["converged"]
    r2 = b.solve(26)
    assert r2["iterations"] == ra["iterations"]
    assert np.array_equal(r2["X"], ra["X"])
    assert len(r2["J_true"]) == len(ra["J_true"]) + 1        # one extra leading entry per call (scp_gusto.jl:73)


def _notebook_run(dist_model=None, max_iter=40):
    o = go.Oracle(go.FREEFLYER_SE2, 200, boxes=P.freeflyer_env())
    if dist_model:
        o.set_distance_model(**dist_model)
    o.set_problem(P.FREEFLYER_X_INIT, P.FREEFLYER_X_GOAL, P.FREEFLYER_X_GOAL, P.FREEFLYER_TF)
    return o, o.solve(max_iter)


def _trace():
    import json
    with open(os.path.join(GOLD, "freeflyer_se2_notebook_trace.json")) as f:
        t = json.load(f)
    code = {v: k for k, v in go.SCP_STATUS.items()}
    t["scp_status"] = [code[x] for x in t["scp_status"]]
    return t


def _head_matches(r, t, n=7):
    """entries 0..n-1 of the recorded histories: the run-up before the reference's first rejection"""
    assert list(r["accept"][:n]) == t["accept_solution"][:n]
    assert list(r["scp_status"][:n]) == t["scp_status"][:n]
    np.testing.assert_array_equal(r["Delta"][:n], t["Delta_vec"][:n])
    np.testing.assert_array_equal(r["omega"][:n], t["omega_vec"][:n])


def test_notebook_trace_head_analytic_distance():
    """The only run the reference records (examples/freeflyerSE2.ipynb:87-97, N=200, Gurobi + BulletCollision;
    committed as tests/golden/freeflyer_se2_notebook_trace.json) against the oracle with the product's analytic
    planar distance.  Iterations 1-6 (all accepted, Delta = 3, omega = 1): identical decisions; J_true within 4 %
    (measured -2.3 ... -3.8 %), convergence_measure[1] within 1 %.  DESIGN section 8 has the account."""
    t = _trace()
    _, r = _notebook_run(max_iter=8)
    _head_matches(r, t)
    J = np.array(t["J_true"])
    err = (r["J_true"][1:7] - J[1:7]) / J[1:7]
    assert np.abs(err).max() < 0.04 and (err < 0).all()     # uniformly a little cheaper than the recorded run
    assert abs(r["conv"][1] / t["convergence_measure"][1] - 1) < 0.01
    assert np.abs(r["conv"][2:4] / np.array(t["convergence_measure"][2:4]) - 1).max() < 0.06


@pytest.mark.parametrize("dm,tol", [
    (dict(n_poly=32), 0.013),                       # 32-gon prism, vertices on the circle, no margin
    (dict(margin=0.00125), 0.0125),                 # exact disc, every distance 1.25 mm smaller
    (dict(n_poly=64, margin=0.00075), 0.0065),      # best of the scan (scratch study)
])
def test_notebook_trace_head_distance_study(dm, tol):
    """One-parameter study of the unpinned external distance (BulletCollision: tessellated hull, margins unknown):
    sub-millimetre changes of the distance model move J_true[1..6] by whole per cents, and several physically
    plausible ones bring all six within ~1 % of the recorded values while keeping every recorded decision."""
    t = _trace()
    _, r = _notebook_run(dm, max_iter=8)
    _head_matches(r, t)
    J = np.array(t["J_true"])
    assert np.abs((r["J_true"][1:7] - J[1:7]) / J[1:7]).max() < tol


def test_notebook_trace_start_state_sits_on_the_acceptance_threshold():
    """Why the run is so sensitive: x_init = (0.2, 2.4) puts the body 0.043 m from the x = 0 table slab, i.e. the
    k = 1 obstacle row is violated by 0.007 for ever (x_1 is fixed) with eps = 0.01 as the acceptance threshold
    (scp_gusto.jl:321).  3 mm less distance and every iterate 'violates constraints': omega escalates to omega_max."""
    o, _ = _notebook_run(max_iter=1)
    d, _n = o.signed_distance(0, P.FREEFLYER_X_INIT[:2], 1)
    assert abs(d - 0.043) < 1e-12 and 0 < 0.05 - d < 0.01
    _, r = _notebook_run(dict(margin=0.004), max_iter=8)
    assert (r["scp_status"][1:] == 3).all() and r["omega"][-1] >= 1e5


def test_notebook_trace_tail_needs_a_discontinuous_distance():
    """Entries 7-9 and 13-19 of the recorded run are :InaccurateModel, i.e. rho > rho1 = 0.3, for steps of ~1 % of
    the trajectory scale.  rho's denominator (freeflyer_se2.jl:403-421) sums |clearance - d - n.(r - r0)| over
    200 knots x 2 robot components x 14 obstacles ~ 6e3 m, so rho > 0.3 needs ~2e3 m of linearisation error:
    no distance function with bounded error can produce it (ours: rho < 1e-2 over the whole run, with or without
    tessellation), and returning 'no result' sentinels for penetrating / touching pairs breaks iteration 1, which
    the reference accepted.  The tail of the recorded trace is therefore not reproducible from the repository
    (DESIGN section 8); what we pin is that the oracle's own run converges with omega = 1."""
    t = _trace()
    for dm in (None, dict(n_poly=25), dict(n_poly=32)):
        o, r = _notebook_run(dm)
        assert r["converged"] and r["omega"][-1] == 1.0 and (r["scp_status"][1:] == 1).all()
        assert np.nanmax(r["rho"][2:]) < 1e-2
        Xp, Up = o.traj()
        den = sum(abs(0.05 - o.signed_distance(c, Xp[k, :2], i)[0]) for k in range(200) for c in range(2)
                  for i in range(14))
        assert den > 5e3
    assert t["scp_status"].count(2) == 10 and max(t["omega_vec"]) == 100.0
    for pen_mode, kw in ((1, {}), (2, {}), (3, dict(pen_band=1e-3))):        # sentinel hypotheses
        _, r = _notebook_run(dict(pen_mode=pen_mode, pen_value=1e18, **kw), max_iter=2)
        assert r["accept"][1] == 0                                           # the reference accepted iteration 1


def test_infeasible_subproblem_is_reported_not_raised():
    """dubins with a heading along the straight line: the x goal row is uncontrollable in the linearisation, the
    subproblem is infeasible and the loop stops with SubproblemFailed (scp_gusto.jl:106-111)."""
    x0, glo, ghi, tf = P.dubins_batch(6)
    o = go.Oracle(go.DUBINS_CAR, 30)
    o.set_problem(x0[4], glo[4], ghi[4], tf[4])
    r = o.solve(30)
    assert r["stop_reason"] == 2 and r["iterations"] == 0 and not r["converged"]
    assert np.isfinite(r["X"]).all()        # the stored trajectory is still the initial guess


def test_clearance_of_successful_trajectories():
    env = P.freeflyer_env()
    x0, glo, ghi, tf = P.freeflyer_batch(12)
    o = go.Oracle(go.FREEFLYER_SE2, 50, boxes=env)
    n_ok = 0
    for b in range(12):
        o.set_problem(x0[b], glo[b], ghi[b], tf[b])
        r = o.solve(30)
        if not r["successful"]:
            continue
        n_ok += 1
        d = min(o.signed_distance(0, r["X"][k, :2], i)[0] for k in range(1, 50) for i in range(len(env)))
        assert d >= 0.05 - 1e-2 - 1e-3       # linearised rows within eps = 1e-2 (convex_ineq_satisfied)
    assert n_ok >= 6


def test_dubins_trip0_failures_are_infeasible():
    """Config 3 (dubins_car, B = 65 536) has a yield of ~0.65: part of the batch stops with SubproblemFailed.  The ones that
    fail in their FIRST trip are certified here without the oracle's interior point method: the HARD rows of that
    subproblem -- x_1 = x_init, x_N = goal, the trapezoid rows linearised at the straight-line guess (dubins_car.jl:134-146,
    Jacobians by complex step of an independent f) and |u_k| <= 10 for k < N (dynamics.jl:73-81; u_N is free, the a10
    quirk) -- form a linear feasibility problem, which HiGHS declares infeasible.  The penalised state box rows cannot make a
    subproblem infeasible and are left out.  So `SubproblemFailed` at trip 0 is the right verdict of the reference's own
    formulation (scp_gusto.jl:106-111 returns on such a solver status), not a weakness of the solver in oracle/."""
    from scipy.optimize import linprog
    N, n, m, tf, v, kk = 30, 3, 1, 10.0, 2.0, 1.0
    dt = tf / (N - 1)
    x0s, glo, ghi, tfs = P.dubins_batch(1200)
    o = go.Oracle(go.DUBINS_CAR, N)
    failures, feasible_checked = [], 0

    def f(x, u):
        return np.array([v * np.cos(x[2]), v * np.sin(x[2]), kk * u[0]])

    def lp_feasible(b):
        t = np.arange(N) / (N - 1)
        Xp = (1 - t)[:, None] * x0s[b][None] + t[:, None] * glo[b][None]      # init_traj_straightline, dubins_car.jl:86-96
        Up = np.zeros((N, m))
        nz = n + m
        Aeq, beq = [], []
        for i in range(n):                                                   # x_1 = x_init, x_N = goal
            r = np.zeros(N * nz); r[i] = 1; Aeq.append(r); beq.append(x0s[b][i])
            r = np.zeros(N * nz); r[(N - 1) * nz + i] = 1; Aeq.append(r); beq.append(glo[b][i])
        lin = []
        for k in range(N):
            A = np.zeros((n, n))
            for j in range(n):
                e = np.zeros(n, complex); e[j] = 1e-30j
                A[:, j] = np.array([v * np.cos(Xp[k, 2] + e[2]), v * np.sin(Xp[k, 2] + e[2]), kk * Up[k, 0] + 0 * e[2]]).imag / 1e-30
            Bm = np.array([[0.0], [0.0], [kk]])
            lin.append((f(Xp[k], Up[k]) - A @ Xp[k] - Bm @ Up[k], A, Bm))
        for k in range(1, N):                                                # (x_{k-1} - x_k) + dt/2 (lin_{k-1} + lin_k) = 0
            (c0, A0, B0), (c1, A1, B1) = lin[k - 1], lin[k]
            for i in range(n):
                r = np.zeros(N * nz)
                r[(k - 1) * nz + i] += 1; r[k * nz + i] -= 1
                r[(k - 1) * nz:(k - 1) * nz + n] += 0.5 * dt * A0[i]; r[(k - 1) * nz + n] += 0.5 * dt * B0[i, 0]
                r[k * nz:k * nz + n] += 0.5 * dt * A1[i]; r[k * nz + n] += 0.5 * dt * B1[i, 0]
                Aeq.append(r); beq.append(-0.5 * dt * (c0[i] + c1[i]))
        bounds = []
        for k in range(N):
            bounds += [(None, None)] * n + [(-10.0, 10.0) if k < N - 1 else (None, None)]
        res = linprog(np.zeros(N * nz), A_eq=np.array(Aeq), b_eq=np.array(beq), bounds=bounds, method="highs")
        return res.status       # 0 = feasible optimum found, 2 = infeasible

    for b in range(len(x0s)):
        o.set_problem(x0s[b], glo[b], ghi[b], tfs[b])
        r = o.solve(1)
        if r["stop_reason"] == 2 and r["iterations"] == 0:
            failures.append(b)
            assert lp_feasible(b) == 2, b                  # HiGHS: infeasible
        elif feasible_checked < 16 and r["iterations"] == 1:
            feasible_checked += 1
            assert lp_feasible(b) == 0, b                  # ... and the certificate is not vacuous
        if len(failures) >= 32 and feasible_checked >= 16:
            break
    assert len(failures) >= 32, len(failures)


def test_warm_start_defaults_are_the_documented_triples():
    """gusto_ipm_opts.mu_warm < 0 selects the model's warm-start triple (gusto_hip.h; common.hpp: warm_defaults; the oracle's
    table in ipm_solve): a run with the documented triple spelled out must be bit for bit the run with the defaults, and both
    libraries hand out the sentinel."""
    d = g.default_ipm_opts()
    assert d.mu_warm < 0 and d.mu_warm_gain < 0 and d.acc_iter == 0 and d.tol == 1e-8 and d.sigma_max < 0 and d.mu_floor < 0
    boxes, spheres = P.iss_corner_env(True)
    cases = [(go.FREEFLYER_SE2, 50, P.freeflyer_env(), None, P.freeflyer_batch(3), (1e-4, 0.1, 1e-2)),
             (go.DUBINS_CAR, 30, None, None, P.dubins_batch(4), (1e-9, 0.0, 1e-9)),
             (go.ASTROBEE_SE3, 50, boxes, spheres, P.astrobee_se3_batch(2), (1e-6, 1.0, 1e-2)),
             (go.ASTROBEE_SE3_MANIFOLD, 50, boxes, spheres, P.astrobee_manifold_batch(2), (1e-4, 1.0, 1e-2))]
    for model, N, bx, sp, (x0, glo, ghi, tf), (lo, gain, hi) in cases:
        a = go.Oracle(model, N, boxes=bx, spheres=sp)
        io = go.IpmOpts(tol=1e-8, tol_acc=1e-5, mu_floor=1e-10 if model == go.ASTROBEE_SE3_MANIFOLD else 1e-11, tr_tol=1e-6, mu_warm=lo, max_iter=60, acc_iter=0, mu_warm_gain=gain,
                        mu_warm_max=hi, sigma_max=0.1)
        b = go.Oracle(model, N, boxes=bx, spheres=sp, ipm_opts=io)
        for k in range(len(x0)):
            a.set_problem(x0[k], glo[k], ghi[k], tf[k]); b.set_problem(x0[k], glo[k], ghi[k], tf[k])
            ra, rb = a.solve(12), b.solve(12)
            assert ra["iterations"] == rb["iterations"] and list(ra["ipm_iters"]) == list(rb["ipm_iters"])
            np.testing.assert_array_equal(ra["X"], rb["X"])
        # ... and the level does follow the trajectory change: a constant level is a different (slower or equal) run
        if gain > 0:
            io.mu_warm_gain = 0.0
            c = go.Oracle(model, N, boxes=bx, spheres=sp, ipm_opts=io)
            c.set_problem(x0[0], glo[0], ghi[0], tf[0]); a.set_problem(x0[0], glo[0], ghi[0], tf[0])
            assert list(c.solve(12)["ipm_iters"]) != list(a.solve(12)["ipm_iters"])


@pytest.mark.parametrize("name", ["freeflyer_se2_n50", "dubins_car_n30", "astrobee_se3_n50", "astrobee_se3_manifold_n50"])
def test_oracle_reproduces_the_frozen_round3_goldens(name):
    """tests/golden/frozen_r3 (generated by the round-3 oracle, never regenerated): today's oracle with the round-3 interior point
    options spelled out -- constant warm start 1e-4, no bound on Mehrotra's centring parameter, complementarity floor 1e-11 --
    gives those runs again: same trip counts, flags and stop reasons, subproblem optimum to 1e-6 (1e-5 manifold), final
    trajectory to 1e-3.  The defaults may move; this anchor does not move with them."""
    import os
    d = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "frozen_r3", name + ".npz"))
    model, N = int(d["model"]), int(d["N"])
    io = go.IpmOpts(tol=1e-8, tol_acc=1e-5, mu_floor=1e-11, tr_tol=1e-6, mu_warm=1e-4, max_iter=60, acc_iter=0, mu_warm_gain=0.0,
                    mu_warm_max=1e-4, sigma_max=0.0)
    o = go.Oracle(model, N, boxes=d["boxes"], spheres=d["spheres"], ipm_opts=io)
    for b in range(min(3, len(d["x_init"]))):
        o.set_problem(d["x_init"][b], d["goal_lo"][b], d["goal_hi"][b], d["tf"][b])
        Xi, Ui = o.init_straightline()
        sub = o.subproblem(Xi, Ui, o.sp.Delta0, 1.0, o.sp.Delta0 / 8 + o.mp.clearance)
        assert sub["status"] == int(d["sub_status"][b])
        if sub["status"] == 1:
            tol = 1e-5 if model == go.ASTROBEE_SE3_MANIFOLD else 1e-6
            assert np.abs(sub["X"] - d["sub_X"][b]).max() < tol and np.abs(sub["U"] - d["sub_U"][b]).max() < tol
        o.set_problem(d["x_init"][b], d["goal_lo"][b], d["goal_hi"][b], d["tf"][b])
        r = o.solve(int(d["max_iter"]))
        assert r["iterations"] == int(d["iterations"][b]) and r["converged"] == bool(d["converged"][b]) and r["stop_reason"] == int(d["stop_reason"][b])
        assert np.abs(r["X"] - d["X"][b]).max() < 1e-3
