"""Per-problem environments (gusto_set_env_batch): in the reference every ProblemDefinition owns its env and its
Workspace(robot, env) (src/types.jl:12-24,32-39), so a batch may mix obstacle layouts -- BASELINE.json's north star names
"random initial states / obstacle layouts".  The HIP path with one keep-out set per problem is compared with the CPU
oracle (one oracle instance per layout) and, bit for bit, with the same kernel run layout by layout through gusto_set_env."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

SUB_ATOL = 1e-6


def _mods():
    import gusto_jl_amd as g
    import gusto_oracle as go
    return g, go


def _freeflyer_layouts():
    """Four distinct keep-out sets: the notebook's (4 slabs + 10 boxes), slabs + 4 of the boxes + a disc, the slabs alone,
    and nothing at all (BlankEnv)."""
    g, _ = _mods()
    P = g.problems
    full = P.freeflyer_env()
    return [(full, None),
            (np.vstack([full[:4], full[[5, 8, 9, 12]]]), np.array([[1.7, 1.2, 0.0, 0.18]])),
            (full[:4].copy(), None),
            (None, None)]


def _compare_with_oracle(model, N, layouts, lay_of, x0, glo, ghi, tf, Delta, omega, toggle, max_iter, sub_atol=SUB_ATOL):
    g, go = _mods()
    B = len(x0)
    bl = [layouts[l][0] for l in lay_of]
    sl = [layouts[l][1] for l in lay_of]
    s = g.BatchSolver(model, N, B, hist_cap=max_iter + 8)
    s.set_env_batch(bl, sl)
    s.set_schedule(0, 1)
    s.set_problems(x0, glo, ghi, tf)
    Xp, Up = s.traj()
    sub = s.subproblem(Xp, Up, Delta, omega, toggle)
    s.set_problems(x0, glo, ghi, tf)
    s.solve(max_iter)
    X, U = s.traj()
    st, h = s.status(), s.history()
    oracles = [go.Oracle(model, N, boxes=b, spheres=sp) for b, sp in layouts]
    n_diverged = 0
    for b in range(B):
        o = oracles[lay_of[b]]
        o.set_problem(x0[b], glo[b], ghi[b], tf[b])
        ro = o.subproblem(Xp[b], Up[b], Delta, omega, toggle)
        assert sub["status"][b] == ro["status"], (b, sub["status"][b], ro["status"])
        if ro["status"] in (1, 2):
            assert np.abs(sub["X"][b] - ro["X"]).max() < sub_atol and np.abs(sub["U"][b] - ro["U"]).max() < sub_atol, b
            assert abs(sub["obj"][b] - ro["obj"]) <= 1e-8 * max(1.0, abs(ro["obj"])), b
        r = o.solve(max_iter)
        nh = int(h["n_hist"][b])
        same = (nh == len(r["omega"]) and np.array_equal(h["scp_status"][b, :nh], r["scp_status"])
                and np.array_equal(h["omega"][b, :nh], r["omega"]) and np.array_equal(h["Delta"][b, :nh], r["Delta"]))
        if not same:          # two implementations may take a different branch late in a run (test_gpu_parity._scp_parity)
            n_diverged += 1
            continue
        assert bool(st["converged"][b]) == r["converged"] and int(st["iterations"][b]) == r["iterations"], b
        assert int(st["stop_reason"][b]) == r["stop_reason"], b
        w = max(1.0, r["omega"].max() / 1e3)
        assert np.abs(X[b] - r["X"]).max() < 1e-3 * w and np.abs(U[b] - r["U"]).max() < 1e-3 * w, b
        np.testing.assert_allclose(h["J_true"][b, :h["nJ"][b]], r["J_true"], rtol=1e-4 * w, atol=1e-9)
    assert n_diverged <= max(1, B // 32), n_diverged
    # the same kernel, layout by layout through gusto_set_env: bit for bit what the mixed batch produced
    for l, (bx, sp) in enumerate(layouts):
        idx = np.nonzero(np.asarray(lay_of) == l)[0]
        if len(idx) == 0:
            continue
        s1 = g.BatchSolver(model, N, len(idx), hist_cap=max_iter + 8, boxes=bx, spheres=sp)
        s1.set_schedule(0, 1)
        s1.set_problems(x0[idx], glo[idx], ghi[idx], tf[idx])
        s1.solve(max_iter)
        X1, U1 = s1.traj()
        assert np.array_equal(X1, X[idx]) and np.array_equal(U1, U[idx]), l
        h1 = s1.history()
        assert np.array_equal(h1["n_hist"], h["n_hist"][idx]) and np.array_equal(h1["n_rho"], h["n_rho"][idx])
        for j, b in enumerate(idx):      # (entries beyond the counts are whatever the allocation held)
            for k, c in (("Delta", "n_hist"), ("omega", "n_hist"), ("scp_status", "n_hist"), ("ipm_iters", "n_hist"),
                         ("rho", "n_rho"), ("J_true", "nJ")):
                assert np.array_equal(h1[k][j, :h1[c][j]], h[k][b, :h[c][b]]), (l, b, k)
        s1.close()
    return st, h


def test_env_batch_freeflyer_four_layouts_vs_oracle():
    g, _ = _mods()
    P = g.problems
    B = 24
    x0, glo, ghi, tf = P.freeflyer_batch(B)
    x0[0] = P.FREEFLYER_X_INIT
    lay_of = [b % 4 for b in range(B)]
    st, h = _compare_with_oracle(g.FREEFLYER_SE2, 50, _freeflyer_layouts(), lay_of, x0, glo, ghi, tf, 3.0, 1.0,
                                 3.0 / 8 + 0.05, 30)
    assert st["converged"].sum() >= B - 2


def test_env_batch_astrobee_se3_three_layouts_vs_oracle():
    """3-D model, boxes and spheres: the ISS corner with add_obstacles!, without, and with two other spheres."""
    g, _ = _mods()
    P = g.problems
    bx, sp = P.iss_corner_env(True)
    bx0, _sp0 = P.iss_corner_env(False)
    other = sp.copy()
    other[:, :3] += np.array([0.25, -0.2, 0.15])
    other[:, 3] *= 0.75
    layouts = [(bx, sp), (bx0, None), (bx0, other)]
    B = 6
    x0, glo, ghi, tf = P.astrobee_se3_batch(B)
    _compare_with_oracle(g.ASTROBEE_SE3, 50, layouts, [b % 3 for b in range(B)], x0, glo, ghi, tf, 10.0, 1.0,
                         10.0 / 8 + 0.03, 12)


def test_env_batch_random_layouts_full_batch_properties():
    """B = 4096 freeflyer problems, every one with its own random keep-out set: size-independent properties of the solve
    (hard rows, trapezoid defects, clearance against the problem's OWN obstacles for the successful ones), bitwise
    determinism, and a sample of 24 problems against the oracle."""
    g, go = _mods()
    P = g.problems
    B, N = 4096, 50
    x0, glo, ghi, tf = P.freeflyer_batch(B)
    bl, sl = P.freeflyer_random_layouts(B)
    assert len({(len(b), len(s_)) for b, s_ in zip(bl, sl)}) >= 8          # really a mix of layouts
    s = g.BatchSolver(g.FREEFLYER_SE2, N, B, hist_cap=40)
    s.set_env_batch(bl, sl)
    s.set_problems(x0, glo, ghi, tf)
    s.solve(30)
    X, U = s.traj()
    st = s.status()
    s.set_problems(x0, glo, ghi, tf)
    s.solve(30)
    X2, U2 = s.traj()
    assert np.array_equal(X, X2) and np.array_equal(U, U2)                 # same batch twice: bitwise equal
    assert np.isfinite(X).all() and np.isfinite(U).all()
    ok = st["converged"]
    assert ok.mean() > 0.9, ok.mean()
    assert np.abs(X[:, 0] - x0).max() < 1e-9 and np.abs(X[ok, -1] - glo[ok]).max() < 1e-6
    mp = g.default_params(g.FREEFLYER_SE2)[1]
    dt = tf[0] / (N - 1)
    f = lambda x, u: np.concatenate([x[..., 3:6], u[..., :2] / mp.mass, u[..., 2:3] / mp.Jdiag[2]], axis=-1)
    defect = X[:, 1:] - X[:, :-1] - 0.5 * dt * (f(X[:, :-1], U[:, :-1]) + f(X[:, 1:], U[:, 1:]))
    assert np.abs(defect[ok]).max() < 1e-6
    # clearance of the successful problems against their own layout (penalised rows below eps = 1e-2, scp_gusto.jl:316-343)
    succ = np.nonzero(st["successful"])[0]
    rng = np.random.default_rng(0)
    for b in rng.choice(succ, 64, replace=False):
        o = go.Oracle(g.FREEFLYER_SE2, N, boxes=bl[b], spheres=sl[b])
        for k in range(N):
            for i in range(len(bl[b]) + len(sl[b])):
                nh = np.zeros(3)
                d = o.L.go_signed_distance(o.h, 0, np.ascontiguousarray(X[b, k, :3] * [1, 1, 0]), i, nh)
                assert mp.clearance - d < 1e-2 + 1e-9, (b, k, i, d)
    # ... and a sample against the oracle, each with its own layout
    for b in rng.choice(B, 24, replace=False):
        o = go.Oracle(g.FREEFLYER_SE2, N, boxes=bl[b], spheres=sl[b])
        o.set_problem(x0[b], glo[b], ghi[b], tf[b])
        r = o.solve(30)
        if r["iterations"] != int(st["iterations"][b]):
            continue
        assert bool(st["converged"][b]) == r["converged"], b
        assert np.abs(X[b] - r["X"]).max() < 1e-3 * max(1.0, r["omega"].max() / 1e3), b


def test_env_batch_call_order_and_errors():
    g, _ = _mods()
    P = g.problems
    x0, glo, ghi, tf = P.freeflyer_batch(4)
    full = P.freeflyer_env()
    s = g.BatchSolver(g.FREEFLYER_SE2, 50, 4, hist_cap=40)
    s.set_problems(x0, glo, ghi, tf)
    s.set_env_batch([full] * 3)                      # a different number of problems: refused at the solve, loudly
    with pytest.raises(g.GustoError):
        s.solve(2)
    s.set_env_batch([full] * 4)                      # (either call may come first)
    s.solve(30)
    Xa, _ = s.traj()
    s.set_env(full)                                  # back to one shared keep-out set: same problems, same bits
    s.set_problems(x0, glo, ghi, tf)
    s.solve(30)
    Xb, _ = s.traj()
    assert np.array_equal(Xa, Xb)
    with pytest.raises(g.GustoError):
        s.set_env_batch([np.zeros((65, 6))] * 4)     # more than 64 components in one problem


def test_batch_seam_with_one_environment_per_problem():
    """solve_SCP_batch! with TOPs whose ProblemDefinitions carry different envs (types.jl:32-39) against solve_SCP! one by one."""
    g, _ = _mods()
    H, P = g.host, g.problems
    lays = _freeflyer_layouts()
    x0, glo, ghi, tf = P.freeflyer_batch(6)
    TOPs = []
    for b in range(6):
        model = H.FreeflyerSE2()
        gs = H.GoalSet()
        H.add_goal(gs, H.Goal(H.PointGoal(glo[b]), tf[b], model))
        bx, sp = lays[b % 4]
        TOPs.append(H.TrajectoryOptimizationProblem(H.ProblemDefinition(H.Robot(), model, H.Environment(bx, sp), x0[b], gs),
                                                    50, tf[b], fixed_final_time=True))
    TOSs = [H.TrajectoryOptimizationSolution(t) for t in TOPs]
    out = H.solve_SCP_batch(TOSs, TOPs, H.solve_gusto_hip, H.init_traj_straightline, "hip", max_iter=30)
    for b, t in enumerate(TOPs):
        one = H.solve_SCP(H.TrajectoryOptimizationSolution(t), t, H.solve_gusto_hip, H.init_traj_straightline, "hip", max_iter=30)
        assert one.iterations == out[b].iterations and one.converged == out[b].converged
        assert np.array_equal(one.traj.X, out[b].traj.X)
