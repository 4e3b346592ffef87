"""GPU tests of the lane-per-problem decomposition of the GuSTO solve (gusto_set_decomposition, csrc/lane.hpp): the same
parity levels as the wave-per-problem kernel for dubins_car -- subproblem, every trip in lock step, whole solves, all
against the CPU oracle through the C ABI -- plus what a kernel without any cross-lane operation must satisfy bit for bit
(determinism, independence of a problem's result from its position and from its neighbours in the wavefront, resume)."""
import numpy as np
import pytest

import test_gpu_parity as tp

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _needs_the_lane_kernel():
    """csrc/lane.hpp is built only with -DGUSTO_WITH_LANE (tools/build_variant.sh lane1 1 -DGUSTO_WITH_LANE; round 6: the kernel is
    4.3x slower than the wave kernel and off by default, so the default library does not carry it)."""
    import gusto_jl_amd as g
    probe = g.BatchSolver(g.DUBINS_CAR, 30, 1)
    try:
        probe.set_decomposition(2)
    except g.GustoError:
        pytest.skip("this libgusto_hip.so was built without -DGUSTO_WITH_LANE")


@pytest.fixture
def lane(monkeypatch):
    """every dubins_car handle the test creates gets gusto_set_decomposition(GUSTO_DECOMP_LANE)"""
    import gusto_jl_amd as g
    monkeypatch.setattr(g.BatchSolver, "default_decomposition", 2)


def _dubins(B):
    import gusto_jl_amd as g
    x0, glo, ghi, tf = g.problems.dubins_batch(B)
    x0[0] = [2.0, 2.0, 2.0]
    return g, x0, glo, ghi, tf


def test_lane_subproblem_parity_dubins(lane):
    g, x0, glo, ghi, tf = _dubins(64)
    tp._sub_parity(g.DUBINS_CAR, 30, None, None, x0, glo, ghi, tf, 1e4, 1.0, 1e4 / 8 + 0.01)


def test_lane_lockstep_parity_dubins(lane):
    g, x0, glo, ghi, tf = _dubins(64)
    print("lockstep dubins (lane)", tp._lockstep_parity(g.DUBINS_CAR, 30, None, None, x0, glo, ghi, tf, max_cold_fail=3))


def test_lane_scp_parity_dubins(lane):
    g, x0, glo, ghi, tf = _dubins(64)
    print("scp dubins (lane) diverged", tp._scp_parity(g.DUBINS_CAR, 30, None, None, x0, glo, ghi, tf, max_diverged=1))


def test_lane_horizons_and_box_goal(lane):
    """ragged horizons (the knot loops are runtime loops: nothing is tied to N = 30) and BoxGoal rows on the last knot"""
    import gusto_jl_amd as g
    for N in (5, 17, 64, 100):
        x0, glo, ghi, tf = g.problems.dubins_batch(8)
        tp._sub_parity(g.DUBINS_CAR, N, None, None, x0, glo, ghi, tf, 1e4, 1.0, 1e4 / 8 + 0.01)
    x0, glo, ghi, tf = g.problems.dubins_batch(8)
    glo, ghi = glo.copy(), ghi.copy()
    glo[:, 2] -= 0.3; ghi[:, 2] += 0.3          # heading free within +-0.3 rad at the goal
    glo[:4, 1] = -0.05; ghi[:4, 1] = 0.05
    tp._sub_parity(g.DUBINS_CAR, 30, None, None, x0, glo, ghi, tf, 1e4, 1.0, 1e4 / 8 + 0.01)


def test_lane_is_lane_local_bit_for_bit():
    """No instruction of the lane kernel crosses lanes: a problem's result cannot depend on the batch around it.  Checked
    bit for bit: two runs, a permuted batch, a batch of another size (another number of problems per wavefront), and a
    solve split into two calls (scp_gusto.jl:67 resumes from SCPS)."""
    import gusto_jl_amd as g
    B, N = 1024, 30
    x0, glo, ghi, tf = g.problems.dubins_batch(B)

    def run(idx, split=None):
        s = g.BatchSolver(g.DUBINS_CAR, N, len(idx), hist_cap=40)
        s.set_decomposition(2)
        s.set_problems(x0[idx], glo[idx], ghi[idx], tf[idx])
        if split:
            s.solve(split)
            s.solve(30 - split)
        else:
            s.solve(30)
        X, U = s.traj()
        return X, U, s.status(), s.launch_info()

    all_ = np.arange(B)
    X1, U1, st1, info = run(all_)
    assert info[1] == 0                       # the lane kernel uses no LDS at all
    X2, U2, st2, _ = run(all_)
    np.testing.assert_array_equal(X1, X2); np.testing.assert_array_equal(U1, U2)
    perm = np.random.default_rng(5).permutation(B)
    X3, U3, st3, _ = run(perm)
    np.testing.assert_array_equal(X1[perm], X3); np.testing.assert_array_equal(U1[perm], U3)
    np.testing.assert_array_equal(st1["iterations"][perm], st3["iterations"])
    X4, U4, st4, _ = run(all_[:100])
    np.testing.assert_array_equal(X1[:100], X4)
    np.testing.assert_array_equal(st1["ipm_iters"][:100], st4["ipm_iters"])
    X5, U5, st5, _ = run(all_, split=4)
    cont = st1["iterations"] > 4              # problems that stopped within the first call iterate again on resume
    np.testing.assert_array_equal(X1[cont], X5[cont]); np.testing.assert_array_equal(U1[cont], U5[cont])
    np.testing.assert_array_equal(st1["iterations"][cont], st5["iterations"][cont])


def test_lane_agrees_with_wave_kernel():
    """the two decompositions on the same 4096 problems: same stop reasons and trip counts (but for the few problems where
    a last-bit difference moves a decision), the same converged set, trajectories to 1e-4"""
    import gusto_jl_amd as g
    B, N = 4096, 30
    x0, glo, ghi, tf = g.problems.dubins_batch(B)
    res = []
    for dec in (1, 2):
        s = g.BatchSolver(g.DUBINS_CAR, N, B, hist_cap=40)
        s.set_decomposition(dec)
        s.set_problems(x0, glo, ghi, tf)
        s.solve(30)
        res.append((s.traj(), s.status(), s.history()))
    (Xw, Uw), sw, hw = res[0][0], res[0][1], res[0][2]
    (Xl, Ul), sl, hl = res[1][0], res[1][1], res[1][2]
    same = (sw["iterations"] == sl["iterations"]) & (sw["stop_reason"] == sl["stop_reason"])
    assert same.mean() >= 0.99, same.mean()
    assert (sw["converged"] != sl["converged"]).mean() <= 0.002
    ok = same & sw["converged"]
    assert np.abs(Xw[ok] - Xl[ok]).max() < 1e-4 and np.abs(Uw[ok] - Ul[ok]).max() < 1e-3      # (measured 2.9e-5 over 2628 converged runs)
    # the histories of the problems on the same path: identical decisions, values to 1e-6
    for key in ("accept_solution", "scp_status", "Delta", "omega"):
        for b in np.flatnonzero(ok)[:256]:
            nh = int(hw["n_hist"][b])
            np.testing.assert_array_equal(hw[key][b, :nh], hl[key][b, :nh])
    b = np.flatnonzero(ok)[:256]
    np.testing.assert_allclose(hw["J_true"][b, :5], hl["J_true"][b, :5], rtol=1e-5, atol=1e-9)   # (measured: 1 of 1280 entries at 1.5e-6)


def test_lane_history_capacity_and_hooks(lane):
    """the history-capacity contract (a handle whose history fills up stops with HIST_FULL, never silently) and the
    caller's trust state (gusto_set_trust_state) through the lane kernel, as tests/test_gpu_parity.py holds them for the
    wave kernel"""
    import gusto_jl_amd as g
    B, N = 64, 30
    x0, glo, ghi, tf = g.problems.dubins_batch(B)
    s = g.BatchSolver(g.DUBINS_CAR, N, B, hist_cap=6)
    s.set_problems(x0, glo, ghi, tf)
    s.solve(30)
    st = s.status()
    assert (st["stop_reason"] != 0).all() and (st["stop_reason"] == 4).any()     # 30 trips cannot fit 6 entries: never MaxIter
    h = s.history()
    assert (h["n_hist"] <= 6).all() and (st["iterations"] <= 5).all()
