"""GPU tests of the wave-per-chain kernels of the 12/13-state models (gusto_set_decomposition: GUSTO_DECOMP_WAVE2 / WAVE4,
csrc/segw.hpp): the horizon split into two or four Riccati chains, a wavefront each, and each knot's obstacle rows shared
between the waves.  Same subproblems to the same tolerances as the one-wave kernel; the KKT solve is reassociated, so iterates
differ in rounding.  The oracle runs the reference's sequential recursion (scp_gusto.jl:178-314 through one Riccati sweep).

AUTO picks these kernels by batch size (four waves up to two problems per CU, two up to eight), so every other GPU test of these
models with a small batch runs the four-wave kernel already; here each decomposition is forced, on the same problems."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

WAVE, WAVE2, WAVE4 = 1, 3, 4


def _mods():
    import gusto_jl_amd as g
    import gusto_oracle as go
    return g, go


def _batch(g, name, B, first=0):
    P = g.problems
    boxes, sph = P.iss_corner_env(True)
    if name == "astrobee_se3":
        return g.ASTROBEE_SE3, boxes, sph, P.astrobee_se3_batch(B, first=first)
    return g.ASTROBEE_SE3_MANIFOLD, boxes, sph, P.astrobee_manifold_batch(B, first=first)


def _solve(g, model, boxes, sph, batch, dec, max_iter=30):
    s = g.BatchSolver(model, 50, len(batch[0]), hist_cap=max_iter + 8, boxes=boxes, spheres=sph)
    s.set_decomposition(dec)
    s.set_problems(*batch)
    s.solve(max_iter)
    X, U = s.traj()
    return X, U, s.status(), s.history()


@pytest.mark.parametrize("name", ["astrobee_se3", "astrobee_se3_manifold"])
@pytest.mark.parametrize("dec", [WAVE2, WAVE4])
def test_whole_solves_against_the_oracle(name, dec):
    """48 whole GuSTO solves per model and chain count: SCP iterations, converged flags and stop reasons of EVERY problem equal the
    oracle's; the final trajectory of a converged problem within 1e-3 (the trajectory-level tolerance of test_gpu_parity.py; the
    manifold model's optimum is weakly determined inside its +-1e-4 BoxGoal on the goal quaternion)."""
    g, go = _mods()
    model, boxes, sph, batch = _batch(g, name, 48)
    X, U, st, h = _solve(g, model, boxes, sph, batch, dec)
    r = go.solve_batch(model, 50, boxes, sph, *batch, 30, 0)
    assert np.array_equal(st["iterations"], r["iterations"])
    assert np.array_equal(st["converged"].astype(bool), r["converged"].astype(bool))
    cv = r["converged"].astype(bool)
    assert cv.sum() >= 40
    dx = np.abs(X - r["X"]).reshape(len(X), -1).max(1)
    assert dx[cv].max() < 1e-3, dx[cv].max()
    # the number of KKT solves: a rounding-level difference may move a solve's last iteration
    assert abs(int(st["ipm_iters"].sum()) - int(r["ipm_iters"].sum())) <= 0.02 * r["ipm_iters"].sum()


@pytest.mark.parametrize("name", ["astrobee_se3", "astrobee_se3_manifold"])
def test_chain_counts_agree_with_one_wave(name):
    """one, two and four waves per problem on the same 64 problems (another window of the problem set): identical SCP iteration counts
    and converged flags, trajectories of converged problems within 1e-4 of the one-wave kernel's."""
    g, _ = _mods()
    model, boxes, sph, batch = _batch(g, name, 64, first=300)
    X1, U1, st1, _ = _solve(g, model, boxes, sph, batch, WAVE)
    cv = st1["converged"].astype(bool)
    for dec in (WAVE2, WAVE4):
        X, U, st, _ = _solve(g, model, boxes, sph, batch, dec)
        assert np.array_equal(st["iterations"], st1["iterations"]), dec
        assert np.array_equal(st["converged"], st1["converged"]), dec
        dx = np.abs(X - X1).reshape(len(X), -1).max(1)
        assert dx[cv].max() < 1e-4, (dec, dx[cv].max())


@pytest.mark.parametrize("dec", [WAVE2, WAVE4])
def test_subproblem_parity_with_chains(dec, monkeypatch):
    """the convex subproblem alone (gusto_subproblem) through the chain kernels, against the oracle: the tolerances of
    test_gpu_parity.py's subproblem tests"""
    import test_gpu_parity as tp
    g, _ = _mods()
    monkeypatch.setattr(g.BatchSolver, "default_decomposition", dec)
    tp.test_subproblem_parity_astrobee_se3()
    tp.test_subproblem_parity_astrobee_manifold()


def test_short_horizons_and_other_models_are_refused():
    """a chain needs GUSTO_SEG_MIN_N = 4 stages: N = 12 has no four-wave kernel (two waves are fine); freeflyerSE2 has neither"""
    g, _ = _mods()
    P = g.problems
    boxes, sph = P.iss_corner_env(True)
    x0, glo, ghi, tf = P.astrobee_se3_batch(4)
    s = g.BatchSolver(g.ASTROBEE_SE3, 12, 4, hist_cap=16, boxes=boxes, spheres=sph)
    s.set_problems(x0, glo, ghi, tf)
    s.set_decomposition(WAVE4)
    with pytest.raises(g.GustoError):
        s.solve(3)
    s.set_decomposition(WAVE2)
    s.solve(3)
    ref = g.BatchSolver(g.ASTROBEE_SE3, 12, 4, hist_cap=16, boxes=boxes, spheres=sph)
    ref.set_decomposition(WAVE)
    ref.set_problems(x0, glo, ghi, tf)
    ref.solve(3)
    assert np.array_equal(s.status()["iterations"], ref.status()["iterations"])
    assert np.abs(s.traj()[0] - ref.traj()[0]).max() < 1e-6
    f = g.BatchSolver(g.FREEFLYER_SE2, 50, 4, hist_cap=16, boxes=P.freeflyer_env())
    f.set_problems(*P.freeflyer_batch(4))
    f.set_decomposition(WAVE2)
    with pytest.raises(g.GustoError):
        f.solve(3)
    with pytest.raises(g.GustoError):
        f.set_decomposition(5)


def test_odd_horizons():
    """N = 33 and N = 63 (chains of unequal length; 63 of 64 lanes in use): whole solves of 12 problems with two and four waves equal
    the one-wave kernel's iteration counts"""
    g, _ = _mods()
    P = g.problems
    boxes, sph = P.iss_corner_env(True)
    for N in (33, 63):
        batch = P.astrobee_se3_batch(12)
        out = {}
        for dec in (WAVE, WAVE2, WAVE4):
            s = g.BatchSolver(g.ASTROBEE_SE3, N, 12, hist_cap=24, boxes=boxes, spheres=sph)
            s.set_decomposition(dec)
            s.set_problems(*batch)
            s.solve(12)
            out[dec] = (s.traj()[0], s.status())
        for dec in (WAVE2, WAVE4):
            assert np.array_equal(out[dec][1]["iterations"], out[WAVE][1]["iterations"]), (N, dec)
            cv = out[WAVE][1]["converged"].astype(bool)
            if cv.any():
                assert np.abs(out[dec][0] - out[WAVE][0])[cv].max() < 1e-4, (N, dec)


@pytest.mark.parametrize("dec", [WAVE2, WAVE4])
def test_resume_and_active_mask_with_chains(dec):
    """the state machine around the chain kernels: solve(4) then solve(26) continues exactly like one solve(30) (scp_gusto.jl:67; bit for
    bit: the same kernel runs the same trips), and gusto_set_active hands out only the listed problems (the others keep trajectory,
    histories and counters untouched) -- the run of the listed ones does not depend on who else is in the launch."""
    g, _ = _mods()
    model, boxes, sph, batch = _batch(g, "astrobee_se3_manifold", 24)
    a = g.BatchSolver(model, 50, 24, hist_cap=80, boxes=boxes, spheres=sph)
    a.set_decomposition(dec); a.set_problems(*batch); a.solve(30)
    Xa, Ua = a.traj(); sa = a.status()
    b = g.BatchSolver(model, 50, 24, hist_cap=80, boxes=boxes, spheres=sph)
    b.set_decomposition(dec); b.set_problems(*batch); b.solve(4)
    assert (b.status()["iterations"] <= 4).all()
    b.solve(26)
    Xb, Ub = b.traj(); sb = b.status()
    same = sa["iterations"] > 4          # (problems that stopped within the first call keep iterating on resume)
    np.testing.assert_array_equal(sa["iterations"][same], sb["iterations"][same])
    np.testing.assert_array_equal(Xa[same], Xb[same])
    np.testing.assert_array_equal(Ua[same], Ub[same])
    c = g.BatchSolver(model, 50, 24, hist_cap=80, boxes=boxes, spheres=sph)
    c.set_decomposition(dec); c.set_problems(*batch)
    X0, U0 = c.traj()
    act = np.zeros(24, bool); act[[1, 5, 6, 17, 23]] = True
    c.set_active(act); c.solve(30)
    Xc, Uc = c.traj(); sc = c.status()
    np.testing.assert_array_equal(Xc[~act], X0[~act])
    assert (sc["iterations"][~act] == 0).all()
    np.testing.assert_array_equal(sc["iterations"][act], sa["iterations"][act])
    np.testing.assert_array_equal(Xc[act], Xa[act])
