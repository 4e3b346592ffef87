"""Solver- AND assembly-independent check of the oracle's convex subproblem for the three models the freeflyer SLSQP
test does not cover (dubins_car, astrobeeSE3, astrobeeSE3manifold).  tests/np_models.py writes the dynamics, the
constraint registry and the penalisation straight from the reference's model files (no `o.rows()`, no oracle
Jacobians -- complex-step derivatives of f) and solves the full slack formulation with scipy SLSQP; the oracle's own
row assembly (`assemble_rows`) and interior point method must land on the same optimum.

Tolerances: objective 1e-6 relative (SLSQP at ftol 1e-15 stalls in its line search around 1e-8), U 1e-4; X is only
weakly determined where the trust region is wide (astrobeeSE3 Delta0 = 10, manifold: none), hence 5e-3 there."""
import numpy as np
import pytest

import gusto_oracle as go
import np_models as M
import gusto_jl_amd as g

P = g.problems


def _compare(model_id, model, N, batch, b, boxes=None, sph=None, Delta=None, omega=1.0, xtol=5e-3, utol=1e-4):
    x0, glo, ghi, tf = batch
    o = go.Oracle(model_id, N, boxes=boxes, spheres=sph)
    o.set_problem(x0[b], glo[b], ghi[b], tf[b])
    Xp, Up = o.init_straightline()
    D = Delta or model.Delta0
    r = o.subproblem(Xp, Up, D, omega, D / 8 + model.clearance)
    assert r["status"] == 1
    s = M.solve_subproblem(model, N, tf[b], x0[b], glo[b], ghi[b], Xp, Up, D, omega,
                           boxes if boxes is not None else (), sph if sph is not None else ())
    assert s["eq_violation"] < 1e-10 and s["ineq_min"] > -1e-9           # SLSQP's point is feasible
    assert abs(s["obj"] - r["obj"]) <= 1e-6 * max(1.0, abs(r["obj"])), (s["obj"], r["obj"], s["res"].message)
    # (the interior point objective carries its slacks at the final barrier level: it sits ~1e-8 above SLSQP's)
    assert np.abs(s["U"] - r["U"]).max() < utol, np.abs(s["U"] - r["U"]).max()
    assert np.abs(s["X"] - r["X"]).max() < xtol, np.abs(s["X"] - r["X"]).max()
    return s


@pytest.mark.parametrize("b", [0, 2, 3])
def test_dubins_against_independent_slsqp(b):
    s = _compare(go.DUBINS_CAR, M.Dubins, 12, P.dubins_batch(4), b, xtol=1e-8, utol=1e-8)
    assert s["n_pen"] == 6 * 12 and s["n_hard"] == 2 * 11      # 180/58 rows at N = 30 (SURVEY.md 8(a) size table)


@pytest.mark.parametrize("b,Delta,omega", [(0, None, 1.0), (1, 0.5, 10.0), (2, 2.0, 100.0)])
def test_astrobee_se3_against_independent_slsqp(b, Delta, omega):
    bx, sp = P.iss_corner_env(True)
    _compare(go.ASTROBEE_SE3, M.AstrobeeSE3, 8, P.astrobee_se3_batch(3), b, bx, sp, Delta, omega)


@pytest.mark.parametrize("b,omega", [(0, 1.0), (1, 10.0)])
def test_manifold_against_independent_slsqp(b, omega):
    """includes the +-eps pair of the penalised quaternion-norm equality (scp_gusto.jl:297-311: j = 1 a hard bound,
    j = 2 the L1 penalty), -qw <= 0 and the BoxGoal rows on q"""
    bx, sp = P.iss_corner_env(True)
    s = _compare(go.ASTROBEE_SE3_MANIFOLD, M.AstrobeeSE3Manifold, 6, P.astrobee_manifold_batch(3), b, bx[:12], sp, None, omega)
    assert s["n_pen"] == 6 * (4 + 12 + 2)       # per knot: quat-norm, -qw, v, w + every obstacle (toggle = 125 m)


def test_complex_step_jacobians_match_the_oracle_tables():
    """the hand-written Jacobian tables of the oracle (astrobee_se3.jl:206-233, astrobee_se3_manifold.jl:248-296)
    against complex-step derivatives of the independently written f"""
    rng = np.random.default_rng(5)
    for mid, model in ((go.DUBINS_CAR, M.Dubins), (go.ASTROBEE_SE3, M.AstrobeeSE3), (go.ASTROBEE_SE3_MANIFOLD, M.AstrobeeSE3Manifold)):
        o = go.Oracle(mid, 8)
        for _ in range(4):
            x, u = rng.uniform(-0.6, 0.6, model.n), rng.uniform(-0.6, 0.6, model.m)
            f, A, B = o.dynamics(x, u)
            A2, B2 = M.jac(model, x, u)
            assert np.abs(f - model.f(x, u)).max() < 1e-14
            assert np.abs(A - A2).max() < 1e-13 and np.abs(B - B2).max() < 1e-13
