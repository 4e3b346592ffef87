"""Host-side mirror logic that needs no GPU: goal flattening, the reference's free-final-time and goal-timeline
semantics, shard bounds, trajectory export."""
import os
import sys

import numpy as np
import pytest

import gusto_jl_amd as g

H, P = g.host, g.problems


def _top(goals, tf=200.0, fixed=True, N=50):
    model = H.FreeflyerSE2()
    gs = H.GoalSet()
    for gl in goals:
        H.add_goal(gs, gl(model))
    PD = H.ProblemDefinition(H.Robot(), model, H.Environment(P.freeflyer_env()), P.FREEFLYER_X_INIT, gs)
    return H.TrajectoryOptimizationProblem(PD, N, tf, fixed_final_time=fixed)


def test_free_final_time_is_accepted_and_inert():
    """scp_gusto.jl:185-187,248-250: Tf is a variable with the single row Tf >= 0.1 and enters nothing else, so the
    problem is the fixed-final-time one (dt = tf_guess/(N-1))."""
    a = _top([lambda m: H.Goal(H.PointGoal(P.FREEFLYER_X_GOAL), 200.0, m)], fixed=True)
    b = _top([lambda m: H.Goal(H.PointGoal(P.FREEFLYER_X_GOAL), 200.0, m)], fixed=False)
    assert not b.fixed_final_time and b.tf_guess == a.tf_guess
    ta, tb = H.init_traj_straightline(a), H.init_traj_straightline(b)
    assert np.array_equal(ta.X, tb.X) and ta.dt == tb.dt == 200.0 / 49
    with pytest.raises(ValueError):
        _top([lambda m: H.Goal(H.PointGoal(P.FREEFLYER_X_GOAL), 0.05, m)], tf=0.05, fixed=False)     # violates Tf >= 0.1


def test_goal_timeline_as_in_the_reference():
    """Only the goals AT tf_guess are registered (freeflyer_se2.jl:352-358) and drive init_traj_straightline
    (:97-111); goals at intermediate times are carried and ignored.  k_timestep follows goals.jl:20 literally."""
    way = np.array([1.5, 2.0, 0.0, 0.0, 0.0, 0.0])
    top = _top([lambda m: H.Goal(H.PointGoal(way), 100.0, m),
                lambda m: H.Goal(H.PointGoal(P.FREEFLYER_X_GOAL[:2]), 200.0, [0, 1]),
                lambda m: H.Goal(H.BoxGoal([-0.1], [0.1]), 200.0, [2])])
    lo, hi = H._goal_bounds(top.PD.goal_set, 6, 200.0)
    assert np.array_equal(lo[:2], P.FREEFLYER_X_GOAL[:2]) and np.array_equal(hi[:2], P.FREEFLYER_X_GOAL[:2])
    assert lo[2] == -0.1 and hi[2] == 0.1 and np.isinf(lo[3:]).all() and np.isinf(hi[3:]).all()
    assert [gl.k_timestep for gl in top.PD.goal_set.goals] == [2, 1, 1]          # floor(200/100), floor(200/200)
    X = H.init_traj_straightline(top).X
    assert np.allclose(X[:2, -1], P.FREEFLYER_X_GOAL[:2]) and np.allclose(X[2:, -1], 0.0)   # centre of the final goals, zeros elsewhere


def test_shard_bounds_cover_the_batch_once():
    for B, G in ((4096, 8), (37, 2), (5, 8), (2048, 3)):
        cuts = [H.shard_bounds(B, G, r) for r in range(G)]
        assert cuts[0][0] == 0 and cuts[-1][1] == B
        assert all(a[1] == b[0] for a, b in zip(cuts, cuts[1:])) and all(hi >= lo for lo, hi in cuts)


@pytest.mark.parametrize("ext", [".mat", ".npz"])
def test_trajectory_export_round_trip(tmp_path, ext):
    """examples/freeflyerSE2.ipynb cell 6: traj/{x_traj, u_traj, t_traj} + zero-indexed ind_x / ind_u."""
    rng = np.random.default_rng(0)
    B, N = 3, 50
    X, U, tf = rng.standard_normal((B, N, 6)), rng.standard_normal((B, N, 3)), np.array([200.0, 150.0, 100.0])
    path = os.path.join(tmp_path, "predefined_trajectory_example" + ext)
    g.export.write(path, g.FREEFLYER_SE2, X, U, tf, dict(converged=np.array([1, 1, 0])))
    d = g.export.read(path)
    assert d["traj"]["x_traj"].shape == (B, 6, N) and d["traj"]["u_traj"].shape == (B, 3, N)
    assert np.array_equal(d["traj"]["x_traj"][1], X[1].T) and np.array_equal(d["traj"]["u_traj"][2], U[2].T)
    assert np.allclose(d["traj"]["t_traj"][1], np.arange(N) * 150.0 / (N - 1))         # collect(0:dt:Tf)
    assert int(d["ind_x"]["theta"]) == 2 and int(d["ind_x"]["omega"]) == 5 and int(d["ind_u"]["M"]) == 2
    assert list(np.asarray(d["status"]["converged"]).ravel()) == [1, 1, 0]
    # single trajectory, the notebook's shapes
    g.export.write(path, g.FREEFLYER_SE2, X[0], U[0], 200.0)
    d = g.export.read(path)
    assert d["traj"]["x_traj"].shape == (6, N) and d["traj"]["t_traj"].shape == (N,) and d["traj"]["t_traj"][-1] == 200.0


def _h5_tool(name):
    import shutil
    for c in (shutil.which(name), "/opt/conda/bin/" + name):
        if c and os.path.exists(c):
            return c
    return None


def test_trajectory_export_hdf5(tmp_path):
    """The notebook's own container (cell 6, `h5open(...)`): a real HDF5 file, parsed back by the spec-following reader
    of the suite and -- when libhdf5's command line tools are installed -- by libhdf5 itself."""
    import subprocess
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    import h5read
    rng = np.random.default_rng(1)
    N = 50
    X, U = rng.standard_normal((N, 6)), rng.standard_normal((N, 3))
    path = os.path.join(tmp_path, "predefined_trajectory_example.h5")
    g.export.write(path, g.FREEFLYER_SE2, X, U, 200.0)
    d = h5read.read_h5(path)
    assert sorted(d) == ["ind_u", "ind_x", "traj"] and sorted(d["traj"]) == ["t_traj", "u_traj", "x_traj"]
    # HDF5 dimensions (N, x_dim): HDF5.jl (column-major) presents it as the notebook's x_dim x N matrix TOS.traj.X
    assert d["traj"]["x_traj"].shape == (N, 6) and np.array_equal(d["traj"]["x_traj"], X)
    assert d["traj"]["u_traj"].shape == (N, 3) and np.array_equal(d["traj"]["u_traj"], U)
    assert np.allclose(d["traj"]["t_traj"], np.arange(N) * 200.0 / (N - 1)) and d["traj"]["t_traj"].dtype == np.float64
    assert {k: int(v) for k, v in d["ind_x"].items()} == dict(x=0, y=1, theta=2, vx=3, vy=4, omega=5)
    assert {k: int(v) for k, v in d["ind_u"].items()} == dict(Fx=0, Fy=1, M=2)
    # a batch with status vectors, 13 index entries (more than the default 8 of a symbol table node)
    Xb, Ub = rng.standard_normal((3, N, 13)), rng.standard_normal((3, N, 6))
    pb = os.path.join(tmp_path, "batch.h5")
    g.export.write(pb, g.ASTROBEE_SE3_MANIFOLD, Xb, Ub, np.array([40.0, 41.0, 42.0]),
                   dict(converged=np.array([1, 0, 1]), iterations=np.array([7, 30, 9], dtype=np.int32)))
    db = h5read.read_h5(pb)
    assert np.array_equal(db["traj"]["x_traj"], Xb) and len(db["ind_x"]) == 13 and int(db["ind_x"]["wz"]) == 12
    assert db["status"]["iterations"].dtype == np.int32 and list(db["status"]["converged"]) == [1, 0, 1]
    tool = _h5_tool("h5dump")
    if tool:       # libhdf5 reads what h5lite wrote
        out = subprocess.run([tool, "-d", "/traj/t_traj", "-d", "/ind_x/omega", path], capture_output=True, text=True)
        assert out.returncode == 0, out.stderr
        assert "H5T_IEEE_F64LE" in out.stdout and "( 50 ) / ( 50 )" in out.stdout and "H5T_STD_I64LE" in out.stdout
        assert "(0): 5" in out.stdout
        ls = subprocess.run([_h5_tool("h5ls"), "-r", pb], capture_output=True, text=True)
        assert ls.returncode == 0 and "/traj/x_traj Dataset {3, 50, 13}" in " ".join(ls.stdout.split()), ls.stdout
