"""GPU tests of the drop-in seam around the HIP path: free final time / goal timeline through
solve_gusto_hip, trajectory export of a real device solve (rank 4), long resumed runs, and a non-Python consumer of the
C ABI (tests/c/c_abi_smoke.c)."""
import os
import subprocess
import sys

import numpy as np
import pytest

import gusto_jl_amd as g
import gusto_oracle as go

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
H, P = g.host, g.problems


def _top(x_init, goals, tf=200.0, fixed=True, N=50):
    model = H.FreeflyerSE2()
    gs = H.GoalSet()
    for gl in goals:
        H.add_goal(gs, gl(model))
    PD = H.ProblemDefinition(H.Robot(), model, H.Environment(P.freeflyer_env()), x_init, gs)
    return H.TrajectoryOptimizationProblem(PD, N, tf, fixed_final_time=fixed)


def _solve(TOP, max_iter=30):
    TOS = H.TrajectoryOptimizationSolution(TOP)
    H.solve_SCP(TOS, TOP, H.solve_gusto_hip, H.init_traj_straightline, "hip", max_iter=max_iter)
    return TOS


def test_free_final_time_and_goal_timeline_through_the_hip_path():
    """scp_gusto.jl:185-187,248-250 (`Tf` is a variable with the single row Tf >= 0.1) and goals.jl:18-30 (a GoalSet with a
    goal at an intermediate time): both are inert in the reference, so the HIP solve must return
    bit-for-bit the trajectory of the fixed-time / single-goal problem -- and that trajectory is the oracle's."""
    x_init = P.freeflyer_random_x_init(3)[2]
    final = lambda m: H.Goal(H.PointGoal(P.FREEFLYER_X_GOAL), 200.0, m)
    way = lambda m: H.Goal(H.PointGoal(np.array([1.5, 1.5])), 80.0, [0, 1])             # waypoint at t = 80 s
    box = lambda m: H.Goal(H.BoxGoal(np.array([-0.1]), np.array([0.1])), 120.0, [2])     # heading box at t = 120 s
    ref = _solve(_top(x_init, [final], fixed=True))
    free = _solve(_top(x_init, [final], fixed=False))
    multi = _solve(_top(x_init, [final, way, box], fixed=False))
    assert ref.SCPS.converged and ref.SCPS.iterations >= 3
    for other in (free, multi):
        assert np.array_equal(other.traj.X, ref.traj.X) and np.array_equal(other.traj.U, ref.traj.U)
        assert other.traj.Tf == 200.0 and other.SCPS.iterations == ref.SCPS.iterations
        assert other.SCPS.J_true == ref.SCPS.J_true and other.SCPS.scp_status == ref.SCPS.scp_status
    ks = {round(gl.t_guess): gl.k_timestep for gl in multi.SCPS.SCPP.PD.goal_set.goals}
    assert ks == {80: 2, 120: 1, 200: 1}                                                  # fld(N tf, N t), goals.jl:18-22
    o = go.Oracle(go.FREEFLYER_SE2, 50, boxes=P.freeflyer_env())
    o.set_problem(x_init, P.FREEFLYER_X_GOAL, P.FREEFLYER_X_GOAL, 200.0)
    r = o.solve(30)
    assert r["converged"] == ref.SCPS.converged and r["iterations"] == ref.SCPS.iterations
    assert np.abs(ref.traj.X.T - r["X"]).max() < 1e-3 and np.abs(np.array(ref.SCPS.J_true) - r["J_true"]).max() < 1e-4 * max(1.0, r["J_true"].max())
    with pytest.raises(ValueError):                                                       # the one row there is: Tf >= 0.1
        _top(x_init, [lambda m: H.Goal(H.PointGoal(P.FREEFLYER_X_GOAL), 0.05, m)], tf=0.05, fixed=False)


def test_export_of_a_device_solve(tmp_path):
    """write_solution / write_batch on results that come straight from the device (notebook cell 6)."""
    import h5read
    x_init = P.freeflyer_random_x_init(2)[1]
    TOS = _solve(_top(x_init, [lambda m: H.Goal(H.PointGoal(P.FREEFLYER_X_GOAL), 200.0, m)]))
    for ext in (".h5", ".mat", ".npz"):
        path = os.path.join(tmp_path, "predefined_trajectory_example" + ext)
        g.export.write_solution(path, TOS)
        d = h5read.read_h5(path) if ext == ".h5" else g.export.read(path)
        x = d["traj"]["x_traj"] if ext != ".h5" else d["traj"]["x_traj"].T        # (.h5 holds the column-major view)
        assert np.array_equal(x, TOS.traj.X) and x.shape == (6, 50)
        assert np.allclose(d["traj"]["t_traj"], np.arange(50) * TOS.traj.dt) and int(np.asarray(d["status"]["converged"])) == 1
    B = 6
    x0, glo, ghi, tf = P.freeflyer_batch(B)
    s = g.BatchSolver(g.FREEFLYER_SE2, 50, B, hist_cap=64, boxes=P.freeflyer_env())
    s.set_problems(x0, glo, ghi, tf)
    s.solve(30)
    X, U = s.traj()
    path = os.path.join(tmp_path, "batch.h5")
    g.export.write_batch(path, s, tf)
    d = h5read.read_h5(path)
    assert np.array_equal(d["traj"]["x_traj"], X) and np.array_equal(d["traj"]["u_traj"], U)
    assert np.array_equal(d["status"]["iterations"], s.status()["iterations"]) and d["traj"]["t_traj"].shape == (B, 50)


def test_many_short_resumed_calls_do_not_hit_the_history_capacity():
    """solve_SCPshooting! resumes the SCP one trip per call (traj_opt.jl:29-39); every call consumes two history entries.
    A problem that needs more than 32 trips must run to its budget (the reference's vectors grow without bound)."""
    model = H.DubinsCar()
    x0, glo, ghi, tf = P.dubins_batch(64)
    sp, _ = g.default_params(g.DUBINS_CAR)
    gs = H.GoalSet()
    H.add_goal(gs, H.Goal(H.PointGoal(glo[5]), tf[5], model))
    TOP = H.TrajectoryOptimizationProblem(H.ProblemDefinition(H.Robot(), model, H.BlankEnv(), x0[5], gs), 30, tf[5], True)
    SCPP = H.SCPProblem(TOP)
    SCPP.scp_params.convergence_threshold = 0.0                  # never converges: the run takes its whole budget
    SCPS = H.SCPSolution(SCPP, H.init_traj_straightline(TOP))
    budget = 40
    H.solve_gusto_hip(SCPS, SCPP, "hip", 1, hist_cap=H._hist_cap(budget))
    while SCPS.iterations < budget and SCPS.stop_reason == "MaxIter":
        H.solve_gusto_hip(SCPS, SCPP, "hip", 1)
    assert SCPS.iterations == budget or SCPS.stop_reason in ("SubproblemFailed", "OmegaMaxExceeded")
    assert len(SCPS.J_true) == 2 * SCPS.iterations               # one leading entry per call + one per trip
    # without the sizing the 33rd call reports the full history instead of silently stopping
    SCPS2 = H.SCPSolution(SCPP, H.init_traj_straightline(TOP))
    with pytest.raises(g.GustoError):
        for _ in range(budget):
            H.solve_gusto_hip(SCPS2, SCPP, "hip", 1)
            if SCPS2.stop_reason != "MaxIter":
                pytest.skip("the probe problem stopped early")


def test_c_program_through_the_c_abi(tmp_path):
    """A plain C consumer of include/gusto_hip.h (tests/c/c_abi_smoke.c) on the notebook problem against the oracle."""
    exe = os.path.join(tmp_path, "c_abi_smoke")
    lib = os.path.join(ROOT, "gusto.jl_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "c_abi_smoke.c"), "-o", exe, "-L" + lib, "-lgusto_hip",
                           "-Wl,-rpath," + lib])
    env = P.freeflyer_env()
    boxes = os.path.join(tmp_path, "boxes.txt")
    with open(boxes, "w") as f:
        f.write(f"{len(env)}\n" + "\n".join(" ".join(repr(float(v)) for v in row) for row in env) + "\n")
    out = subprocess.run([exe, boxes], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr
    v = out.stdout.split()
    it, conv, succ, stop, ipm, nh, nJ, nr = (int(x) for x in v[:8])
    o = go.Oracle(go.FREEFLYER_SE2, 50, boxes=env)
    o.set_problem(P.FREEFLYER_X_INIT, P.FREEFLYER_X_GOAL, P.FREEFLYER_X_GOAL, P.FREEFLYER_TF)
    r = o.solve(30)
    assert (it, bool(conv), bool(succ), stop) == (r["iterations"], r["converged"], r["successful"], r["stop_reason"])
    assert nh == len(r["Delta"]) and nJ == len(r["J_true"]) and nr == len(r["rho"])
    p = 8
    J = np.array([float(x) for x in v[p:p + nJ]]); p += nJ
    assert np.abs(J - r["J_true"]).max() < 1e-4 * max(1.0, r["J_true"].max())
    for i in range(nh):
        D, w, cm, acc, scp = float(v[p]), float(v[p + 1]), float(v[p + 2]), int(v[p + 3]), int(v[p + 4]); p += 5
        assert (D, w, acc, scp) == (r["Delta"][i], r["omega"][i], r["accept"][i], r["scp_status"][i])
        assert abs(cm - r["conv"][i]) < 1e-6
    xN = np.array([float(x) for x in v[p:p + 6]]); p += 6
    um = np.array([float(x) for x in v[p:p + 3]])
    assert np.abs(xN - P.FREEFLYER_X_GOAL).max() < 1e-7 and np.abs(um - r["U"][25]).max() < 1e-3


def test_c_program_replays_the_batch_wrappers(tmp_path):
    """tests/c/c_abi_batch.c: the entry points GuSTOHIPBatch.jl's batch wrappers use -- gusto_set_env_batch, asynchronous shards,
    gusto_gather_peer, gusto_set_active with one-trip solves and gusto_shoot, gusto_solve_trajopt_async + gusto_wait -- called in
    the wrappers' order by a plain C program (no Julia in the image: this is their non-Python consumer) with its own checks."""
    exe = os.path.join(tmp_path, "c_abi_batch")
    lib = os.path.join(ROOT, "gusto.jl_amd")
    subprocess.check_call(["gcc", "-std=c99", "-Wall", "-I" + os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "c", "c_abi_batch.c"), "-o", exe, "-L" + lib, "-lgusto_hip", "-lm",
                           "-Wl,-rpath," + lib])
    env = P.freeflyer_env()
    boxes = os.path.join(tmp_path, "boxes.txt")
    with open(boxes, "w") as f:
        f.write(f"{len(env)}\n" + "\n".join(" ".join(repr(float(v)) for v in row) for row in env) + "\n")
    out = subprocess.run([exe, boxes], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr + out.stdout
    assert out.stdout.split()[0] == "ok"


def test_device_side_gather_of_two_handles():
    """gusto_gather_peer (SURVEY.md 8(b) threading row, 8(e)): the final gather of a one-process multi-GPU run in the C ABI --
    two handles (here on one GPU: the box has one) with shards of different sizes, solves still in flight when the gather
    is called; host copies and the device views equal what each handle returns on its own.  Also a TrajOpt pair, whose
    device rows (u | defect) are compacted to u_dim columns before they travel, and the error paths."""
    env = P.freeflyer_env()
    x0, glo, ghi, tf = P.freeflyer_batch(40)
    a = g.BatchSolver(g.FREEFLYER_SE2, 50, 24, hist_cap=40, boxes=env)
    b = g.BatchSolver(g.FREEFLYER_SE2, 50, 16, hist_cap=40, boxes=env)
    a.set_problems(x0[:24], glo[:24], ghi[:24], tf[:24])
    b.set_problems(x0[24:], glo[24:], ghi[24:], tf[24:])
    a.solve_async(30)
    b.solve_async(30)
    X, U = a.gather_peer([a, b])                       # completes both solves
    Xa, Ua = a.traj()
    Xb, Ub = b.traj()
    assert np.array_equal(X, np.concatenate([Xa, Xb])) and np.array_equal(U, np.concatenate([Ua, Ub]))
    Xd, Ud = b.gather_peer([b, a], host=False)         # the other way round, device views
    assert np.array_equal(Xd.cpu().numpy(), np.concatenate([Xb, Xa])) and np.array_equal(Ud.cpu().numpy(), np.concatenate([Ub, Ua]))
    one = g.BatchSolver(g.FREEFLYER_SE2, 50, 40, hist_cap=40, boxes=env)
    one.set_problems(x0, glo, ghi, tf)
    one.solve(30)
    assert np.array_equal(one.traj()[0], X)            # ... and what one handle solving the whole batch produces
    # a handle of another horizon / without problems is refused
    c = g.BatchSolver(g.FREEFLYER_SE2, 40, 4, hist_cap=40, boxes=env)
    with pytest.raises(g.GustoError):
        a.gather_peer([a, c])
    # TrajOpt handles: U travels with the model's u_dim columns
    ta = g.TrajOptSolver(g.FREEFLYER_SE2, 20, 3, boxes=env)
    tb = g.TrajOptSolver(g.FREEFLYER_SE2, 20, 2, boxes=env)
    ta.set_problems(x0[:3], glo[:3], ghi[:3], tf[:3])
    tb.set_problems(x0[3:5], glo[3:5], ghi[3:5], tf[3:5])
    ta.solve(10)
    tb.solve(10)
    Xt, Ut = ta.gather_peer([ta, tb])
    assert Ut.shape == (5, 20, 3)
    assert np.array_equal(Ut, np.concatenate([ta.traj()[1], tb.traj()[1]])) and np.array_equal(Xt, np.concatenate([ta.traj()[0], tb.traj()[0]]))
    Xv, Uv = ta.traj_dev()                              # (the device view of a TrajOpt handle is the compact layout too)
    assert np.array_equal(Uv.cpu().numpy(), ta.traj()[1])
    with pytest.raises(g.GustoError):
        a.gather_peer([a, ta])


def test_trajopt_handle_is_refused_by_the_gusto_only_entry_points():
    """gusto_subproblem / gusto_set_trust_state / gusto_get_history on a TrajOpt handle: GUSTO_ERR_STATE before any buffer is
    touched (on such a handle the device rows of U are u_dim + x_dim wide: the caller's [B][N][u_dim] array would be over-read)."""
    env = P.freeflyer_env()
    x0, glo, ghi, tf = P.freeflyer_batch(2)
    t = g.TrajOptSolver(g.FREEFLYER_SE2, 20, 2, boxes=env)
    t.set_problems(x0, glo, ghi, tf)
    Xp, Up = t.traj()
    with pytest.raises(g.GustoError) as e:
        g.BatchSolver.subproblem(t, Xp, Up, 3.0, 1.0, 0.4)
    assert "-3" in str(e.value)
    with pytest.raises(g.GustoError):
        t.set_trust_state(3.0, 1.0)
    with pytest.raises(g.GustoError):
        g.BatchSolver.history(t)
    Xq, Uq = t.traj()
    assert np.array_equal(Xq, Xp) and np.array_equal(Uq, Up)      # nothing was overwritten
