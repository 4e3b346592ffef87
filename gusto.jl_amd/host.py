"""Host-side mirror of the reference's driver interface for the GuSTO path.

The reference is Julia (no Julia toolchain in this image), so the host layer above the C ABI is written in Python
with the reference's names and argument meaning; julia/GuSTOHIP.jl holds the same logic as `ccall`s for a Julia
host.  What is mirrored (reference file:line):

  Goal / PointGoal / BoxGoal / GoalSet / add_goal!        src/goals.jl:1-48, src/types.jl:26-30
  ProblemDefinition, Trajectory                           src/types.jl:32-46,235
  TrajectoryOptimizationProblem / ...Solution             src/types.jl:48-61,196-202,228
  SCPProblem, SCPSolution (+ SCPParam_GuSTO histories)    src/types.jl:78-86,150-173,233,256-259; scp_gusto.jl:4-24
  solve_SCP!(TOS, TOP, solve_method!, init, solver; ...)  src/traj_opt.jl:47-72
  solve_gusto_jump! -> solve_gusto_hip!                   src/scp/scp_gusto.jl:49   (the plug-in seam)

plus `solve_SCP_batch!`, which the reference does not have (it solves one problem at a time), and the rank
sharding used for multi-GPU runs (independent problems, no data-path collective).
"""
import numpy as np

from . import _capi
from ._capi import (ASTROBEE_SE3, ASTROBEE_SE3_MANIFOLD, DUBINS_CAR, FREEFLYER_SE2, MODEL_DIMS, SCP_STATUS,
                    SOLVER_STATUS, BatchSolver)


# ---- models / robots / environments: parameter holders (src/dynamics/*.jl, src/robot/*.jl, src/environment/*.jl) ----
class _Model:
    model_id = None

    def __init__(self):
        self.x_dim, self.u_dim = MODEL_DIMS[self.model_id]


class FreeflyerSE2(_Model):
    model_id = FREEFLYER_SE2


class DubinsCar(_Model):
    model_id = DUBINS_CAR


class AstrobeeSE3(_Model):
    model_id = ASTROBEE_SE3


class AstrobeeSE3Manifold(_Model):
    model_id = ASTROBEE_SE3_MANIFOLD


class Robot:
    """Freeflyer() / Astrobee3D() / Car(): the constants live in gusto_model_params (gusto_default_params)."""


class Environment:
    """keepout_zones + obstacle_set as AABBs [min xyz | max xyz] and spheres [c xyz | r]  (types.jl:12-24)."""

    def __init__(self, boxes=None, spheres=None):
        self.boxes = np.zeros((0, 6)) if boxes is None else np.asarray(boxes, float).reshape(-1, 6)
        self.spheres = np.zeros((0, 4)) if spheres is None else np.asarray(spheres, float).reshape(-1, 4)


def Table(room="stanford"):
    from . import problems
    if room != "stanford":
        raise NotImplementedError("only Table(:stanford) is tabulated")
    return Environment(problems.table_stanford_boxes())


def ISSCorner(with_obstacles=False):
    from . import problems
    b, s = problems.iss_corner_env(with_obstacles)
    return Environment(b, s)


def BlankEnv():
    return Environment()


# ---- goals (src/goals.jl) ---------------------------------------------------------------------------
class PointGoal:
    def __init__(self, point):
        self.point = np.asarray(point, float)


class BoxGoal:
    def __init__(self, lower_bound, upper_bound):
        self.lower_bound, self.upper_bound = np.asarray(lower_bound, float), np.asarray(upper_bound, float)


class Goal:
    def __init__(self, params, t_guess, ind_coordinates):
        """ind_coordinates: a model (all coordinates) or 0-based indices (the reference is 1-based)."""
        self.params, self.t_guess, self.k_timestep, self.t_final = params, float(t_guess), None, None
        if isinstance(ind_coordinates, _Model):
            ind_coordinates = range(ind_coordinates.x_dim)
        self.ind_coordinates = np.asarray(list(ind_coordinates), int)


class GoalSet:
    def __init__(self):
        self.goals = []


def add_goal(goal_set, goal):
    goal_set.goals.append(goal)
    goal_set.goals.sort(key=lambda g: g.t_guess)


def _goal_bounds(goal_set, x_dim, tf_guess):
    """Flatten the goals active at tf_guess into (lo, hi): lo == hi -> hard equality row (csbce_goal_constraints),
    lo < hi -> hard box rows (csbci_goal_constraints), +-inf -> no goal on that coordinate (dynamics.jl:30-42)."""
    lo, hi = np.full(x_dim, -np.inf), np.full(x_dim, np.inf)
    for g in goal_set.goals:
        if g.t_guess != tf_guess:
            continue   # every row function of the reference hard-codes knot N (dynamics.jl:33,40)
        if isinstance(g.params, PointGoal):
            lo[g.ind_coordinates] = hi[g.ind_coordinates] = g.params.point
        elif isinstance(g.params, BoxGoal):
            lo[g.ind_coordinates], hi[g.ind_coordinates] = g.params.lower_bound, g.params.upper_bound
        else:
            raise NotImplementedError("BallGoal has no row function in the reference either")
    return lo, hi


# ---- problem / solution containers (src/types.jl) ---------------------------------------------------
class ProblemDefinition:
    def __init__(self, robot, model, env, x_init, goal_set):
        self.robot, self.model, self.env = robot, model, env
        self.x_init, self.goal_set = np.asarray(x_init, float), goal_set


class Trajectory:
    def __init__(self, X, U, Tf):
        self.X, self.U, self.Tf = X, U, float(Tf)        # X is x_dim x N as in the reference
        self.dt = self.Tf / (self.X.shape[1] - 1)         # types.jl:235


class TrajectoryOptimizationProblem:
    """types.jl:48-61,228.  fixed_final_time=False mirrors what the reference actually does with it: `Tf` becomes a free
    JuMP variable with the single row `Tf >= 0.1` (scp_gusto.jl:185-187,248-250), but no dynamics row, cost term or
    trust region contains it -- the trapezoid rows use traj_prev.dt (freeflyer_se2.jl:170) -- so the subproblem leaves
    it at its start value tf_guess (scp_gusto.jl:102) and every trip runs with dt = tf_guess/(N-1).  The mirror keeps
    Tf = tf_guess and enforces the one row there is."""

    def __init__(self, PD, N, tf_guess, fixed_final_time=False):
        if not fixed_final_time and tf_guess < 0.1:
            raise ValueError("free final time: the only row on Tf is Tf >= 0.1 (scp_gusto.jl:248-250); tf_guess violates it")
        self.PD, self.N, self.tf_guess, self.fixed_final_time = PD, int(N), float(tf_guess), bool(fixed_final_time)
        assign_timesteps(PD.goal_set, self.N, self.tf_guess)       # types.jl:58 via the constructor at :228


def assign_timesteps(goal_set, N, tf_guess):
    """goals.jl:18-22, literally: k_timestep = fld(N*tf_guess, N*t_guess) = floor(tf_guess / t_guess) -- 1 for a goal at the
    final time, NOT its knot index.  Nothing reads it for placement: the goal row functions hard-code knot N
    (dynamics.jl:33,40) and SCPConstraints only registers the goals whose time equals tf_guess (freeflyer_se2.jl:352-358),
    so goals at intermediate times are carried in the GoalSet and ignored by the solve -- here exactly as there."""
    for g in goal_set.goals:
        g.k_timestep = int((N * tf_guess) // (N * g.t_guess)) if g.t_guess > 0 else None


class SCPProblem:
    def __init__(self, TOP):
        self.PD, self.N, self.tf_guess = TOP.PD, TOP.N, TOP.tf_guess
        self.scp_params, self.model_params = _capi.default_params(TOP.PD.model.model_id)   # SCPParam + SCPParam_GuSTO
        self.Delta_vec, self.omega_vec, self.rho_vec = [], [], []
        self.trust_region_satisfied_vec, self.convex_ineq_satisfied_vec = [], []


class SCPSolution:
    def __init__(self, SCPP, traj_init):
        self.traj, self.SCPP = traj_init, SCPP
        self.dual = np.zeros(SCPP.PD.model.x_dim)
        self.J_true, self.J_full = [], []
        self.solver_status, self.scp_status = ["NA"], ["NA"]
        self.accept_solution, self.convergence_measure = [True], [0.0]
        self.successful = self.converged = False
        self.stop_reason = "MaxIter"      # gusto_get_status stop reason of the last solve_method! call (_capi.STOP_REASON)
        self.iterations, self.iter_elapsed_times, self.total_time = 0, [0.0], 0.0
        self._solver = None     # the GuSTO gusto_handle that owns the device-side state of this solution (resume, shooting)
        self._solver_trajopt = None   # the TrajOpt handle of this solution (gusto_create_trajopt), kept apart: neither algorithm
                                      # ever finds the other's handle


class TrajectoryOptimizationSolution:
    def __init__(self, TOP):
        n, m = TOP.PD.model.x_dim, TOP.PD.model.u_dim
        self.traj = Trajectory(np.zeros((n, TOP.N)), np.zeros((m, TOP.N)), TOP.tf_guess)
        self.SCPS = None
        self.total_time = 0.0


def init_traj_straightline(TOP):
    """freeflyer_se2.jl:97-111: linear interpolation x_init -> centre of the goals at the final time, U = 0."""
    n, m, N = TOP.PD.model.x_dim, TOP.PD.model.u_dim, TOP.N
    lo, hi = _goal_bounds(TOP.PD.goal_set, n, TOP.tf_guess)
    fin = np.isfinite(lo) & np.isfinite(hi)
    xg = np.where(fin, 0.5 * (np.where(fin, lo, 0.0) + np.where(fin, hi, 0.0)), 0.0)
    t = np.arange(N) / (N - 1)
    X = (1 - t)[None, :] * TOP.PD.x_init[:, None] + t[None, :] * xg[:, None]
    return Trajectory(X, np.zeros((m, N)), TOP.tf_guess)


# ---- the plug-in: solve_method!(SCPS, SCPP, solver, max_iter, force; kw...) ---------------------------------------
def _fetch(bs, XU=None):
    """Everything a solution needs from one handle, copied device -> host ONCE per solve (not once per problem).
    `XU`: the shard's trajectories when a device-side gather (gusto_gather_peer) has already brought them over."""
    X, U = XU if XU is not None else bs.traj()
    return dict(X=X, U=U, st=bs.status(), h=bs.history(), dual=bs.dual())


def _fill_solution(SCPS, SCPP, snap, b, elapsed):
    st, h = snap["st"], snap["h"]
    stop = int(st["stop_reason"][b])
    if stop == 4:
        raise _capi.GustoError("history capacity of the handle reached before iter_cap (GUSTO_STOP_HIST_FULL): the "
                               "solver would silently stop iterating; create the solution with a larger hist_cap")
    SCPS.traj.X, SCPS.traj.U = snap["X"][b].T.copy(), snap["U"][b].T.copy()
    nh, nJ, nr = h["n_hist"][b], h["nJ"][b], h["n_rho"][b]
    SCPS.J_true, SCPS.J_full = list(h["J_true"][b, :nJ]), list(h["J_full"][b, :nJ])
    # a failed subproblem pushes its status and nothing else before the early return (scp_gusto.jl:106-111)
    ns = nh + 1 if stop == 2 else nh
    SCPS.solver_status = [SOLVER_STATUS[int(v)] for v in h["solver_status"][b, :ns]]
    SCPS.scp_status = [SCP_STATUS[int(v)] for v in h["scp_status"][b, :nh]]
    SCPS.accept_solution = [bool(v) for v in h["accept_solution"][b, :nh]]
    SCPS.convergence_measure = list(h["convergence_measure"][b, :nh])
    SCPS.iterations = int(st["iterations"][b])
    SCPS.converged, SCPS.successful = bool(st["converged"][b]), bool(st["successful"][b])
    SCPS.stop_reason = _capi.STOP_REASON[stop]
    SCPS.dual = snap["dual"][b].copy()
    SCPS.total_time += elapsed
    SCPS.iter_elapsed_times = [0.0] + [SCPS.total_time / max(1, SCPS.iterations)] * SCPS.iterations
    SCPP.Delta_vec, SCPP.omega_vec = list(h["Delta"][b, :nh]), list(h["omega"][b, :nh])
    SCPP.rho_vec = list(h["rho"][b, :nr])
    SCPP.trust_region_satisfied_vec = [bool(v) for v in h["trust_region_satisfied"][b, :nh]]
    SCPP.convex_ineq_satisfied_vec = [bool(v) for v in h["convex_ineq_satisfied"][b, :nh]]
    SCPP.obstacle_toggle_distance = SCPP.Delta_vec[-1] / 8 + SCPP.model_params.clearance


def _hist_cap(max_iter):
    """Every call adds its iterations plus one leading J_true/rho entry; room for a few resumed calls."""
    return max(64, 4 * max_iter + 16)


def solve_gusto_hip(SCPS, SCPP, solver="hip", max_iter=30, force=False, device=0, hist_cap=None, **kwarg):
    """Same positional signature as solve_gusto_jump! (scp_gusto.jl:49).  Mutates SCPS / SCPP in place; a second
    call resumes from SCPS (iter_cap = iterations + max_iter, scp_gusto.jl:67).

    `hist_cap` sizes the history vectors of the handle the FIRST call creates: every call consumes one leading
    J_true / rho entry plus one entry per trip, so a caller that resumes in short calls (solve_SCPshooting: one trip per
    call) passes the capacity of its whole iteration budget; the default covers a few resumed calls of `max_iter`."""
    model = SCPP.PD.model
    n, N = model.x_dim, SCPP.N
    bs = SCPS._solver
    if not (type(bs) is BatchSolver and bs.h and bs.model == model.model_id and bs.N == N and bs.device == device):
        if bs is not None:      # (a handle of another model / horizon / device cannot resume this solution)
            bs.close()
        env = SCPP.PD.env
        bs = BatchSolver(model.model_id, N, 1, hist_cap=hist_cap or _hist_cap(max_iter), device=device, boxes=env.boxes,
                         spheres=env.spheres, scp_params=SCPP.scp_params, model_params=SCPP.model_params)
        lo, hi = _goal_bounds(SCPP.PD.goal_set, n, SCPP.tf_guess)
        bs.set_problems(SCPP.PD.x_init[None], lo[None], hi[None], [SCPP.tf_guess],
                        SCPS.traj.X.T[None].copy(), SCPS.traj.U.T[None].copy())
        SCPS._solver = bs
    bs.solve(max_iter, force)
    _fill_solution(SCPS, SCPP, _fetch(bs), 0, bs.last_solve_ms() * 1e-3)


def solve_SCP(TOS, TOP, solve_method, init_method, solver="hip", max_iter=30, force=False, **kwarg):
    """traj_opt.jl:47-72 (both overloads: `init_method` may be a function of TOP or a Trajectory)."""
    SCPP = SCPProblem(TOP)
    traj_init = init_method(TOP) if callable(init_method) else init_method
    SCPS = SCPSolution(SCPP, traj_init)
    TOS.traj, TOS.SCPS = SCPS.traj, SCPS          # aliasing as in traj_opt.jl:58
    solve_method(SCPS, SCPP, solver, max_iter, force, **kwarg)
    return SCPS


# ---- TrajOpt behind the same seam: solve_trajopt_jump! -> solve_trajopt_hip! (src/scp/scp_trajopt.jl:33) ------------------
def solve_trajopt_hip(SCPS, SCPP, solver="hip", max_iter=125, force=False, device=0, trajopt_params=None, **kwarg):
    """Same positional signature as solve_trajopt_jump! (scp_trajopt.jl:33): solve_SCP!(TOS, TOP, solve_trajopt_hip!, init,
    "hip").  FreeflyerSE2 and AstrobeeSE3.  Fills the SCPSolution vectors the reference pushes (J_true, J_full,
    convergence_measure, solver_status, dual, iterations, converged) and SCPP.param.alg's rho_vec / mu_vec / s_vec /
    xtol_vec / ftol_vec / ctol_vec (here: attributes of SCPP).  `max_iter` caps the convex subproblems (the reference
    computes iter_cap and never reads it); `force` is unused there too."""
    model = SCPP.PD.model
    n, N = model.x_dim, SCPP.N
    env = SCPP.PD.env
    tp = trajopt_params or _capi.default_trajopt_params(model.model_id)
    total = tp.max_penalty_iteration * tp.max_convex_iteration * tp.max_trust_iteration
    # one handle per SCPSolution: a repeated call re-uses it (device allocations, kernel attributes) and only sets the problem
    # again -- solve_trajopt_jump! has no resume either, every call runs the whole three-loop schedule from SCPS.traj
    bs = SCPS._solver_trajopt
    if not (isinstance(bs, _capi.TrajOptSolver) and bs.h and bs.model == model.model_id and bs.N == N and bs.device == device
            and bs.hist_cap >= 2 * total + 16):
        if bs is not None:
            bs.close()
        bs = _capi.TrajOptSolver(model.model_id, N, 1, hist_cap=2 * total + 16, device=device, boxes=env.boxes,
                                 spheres=env.spheres, model_params=SCPP.model_params, trajopt_params=tp)
    else:
        bs.set_env(env.boxes, env.spheres)
        bs._chk(bs.L.gusto_set_params(bs.h, None, _capi.C.byref(SCPP.model_params)), "set_params")
        bs._chk(bs.L.gusto_set_trajopt_params(bs.h, _capi.C.byref(tp)), "set_trajopt_params")
    if kwarg.get("ipm_opts") is not None:       # (applied on every call, re-used handle or not)
        bs._chk(bs.L.gusto_set_ipm_opts(bs.h, _capi.C.byref(kwarg["ipm_opts"])), "set_ipm_opts")
    lo, hi = _goal_bounds(SCPP.PD.goal_set, n, SCPP.tf_guess)
    bs.set_problems(SCPP.PD.x_init[None], lo[None], hi[None], [SCPP.tf_guess], SCPS.traj.X.T[None].copy(), SCPS.traj.U.T[None].copy())
    bs.solve(max_iter)
    X, U = bs.traj()
    _fill_trajopt_solution(SCPS, SCPP, dict(X=X, U=U, st=bs.status(), h=bs.history(), dual=bs.dual()), 0,
                           bs.last_solve_ms() * 1e-3)
    SCPS._solver_trajopt = bs


def _fill_trajopt_solution(SCPS, SCPP, snap, b, elapsed):
    """The vectors solve_trajopt_jump! pushes (scp_trajopt.jl:60-157) for problem b of a TrajOpt handle's snapshot."""
    X, U, st, h = snap["X"], snap["U"], snap["st"], snap["h"]
    ns = int(st["iterations"][b])
    SCPS.traj.X, SCPS.traj.U = X[b].T.copy(), U[b].T.copy()
    SCPS.J_true = list(h["J_true"][b, :ns + 1])
    SCPS.J_full = list(h["J_full"][b, :ns])
    SCPS.convergence_measure = [0.0] + list(h["convergence_measure"][b, 1:ns + 1])
    stop = int(st["stop_reason"][b])
    SCPS.solver_status = [SOLVER_STATUS[int(v)] for v in h["solver_status"][b, :ns + 1 + (stop == 2)]]
    SCPS.iterations, SCPS.converged, SCPS.successful = ns, bool(st["converged"][b]), False
    SCPS.stop_reason = _capi.STOP_REASON[stop]
    SCPS.dual = snap["dual"][b].copy()
    SCPS.total_time += elapsed
    SCPS.iter_elapsed_times = [0.0] + [SCPS.total_time / max(1, ns)] * ns
    SCPP.rho_vec, SCPP.s_vec = list(h["rho_vec"][b, :ns + 1]), list(h["s_vec"][b, :ns + 1])
    SCPP.mu_vec = list(h["mu_vec"][b, :h["n_mu"][b]])
    SCPP.xtol_vec, SCPP.ftol_vec = list(h["xtol_vec"][b, :h["n_xtol"][b]]), list(h["ftol_vec"][b, :h["n_ftol"][b]])
    SCPP.ctol_vec = list(h["ctol_vec"][b, :h["n_ctol"][b]])
    SCPP.obstacle_toggle_distance = SCPP.model_params.clearance + 1.0      # scp_trajopt.jl:64


# ---- indirect shooting seeded by the SCP dual (src/shooting.jl, src/traj_opt.jl:4-45, src/types.jl:187-227) --------------
class ShootingProblem:
    """types.jl:219-227: p0 = SCPS.dual, tf = SCPS.traj.Tf, x_goal = centre of the goals at the final time (zeros
    elsewhere).  (At HEAD the constructor reads an undefined `N`; `TOP.N` is what it means.)"""

    def __init__(self, TOP, SCPS):
        n = TOP.PD.model.x_dim
        lo, hi = _goal_bounds(TOP.PD.goal_set, n, TOP.tf_guess)
        fin = np.isfinite(lo) & np.isfinite(hi)
        self.PD, self.N, self.tf = TOP.PD, TOP.N, SCPS.traj.Tf
        self.dt = self.tf / (self.N - 1)
        self.p0 = np.array(SCPS.dual, float)
        self.x_goal = np.where(fin, 0.5 * (np.where(fin, lo, 0.0) + np.where(fin, hi, 0.0)), 0.0)
        self._scps = SCPS


class ShootingSolution:
    """types.jl:198-209"""

    def __init__(self, SP, traj_init):
        self.traj, self.SP = traj_init, SP
        self.J_true, self.prob_status, self.convergence_measure = [], ["NA"], [float("nan")]
        self.converged, self.iter_elapsed_times = False, [0.0]


def cost_true(traj):
    """freeflyer_se2.jl:66-76 / dubins_car.jl:54-69: trapezoid control effort."""
    U = traj.U
    return float(np.sum(0.5 * traj.dt * (U[:, :-1] ** 2 + U[:, 1:] ** 2)))


def convergence_metric(traj, traj_prev):
    """traj_opt.jl:74-85"""
    return float(np.linalg.norm(traj.X - traj_prev.X, axis=0).max() / np.linalg.norm(traj.X, axis=0).max())


def solve_shooting(SS, SP, **opts):
    """solve!(SS, SP) (shooting.jl:4-49) on the GPU: gusto_shoot on the handle that holds the SCP state of this problem."""
    import time
    bs = SP._scps._solver
    if bs is None:
        raise _capi.GustoError("solve!(SS, SP): run solve_gusto_hip! on the SCPSolution first (it owns the device state)")
    t0 = time.perf_counter()
    r = bs.shoot(SP.p0[None], **opts)
    el = time.perf_counter() - t0
    if int(r["status"][0]) == 1:                                   # sol_newton.f_converged
        new_traj = Trajectory(r["X"][0].T.copy(), r["U"][0].T.copy(), SP.tf)
        SS.prob_status.append("Optimal")
        SS.J_true.append(cost_true(new_traj))
        SS.convergence_measure.append(convergence_metric(new_traj, SS.traj))
        SS.traj = new_traj
    else:
        SS.prob_status.append("Diverged")
        SS.J_true.append(float("nan"))
        SS.convergence_measure.append(float("nan"))
    SS.iter_elapsed_times.append(el)
    SS.p0 = r["p0"][0]
    return r


def solve_SCPshooting(TOS, TOP, solve_method, init_method, solver="hip", max_iter=30, **kwarg):
    """traj_opt.jl:4-45: one SCP iteration, then alternately a shooting attempt seeded by the SCP dual and another SCP
    iteration, until two consecutive successful shooting runs agree (sum of their convergence measures below the
    threshold) or the SCP itself converges / runs out of iterations."""
    SCPP = SCPProblem(TOP)
    traj_init = init_method(TOP) if callable(init_method) else init_method
    TOS.SCPS = SCPS = SCPSolution(SCPP, traj_init)
    SP = ShootingProblem(TOP, SCPS)
    TOS.SS = SS = ShootingSolution(SP, Trajectory(traj_init.X.copy(), traj_init.U.copy(), traj_init.Tf))
    # one-trip calls: each consumes two history entries (its leading J_true / rho entry and the trip), so the handle is
    # sized for the whole budget here -- the reference's vectors grow without bound (scp_gusto.jl:15-19)
    kwarg.setdefault("hist_cap", _hist_cap(max_iter))
    solve_method(SCPS, SCPP, solver, 1, **kwarg)
    SS.J_true.append(SCPS.J_true[0])
    # (a run that stopped for good -- failed subproblem, omega > omega_max -- leaves the loop: the reference's solve_method!
    # returns early there without counting an iteration, scp_gusto.jl:107-111,163-166, and its loop would spin for ever)
    while not SCPS.converged and SCPS.iterations < max_iter and SCPS.stop_reason not in _DEAD_STOPS:
        SP = ShootingProblem(TOP, SCPS)
        solve_shooting(SS, SP)
        spread = 2
        cm = SS.convergence_measure[-spread:]
        if SCPS.iterations > spread and not any(np.isnan(cm)) and sum(cm) <= SCPP.scp_params.convergence_threshold:
            SS.converged = True
            TOS.traj = Trajectory(SS.traj.X.copy(), SS.traj.U.copy(), SS.traj.Tf)
            TOS.total_time = SCPS.total_time + sum(SS.iter_elapsed_times)
            return TOS
        solve_method(SCPS, SCPP, solver, 1, **kwarg)
    TOS.traj = Trajectory(SCPS.traj.X.copy(), SCPS.traj.U.copy(), SCPS.traj.Tf)
    TOS.total_time = SCPS.total_time + sum(SS.iter_elapsed_times)
    return TOS


_DEAD_STOPS = ("SubproblemFailed", "OmegaMaxExceeded", "HistoryFull")


def solve_SCPshooting_batch(TOSs, TOPs, solve_method=None, init_method=init_traj_straightline, solver="hip", max_iter=30,
                            device=0, **kwarg):
    """solve_SCPshooting! (traj_opt.jl:4-45) for a list of problems of one model and horizon on ONE handle: per round one
    gusto_shoot and one gusto_solve(1) over the problems still in their loop (gusto_set_active carries the per-problem
    condition `!SCPS.converged && SCPS.iterations < max_iter` of traj_opt.jl:23), every problem leaving exactly where the
    single-problem driver above leaves -- its SCPSolution / ShootingSolution are those solve_SCPshooting fills, bit for bit."""
    import time
    if solve_method not in (None, solve_gusto_hip):
        raise NotImplementedError("solve_SCPshooting_batch!: the shooting refinement follows the GuSTO solve (solve_gusto_hip)")
    if len(TOSs) != len(TOPs) or not TOPs:
        raise ValueError("solve_SCPshooting_batch!: need as many solutions as problems, at least one")
    TOP0 = TOPs[0]
    model, N, B = TOP0.PD.model, TOP0.N, len(TOPs)
    n = model.x_dim
    for t in TOPs[1:]:
        if type(t.PD.model) is not type(model) or t.N != N:
            raise ValueError("solve_SCPshooting_batch!: all problems must share the model type and N")
    same_env = all(np.array_equal(t.PD.env.boxes, TOP0.PD.env.boxes) and np.array_equal(t.PD.env.spheres, TOP0.PD.env.spheres)
                   for t in TOPs[1:])
    SCPPs = [SCPProblem(t) for t in TOPs]
    inits = [init_method(t) if callable(init_method) else Trajectory(init_method.X.copy(), init_method.U.copy(), init_method.Tf)
             for t in TOPs]
    SCPSs = [SCPSolution(p, i) for p, i in zip(SCPPs, inits)]
    # ONE handle solves the batch with ONE set of SCP and model parameters (gusto_set_params): a problem with another robot's
    # mass or another trust-region schedule would silently be solved with problem 0's
    for p in SCPPs[1:]:
        if bytes(p.scp_params) != bytes(SCPPs[0].scp_params) or bytes(p.model_params) != bytes(SCPPs[0].model_params):
            raise ValueError("solve_SCPshooting_batch!: all problems must share the SCP parameters and the model / robot parameters")
    bs = BatchSolver(model.model_id, N, B, hist_cap=kwarg.get("hist_cap") or _hist_cap(max_iter), device=device,
                     boxes=TOP0.PD.env.boxes, spheres=TOP0.PD.env.spheres, scp_params=SCPPs[0].scp_params,
                     model_params=SCPPs[0].model_params)
    if not same_env:
        bs.set_env_batch([t.PD.env.boxes for t in TOPs], [t.PD.env.spheres for t in TOPs])
    bounds = [_goal_bounds(t.PD.goal_set, n, t.tf_guess) for t in TOPs]
    bs.set_problems(np.stack([t.PD.x_init for t in TOPs]), np.stack([b[0] for b in bounds]), np.stack([b[1] for b in bounds]),
                    np.array([t.tf_guess for t in TOPs]), np.stack([t.X.T for t in inits]), np.stack([t.U.T for t in inits]))
    SSs = []
    for b, (TOS, TOP) in enumerate(zip(TOSs, TOPs)):
        TOS.SCPS = SCPSs[b]
        SP = ShootingProblem(TOP, SCPSs[b])
        TOS.SS = ShootingSolution(SP, Trajectory(inits[b].X.copy(), inits[b].U.copy(), inits[b].Tf))
        SSs.append(TOS.SS)

    def scp_round(live):       # solve_method!(SCPS, SCPP, solver, 1) of every live problem: ONE launch
        bs.set_active(live)
        bs.solve(1)
        snap = _fetch(bs)
        per = bs.last_solve_ms() * 1e-3 / max(1, int(np.sum(live)))
        for b in np.flatnonzero(live):
            _fill_solution(SCPSs[b], SCPPs[b], snap, b, per)

    live = np.ones(B, bool)
    scp_round(live)
    for b in range(B):
        SSs[b].J_true.append(SCPSs[b].J_true[0])
    done_by_shooting = np.zeros(B, bool)
    while True:
        live = np.array([not S.converged and S.iterations < max_iter and S.stop_reason not in _DEAD_STOPS for S in SCPSs]) \
            & ~done_by_shooting
        if not live.any():
            break
        bs.set_active(live)
        t0 = time.perf_counter()
        r = bs.shoot()                                    # seeds = SCPS.dual of every problem, read on the device
        el = (time.perf_counter() - t0) / int(live.sum())
        for b in np.flatnonzero(live):
            SS, SCPS = SSs[b], SCPSs[b]
            SS.SP = ShootingProblem(TOPs[b], SCPS)
            if int(r["status"][b]) == 1:
                new_traj = Trajectory(r["X"][b].T.copy(), r["U"][b].T.copy(), SS.SP.tf)
                SS.prob_status.append("Optimal")
                SS.J_true.append(cost_true(new_traj))
                SS.convergence_measure.append(convergence_metric(new_traj, SS.traj))
                SS.traj = new_traj
            else:
                SS.prob_status.append("Diverged")
                SS.J_true.append(float("nan"))
                SS.convergence_measure.append(float("nan"))
            SS.iter_elapsed_times.append(el)
            SS.p0 = r["p0"][b]
            cm = SS.convergence_measure[-2:]
            if SCPS.iterations > 2 and not any(np.isnan(cm)) and sum(cm) <= SCPPs[b].scp_params.convergence_threshold:
                SS.converged = True
                done_by_shooting[b] = True
        live &= ~done_by_shooting
        if live.any():
            scp_round(live)
    bs.set_active(None)
    for b, TOS in enumerate(TOSs):
        src = SSs[b].traj if done_by_shooting[b] else SCPSs[b].traj
        TOS.traj = Trajectory(src.X.copy(), src.U.copy(), src.Tf)
        TOS.total_time = SCPSs[b].total_time + sum(SSs[b].iter_elapsed_times)
        SCPSs[b]._solver = None          # (the batch handle is not a per-problem resume handle)
    bs.close()
    return TOSs


# ---- batch API (new: the reference has no batch mode) --------------------------------------------------------------
def shard_bounds(B, world_size, rank):
    """Contiguous block of ceil(B/G) problems per rank; the tail rank may get fewer."""
    per = -(-B // world_size)
    lo = min(B, rank * per)
    return lo, min(B, lo + per)


def solve_SCP_batch(TOSs, TOPs, solve_method=None, init_method=init_traj_straightline, solver="hip", max_iter=30,
                    force=False, device=0, devices=None, decomposition=0):
    """All TOPs must share model and N (each may bring its own environment); one gusto_solve covers the whole list.

    `decomposition` (gusto_set_decomposition; GuSTO handles only): 0 = the library's choice by batch size -- for the 12/13-state
    models two or four wavefronts per problem while the batch leaves SIMDs idle --, 1 one wave per problem, 3 / 4 two / four.

    `devices` = list of GPU ordinals: the problems are sharded in contiguous blocks (shard_bounds) over
    one handle per entry, every shard is enqueued with gusto_solve_async and the shards run concurrently -- the
    single-process form of the multi-GPU path (an ordinal may repeat: two shards on one GPU overlap like two batches)."""
    if solve_method not in (None, solve_gusto_hip, solve_trajopt_hip):
        raise NotImplementedError("solve_SCP_batch! runs the batched kernels: solve_method must be solve_gusto_hip or solve_trajopt_hip")
    trajopt = solve_method is solve_trajopt_hip
    if len(TOSs) != len(TOPs) or not TOPs:
        raise ValueError("solve_SCP_batch!: need as many solutions as problems, at least one")
    TOP0 = TOPs[0]
    model, N = TOP0.PD.model, TOP0.N
    n = model.x_dim
    B = len(TOPs)
    for t in TOPs[1:]:      # one gusto_handle = one model, one horizon, one Workspace
        if type(t.PD.model) is not type(model) or t.N != N:
            raise ValueError("solve_SCP_batch!: all problems must share the model type and N")
    # every ProblemDefinition owns its env (types.jl:32-39): a batch whose environments differ goes through
    # gusto_set_env_batch (one Workspace per problem), a batch that shares one through gusto_set_env
    same_env = all(np.array_equal(t.PD.env.boxes, TOP0.PD.env.boxes) and np.array_equal(t.PD.env.spheres, TOP0.PD.env.spheres)
                   for t in TOPs[1:])
    devs = list(devices) if devices else [device]
    sp, mp = _capi.default_params(model.model_id)
    x0 = np.stack([t.PD.x_init for t in TOPs])
    bounds = [_goal_bounds(t.PD.goal_set, n, t.tf_guess) for t in TOPs]
    lo, hi = np.stack([b[0] for b in bounds]), np.stack([b[1] for b in bounds])
    tf = np.array([t.tf_guess for t in TOPs])
    # every problem owns its initial trajectory (a Trajectory passed for all of them is copied, never aliased)
    inits = [init_method(t) if callable(init_method) else Trajectory(init_method.X.copy(), init_method.U.copy(), init_method.Tf)
             for t in TOPs]
    X0 = np.stack([t.X.T for t in inits])
    U0 = np.stack([t.U.T for t in inits])
    shards = []
    for r, dv in enumerate(devs):
        b0, b1 = shard_bounds(B, len(devs), r)
        if b1 <= b0:
            continue
        if trajopt:     # the second algorithm behind the seam (scp_trajopt.jl:33): one gusto_solve_trajopt per shard
            tp = _capi.default_trajopt_params(model.model_id)
            total = tp.max_penalty_iteration * tp.max_convex_iteration * tp.max_trust_iteration
            bs = _capi.TrajOptSolver(model.model_id, N, b1 - b0, hist_cap=2 * total + 16, device=dv, boxes=TOP0.PD.env.boxes,
                                     spheres=TOP0.PD.env.spheres, model_params=mp, trajopt_params=tp)
        else:
            bs = BatchSolver(model.model_id, N, b1 - b0, hist_cap=_hist_cap(max_iter), device=dv,
                             boxes=TOP0.PD.env.boxes, spheres=TOP0.PD.env.spheres, scp_params=sp, model_params=mp)
            if decomposition:
                bs.set_decomposition(decomposition)
        if not same_env:
            bs.set_env_batch([t.PD.env.boxes for t in TOPs[b0:b1]], [t.PD.env.spheres for t in TOPs[b0:b1]])
        bs.set_problems(x0[b0:b1], lo[b0:b1], hi[b0:b1], tf[b0:b1], X0[b0:b1], U0[b0:b1])
        if trajopt:
            bs.solve_async(max_iter)    # gusto_solve_trajopt_async: the shards of a TrajOpt batch run side by side too
        else:
            bs.solve_async(max_iter, force)
        shards.append((b0, b1, bs))
    out = [None] * B
    gathered = None
    if len(shards) > 1:
        # the final gather of the multi-GPU path below the host language: every shard to the first handle's GPU by one
        # direct peer copy over xGMI (gusto_gather_peer; it completes each shard's solve first), then ONE copy to the host
        gathered = shards[0][2].gather_peer([bs for _, _, bs in shards])
    for b0, b1, bs in shards:
        bs.wait()
        snap = _fetch(bs, None if gathered is None else (gathered[0][b0:b1], gathered[1][b0:b1]))
        per = bs.last_solve_ms() * 1e-3 / (b1 - b0)
        for b in range(b0, b1):
            SCPP = SCPProblem(TOPs[b])
            SCPS = SCPSolution(SCPP, inits[b])
            (_fill_trajopt_solution if trajopt else _fill_solution)(SCPS, SCPP, snap, b - b0, per)
            TOSs[b].traj, TOSs[b].SCPS = SCPS.traj, SCPS
            out[b] = SCPS
    return out


def solve_batch_sharded(model_id, N, x_init, goal_lo, goal_hi, tf, world_size, rank, device=0, max_iter=30, boxes=None,
                        spheres=None, group=None, solver=None):
    """The multi-GPU path of SURVEY.md 8(e) in one call (one process per GPU, launched by torch.distributed.run):
    rank r solves the contiguous block shard_bounds(B, G, r) of the batch on its own GPU -- independent problems, no
    data-path collective -- and the trajectories are gathered to rank 0 STRAIGHT FROM HBM (BatchSolver.traj_dev views,
    RCCL under the nccl backend), together with the per-problem status vectors.  Returns on rank 0 a dict with X, U
    (torch tensors on the gather device), iterations, converged, successful, stop_reason (numpy); None elsewhere.
    `solver` lets a caller keep a BatchSolver (and its device buffers) across calls."""
    x_init, goal_lo, goal_hi = (np.asarray(a, float) for a in (x_init, goal_lo, goal_hi))
    B = x_init.shape[0]
    lo, hi = shard_bounds(B, world_size, rank)
    tf = np.broadcast_to(np.asarray(tf, float), (B,))
    bs = solver or BatchSolver(model_id, N, max(1, hi - lo), hist_cap=_hist_cap(max_iter), device=device, boxes=boxes,
                               spheres=spheres)
    bs.set_problems(x_init[lo:hi], goal_lo[lo:hi], goal_hi[lo:hi], tf[lo:hi])
    bs.solve(max_iter)
    Xd, Ud = bs.traj_dev()
    st = bs.status()
    local = dict(X=Xd, U=Ud, iterations=st["iterations"].astype(np.int64), converged=st["converged"].astype(np.int64),
                 successful=st["successful"].astype(np.int64), stop_reason=st["stop_reason"].astype(np.int64))
    return gather_batch_results(local, world_size, rank, group)


def gather_batch_results(local, world_size, rank, group=None):
    """Final gather of per-rank results to rank 0: the only communication of a multi-GPU run.
    `local` maps names to arrays with the problem index leading: torch tensors ALREADY ON THE GPU (the views
    BatchSolver.traj_dev() returns -- they go to RCCL as they are, no host staging) or numpy arrays (gloo on CPU, or
    small host-side vectors, moved once).  Ranks may hold different numbers of problems.  Rank 0 gets the concatenated
    dict in the type it passed in, the other ranks None."""
    import torch
    import torch.distributed as dist
    import os
    if world_size == 1 and not os.environ.get("GUSTO_FORCE_GATHER"):   # (the variable makes a 1-rank run exercise the collectives)
        return local
    out = {}
    backend = dist.get_backend(group)
    dev = torch.device("cuda", torch.cuda.current_device()) if backend == "nccl" else torch.device("cpu")
    count = torch.tensor([next(iter(local.values())).shape[0]], dtype=torch.int64, device=dev)
    sizes = [torch.zeros(1, dtype=torch.int64, device=dev) for _ in range(world_size)]
    dist.all_gather(sizes, count, group=group)
    sizes = [int(s.item()) for s in sizes]
    mx = max(sizes)
    for k, v in local.items():
        as_numpy = not torch.is_tensor(v)
        t = (torch.from_numpy(np.ascontiguousarray(v)) if as_numpy else v).to(dev)
        if t.shape[0] != mx:                   # ragged tail rank: pad to the common block size
            pad = torch.zeros((mx,) + tuple(t.shape[1:]), dtype=t.dtype, device=dev)
            pad[:t.shape[0]] = t
            t = pad
        bufs = [torch.empty_like(t) for _ in range(world_size)] if rank == 0 else None
        dist.gather(t.contiguous(), bufs, dst=0, group=group)
        if rank == 0:
            cat = torch.cat([b[:s] for b, s in zip(bufs, sizes)], dim=0)
            out[k] = cat.cpu().numpy() if as_numpy else cat
    return out if rank == 0 else None
