/*
 * gusto_hip.h -- C ABI of libgusto_hip.so: batched GuSTO sequential convex programming on MI355X.
 *
 * One call solves a batch of independent SCP problems of one model type on one GPU.  The library
 * replaces the reference's `solve_method!` plug-in for the GuSTO path,
 *     solve_gusto_jump!(SCPS, SCPP, solver, max_iter, force; kw...)      src/scp/scp_gusto.jl:49
 * which is what `solve_SCP!` calls through its function argument           src/traj_opt.jl:47-72
 * Each entry point names the reference interface it stands in for.
 *
 * Conventions: every function returns 0 on success or a negative gusto_rc; per-problem failures are
 * data (status arrays), never a failing return code.  All pointers are HOST memory unless the name ends
 * in `_dev`.  Layouts are the reference's Julia column-major X[n,N], U[m,N] per problem, problem index
 * slowest: X[b][k][i].  A handle is not thread-safe; use one handle per host thread / GPU.
 */
#ifndef GUSTO_HIP_H
#define GUSTO_HIP_H

#ifdef __cplusplus
extern "C" {
#endif

#define GUSTO_MAXN 13
#define GUSTO_MAXM 6

/* typeof(SCPP.PD.model) -> id (src/dynamics/{freeflyer_se2,dubins_car,astrobee_se3,astrobee_se3_manifold}.jl) */
enum gusto_model_id {
    GUSTO_FREEFLYER_SE2 = 0,
    GUSTO_DUBINS_CAR = 1,
    GUSTO_ASTROBEE_SE3 = 2,
    GUSTO_ASTROBEE_SE3_MANIFOLD = 3
};

enum gusto_rc {
    GUSTO_OK = 0,
    GUSTO_ERR_ARG = -1,       /* bad argument / unsupported size            */
    GUSTO_ERR_HIP = -2,       /* HIP runtime error, see gusto_last_error()  */
    GUSTO_ERR_STATE = -3,     /* call order (e.g. solve before set_problems) */
    GUSTO_ERR_NO_DEVICE = -4  /* no usable GPU: there is no CPU fallback     */
};

/* SCPS.scp_status entries (Symbols in scp_gusto.jl:126-146) */
enum { GUSTO_SCP_NA = 0, GUSTO_SCP_OK = 1, GUSTO_SCP_INACCURATE_MODEL = 2, GUSTO_SCP_VIOLATES_CONSTRAINTS = 3,
       GUSTO_SCP_TRUST_REGION_VIOLATED = 4 };
/* SCPS.solver_status entries (MOI termination codes accepted/rejected at scp_gusto.jl:106-111) */
enum { GUSTO_SOLVER_NA = 0, GUSTO_SOLVER_OPTIMAL = 1, GUSTO_SOLVER_ALMOST = 2, GUSTO_SOLVER_FAILED = 3 };
/* why the outer loop of a problem stopped */
enum { GUSTO_STOP_MAXITER = 0, GUSTO_STOP_CONVERGED = 1, GUSTO_STOP_SUBPROBLEM_FAILED = 2, GUSTO_STOP_OMEGA_MAX = 3,
       GUSTO_STOP_HIST_FULL = 4 /* a history vector reached hist_cap before iter_cap: create the handle with more */ };

/* SCPParam + SCPParam_GuSTO (types.jl:65-73, scp_gusto.jl:4-24; per-model values e.g. freeflyer_se2.jl:22-39) */
typedef struct {
    double Delta0, omega0, omega_max, eps, rho0, rho1, beta_succ, beta_fail, gamma_fail;
    double convergence_threshold;
} gusto_scp_params;

/* robot + model scalars (robot/freeflyer.jl:28-62, robot/astrobee3D.jl:15-33, dubins_car.jl:22-33) */
typedef struct {
    double mass, Jdiag[3], radius, clearance;
    double hard_limit_vel, hard_limit_accel, hard_limit_omega, hard_limit_alpha;
    double dubins_v, dubins_k, u_max, u_min;
    double x_max[GUSTO_MAXN], x_min[GUSTO_MAXN];
    int n_robot_comp;      /* convex robot components looped by trust_region_ratio_gusto */
    double comp_off[2][3]; /* their offsets in the robot frame                            */
} gusto_model_params;

/* inner convex solver (stands in for the `kwarg...` forwarded to the optimizer, scp_gusto.jl:82-92) */
/* tr_tol: slack of trust_region_satisfied_gusto.  The reference tests `max_k ||x_k - xp_k||^2 - Delta <= 0` (scp_gusto.jl:34-44)
 * on the optimum its solver returns; whenever the trust region row is ACTIVE the left-hand side is zero up to the solver's
 * accuracy, so the literal test is decided by the last digits of Ipopt / Gurobi (or of the interior point method here).
 * This library evaluates `<= tr_tol * max(1, Delta)` with the DEFAULT tr_tol = 1e-6 (100 x the primal tolerance `tol`), which
 * makes the verdict independent of solver noise; tr_tol = 0 selects the literal test (1 of the 44 844 accept / reject decisions
 * of the 4096-problem freeflyerSE2 batch changes; tests/test_gpu_parity.py runs both settings against the oracle). */
typedef struct {
    double tol, tol_acc, mu_floor /* smallest complementarity target; < 0 (default): 1e-10 for astrobeeSE3manifold, 1e-11 otherwise */, tr_tol;
    double mu_warm; /* complementarity of the centred start used from the second subproblem of an SCP run on (the
                       iterate then starts at the previous optimum); 0 = always the cold start; < 0 (the default) = the model's
                       own triple (mu_warm, mu_warm_gain, mu_warm_max), see below */
    int max_iter;
    int acc_iter;   /* stop with ALMOST_LOCALLY_SOLVED once the acceptable level tol_acc has held for this many consecutive
                       iterations without reaching tol (Ipopt's acceptable_iter; the reference takes MOI.ALMOST_LOCALLY_SOLVED as
                       solved, scp_gusto.jl:107).  0 = only at the iteration cap.  An interior point solve that cycles at the 1e-6
                       level (seen on astrobeeSE3manifold: a period-4 cycle of mu between 3e-10 and 3e-9) then costs 10 extra
                       iterations instead of running to the cap of 60 */
    /* The start level follows the size of the last trajectory change: a subproblem whose linearisation point moved far from
     * the previous one is started further from the boundary,
     *     mu_start = min(max(mu_warm, mu_warm_max), max(mu_warm, mu_warm_gain * c^2)),   c = convergence_measure[end]
     * (traj_opt.jl:74-85, the relative change of the trajectory in the previous SCP iteration).  mu_warm_gain = 0: the fixed
     * level mu_warm.  Model defaults (measured, tools/ipm_opts_scan.py; an internal heuristic of the interior point method, the
     * optimum it converges to is the same): freeflyerSE2 (1e-4, 0.1, 1e-2), dubins_car (1e-9, 0, -), astrobeeSE3
     * (1e-6, 1, 1e-2), astrobeeSE3manifold (1e-4, 1, 1e-2). */
    double mu_warm_gain, mu_warm_max;
    /* Upper bound of Mehrotra's centring parameter sigma = (mu_aff / mu)^3: the corrector never aims at less than a
     * (1 - sigma_max) reduction of the complementarity.  Without the bound (sigma_max >= 1) a predictor that makes no progress
     * sets sigma ~ 1, and the method can then cycle near the solution (mu between 3e-10 and 3e-9 with period 4, traced on
     * astrobeeSE3manifold) until the iteration cap, or fail.  Default (< 0): 0.1 for gusto_solve, none for gusto_solve_trajopt
     * (whose subproblems were validated with the unbounded rule); measured on the BASELINE batches: SubproblemFailed
     * 23 -> 1 of 4096 (freeflyerSE2), 22 -> 0 of 8192 (astrobeeSE3), 10 -> 7 of 2048 (manifold), and 1-3 % fewer KKT solves. */
    double sigma_max;
} gusto_ipm_opts;

typedef struct gusto_handle_s* gusto_handle;

/* fills the per-model defaults: SCPParam(model, fft), SCPParam_GuSTO(model), robot constructors */
int gusto_default_params(int model_id, gusto_scp_params* sp, gusto_model_params* mp);
int gusto_default_ipm_opts(gusto_ipm_opts* o);
int gusto_model_dims(int model_id, int* x_dim, int* u_dim);

/* SCPProblem(TOP) for a batch (types.jl:256-259): allocates all device memory for up to batch_cap problems of
 * N knots and hist_cap SCP iterations of history.  device = HIP device ordinal. */
int gusto_create(gusto_handle* h, int model_id, int N, int batch_cap, int hist_cap, int device);
int gusto_destroy(gusto_handle h);
const char* gusto_last_error(gusto_handle h);

int gusto_set_params(gusto_handle h, const gusto_scp_params* sp, const gusto_model_params* mp);
int gusto_set_ipm_opts(gusto_handle h, const gusto_ipm_opts* o);
/* Workspace(robot, env) (types.jl:12-24): keep-out set = keepout_zones then obstacle_set, as AABBs
 * (min xyz, max xyz) followed by spheres (centre xyz, radius) */
int gusto_set_env(gusto_handle h, int n_box, const double* box_min_max, int n_sph, const double* sph_c_r);
/* One Workspace PER PROBLEM -- in the reference every ProblemDefinition owns its env (types.jl:32-39) and its
 * Workspace(robot, env) (types.jl:12-24), so a batch may mix obstacle layouts.  Problem b has n_box[b] AABBs and n_sph[b]
 * spheres (at most 64 components together); box_min_max / sph_c_r hold the tables of all B problems one after the other
 * in problem order ([sum n_box][6], [sum n_sph][4]).  B must equal the B of gusto_set_problems by the time of the solve
 * (either call may come first).  gusto_set_env returns the handle to one shared keep-out set. */
int gusto_set_env_batch(gusto_handle h, int B, const int* n_box, const double* box_min_max, const int* n_sph,
                        const double* sph_c_r);
/* Longest-first schedule of a gusto_solve call (new; affects time only, results are bit-identical).  gusto_solve is ONE
 * launch of persistent workgroups that pull work from a device-side scheduler.  In batches of at least `min_batch`
 * problems the first `probe_iters` time slices of a problem are one SCP iteration each; between slices the problem
 * waits in the list of its penalty level (number of omega raises so far -- the problems whose omega was raised early are
 * the long ones); workgroups take the highest raised level waiting, then fresh problems (the ones that start deepest
 * inside an obstacle first), then level 0; from slice `probe_iters` on a problem runs to its end (freeflyerSE2 problems of
 * level 0: in slices of four iterations).  probe_iters = 0: first come, first served.  Without this call: every batch that
 * does not fit the GPU's resident workgroups at once (and any batch of >= 2048 problems), 2 probing slices for freeflyerSE2,
 * 1 for the other models.  Results do not depend on the schedule. */
int gusto_set_schedule(gusto_handle h, int probe_iters, int min_batch);
/* How a gusto_solve maps problems to the GPU (new; affects time only -- both kernels run scp_gusto.jl:49-176 on the same
 * subproblem, scp_gusto.jl:178-314, to the same tolerances).  WAVE: one wavefront per problem, lane k = knot k, the
 * Newton system staged in LDS (every model).  LANE: one LANE per problem, 64 problems per wavefront, the Riccati recursion
 * in that lane's registers and its per-knot data streamed through HBM in a [wave][knot][entry][lane] layout -- built for
 * dubins_car (BASELINE.json configs[2]: n = 3, N = 30, batches of 65 536, where a wave per problem leaves 34 of 64 lanes
 * idle and spends its time on 3 x 3 blocks); other models ignore the setting.  AUTO (default) = WAVE: as measured on MI355X
 * the LANE kernel is correct (same trips, statuses and trajectories to 1e-12) but slower at the BASELINE batch sizes, because a
 * wavefront runs as long as the longest of its 64 problems.  The scheduler of gusto_set_schedule belongs
 * to the WAVE kernel.
 * WAVE2 / WAVE4 (round 6; astrobeeSE3 and astrobeeSE3manifold, N <= 64): two or four wavefronts per problem -- the horizon split into
 * as many Riccati chains, a wave each, joined by coarse LQR stages, and each knot's obstacle rows shared between the waves
 * (csrc/segw.hpp).  For batches that leave SIMDs without a wave: AUTO takes WAVE4 up to six problems per CU, WAVE2 up to sixteen
 * (astrobeeSE3manifold: eight), one wave per problem beyond (measured on MI355X); WAVE forces one wave per problem.  Same subproblems to the same tolerances;
 * the iterates differ in rounding (the KKT solve is reassociated), the SCP iteration counts do not on the test batches.  A model
 * or horizon without these kernels answers GUSTO_ERR_ARG at gusto_solve. */
enum { GUSTO_DECOMP_AUTO = 0, GUSTO_DECOMP_WAVE = 1, GUSTO_DECOMP_LANE = 2, GUSTO_DECOMP_WAVE2 = 3, GUSTO_DECOMP_WAVE4 = 4 };
int gusto_set_decomposition(gusto_handle h, int decomposition);
/* run on a caller-owned hipStream_t (NULL = a new stream owned by the handle).  Like every setter it first completes
 * a pending gusto_solve_async on the stream that solve was enqueued on. */
int gusto_set_stream(gusto_handle h, void* hip_stream);

/* ProblemDefinition + init trajectory + SCPSolution(SCPP, traj_init) for B problems (types.jl:32-39,233):
 * goal_lo == goal_hi -> PointGoal row, finite lo < hi -> BoxGoal rows, +-Inf -> coordinate has no goal.
 * X0/U0 == NULL -> init_traj_straightline (freeflyer_se2.jl:97-111).  Resets every history. */
int gusto_set_problems(gusto_handle h, int B, const double* x_init, const double* goal_lo, const double* goal_hi,
                       const double* tf, const double* X0, const double* U0);
/* same, inputs already resident in HBM on the handle's device */
int gusto_set_problems_dev(gusto_handle h, int B, const double* x_init_dev, const double* goal_lo_dev,
                           const double* goal_hi_dev, const double* tf_dev, const double* X0_dev,
                           const double* U0_dev);

/* solve_gusto_jump!(SCPS, SCPP, solver, max_iter, force) for the whole batch (scp_gusto.jl:49-176).
 * Synchronous.  Re-entrant: a second call resumes every problem (iter_cap = iterations + max_iter, :67). */
int gusto_solve(gusto_handle h, int max_iter, int force);
/* The same solve, enqueued on the handle's stream without blocking the host; gusto_wait (or any getter, setter or
 * solve on the handle) completes it.  New: the reference is blocking.  Two handles used alternately keep the GPU
 * full across consecutive batches -- the tail of one batch (its slowest problems) overlaps the head of the next. */
int gusto_solve_async(gusto_handle h, int max_iter, int force);
/* Which problems of the batch the following gusto_solve / gusto_solve_async / gusto_shoot calls work on: active [B], nonzero =
 * iterate, NULL = all again (the state after gusto_set_problems).  The reference's drivers loop per problem --
 * solve_SCPshooting! takes another SCP iteration and another shooting attempt only `while !SCPS.converged &&
 * SCPS.iterations < max_iter` (src/traj_opt.jl:23) -- and a batch needs that condition per problem: an inactive problem
 * keeps its trajectory, histories and counters untouched.  Not for TrajOpt handles nor the lane-per-problem decomposition. */
int gusto_set_active(gusto_handle h, const int* active);
int gusto_wait(gusto_handle h);
/* GPU time of the last gusto_solve, measured with HIP events on the stream the kernel ran on */
int gusto_last_solve_ms(gusto_handle h, double* ms);

/* SCPS.traj (X,U) -- TOS.traj aliases it (traj_opt.jl:58) */
int gusto_get_traj(gusto_handle h, double* X, double* U);
int gusto_get_traj_dev(gusto_handle h, const double** X_dev, const double** U_dev);
/* Final gather of a multi-GPU run from ONE host process (one handle per GPU, shards enqueued with gusto_solve_async): completes
 * every source's solve and copies the shards, in the order of `src`, into buffers on the GPU of `dst` -- one direct peer copy
 * per shard over xGMI (fan-in); `dst` may itself be one of the sources.  Outputs (any may be NULL): device
 * pointers to X [B_total][N][x_dim] and U [B_total][N][u_dim] on dst's GPU (valid until the next gather on dst), host copies
 * of the same, the number of problems.  New: the reference is one process, one problem.  (Between processes -- one rank per
 * GPU -- the same gather runs over RCCL from the views of gusto_get_traj_dev: gusto.jl_amd/host.py gather_batch_results.) */
int gusto_gather_peer(gusto_handle dst, int n_src, const gusto_handle* src, const double** X_dev, const double** U_dev,
                      double* X_host, double* U_host, int* B_total);
/* SCPS.iterations / converged / successful per problem, plus stop reason and inner iteration count */
int gusto_get_status(gusto_handle h, int* iterations, int* converged, int* successful, int* stop_reason,
                     int* ipm_iters);
/* SCPS.dual = -dual(init rows) (freeflyer_se2.jl:486-489): [B][n] */
int gusto_get_dual(gusto_handle h, double* dual);

/* SCPSolution / SCPParam_GuSTO history vectors (types.jl:150-173, scp_gusto.jl:15-19) as [B][hist_cap]
 * arrays; n_hist[b] entries are valid in the per-iteration vectors, nJ[b] in J_true/J_full and n_rho[b] in
 * rho (those get one extra leading entry per gusto_solve call, scp_gusto.jl:73-75).  Any pointer may be NULL. */
typedef struct {
    int hist_cap; /* IN: row capacity of the arrays below, >= the handle's (gusto_get_hist_cap); else GUSTO_ERR_ARG */
    int *n_hist, *nJ, *n_rho;
    double *J_true, *J_full, *convergence_measure, *Delta, *omega, *rho;
    int *accept_solution, *scp_status, *solver_status, *trust_region_satisfied, *convex_ineq_satisfied, *ipm_iters;
} gusto_history;
int gusto_get_history(gusto_handle h, gusto_history* out);
int gusto_get_hist_cap(gusto_handle h, int* hist_cap);
/* When a problem stops with GUSTO_STOP_SUBPROBLEM_FAILED the reference has pushed one more solver_status entry than
 * any other vector (scp_gusto.jl:106): it is solver_status[b][n_hist[b]]. */

/* SCPParam_GuSTO supplied by the caller -- scp_gusto.jl:60 keeps a `param.alg` that is already defined, so a user
 * can start from her own Delta_vec[end] / omega_vec[end].  Overwrites the last Delta / omega history entry of every
 * problem (after gusto_set_problems: the initial Delta0 / omega0).  Either pointer may be NULL.  [B] each. */
int gusto_set_trust_state(gusto_handle h, const double* Delta, const double* omega);

/* Indirect shooting seeded by the SCP dual: solve!(SS::ShootingSolution, SP::ShootingProblem) (src/shooting.jl:4-49)
 * for every problem of the batch, the refinement step of solve_SCPshooting! (src/traj_opt.jl:4-45).  The two models
 * that have a shooting ODE in the reference: DubinsCar (shooting_ode! / get_control, dubins_car.jl:259-280) and
 * AstrobeeSE3Manifold (dynamics_shooting! / shooting_ode! / get_control, astrobee_se3_manifold.jl:831-895, 26 states +
 * costates); other models return GUSTO_ERR_ARG.  The reference's ODE and
 * nonlinear solvers (DifferentialEquations, NLsolve) are external and absent: the scheme is RK4 with `substeps` steps
 * per knot interval and Newton with a forward-difference Jacobian on F(p0) = x_goal - x(tf; p0), |F|_inf <= ftol.
 * p0: [B][n] host seeds, or NULL = SCPS.dual of every problem (what ShootingProblem(TOP, SCPS) takes, types.jl:219-227). */
typedef struct {
    int substeps;    /* RK4 steps per knot interval (default 4)             */
    int max_newton;  /* nlsolve(..., iterations = 100, ...)  shooting.jl:14 */
    double ftol;     /* nlsolve(..., ftol = 1e-3)            shooting.jl:14 */
    int no_group_pass;  /* 0 (default, also what a zero- or brace-initialised struct gives): problems still iterating after 8
                         * Newton steps go on with 16 lanes each; 1: a lane per problem throughout -- the same results bit for
                         * bit, slower for the stragglers (tests compare the two).  Any other value: GUSTO_ERR_ARG */
} gusto_shoot_opts;
int gusto_default_shoot_opts(gusto_shoot_opts* o);
int gusto_shoot(gusto_handle h, const double* p0, const gusto_shoot_opts* opts);
/* status [B]: 1 = :Optimal (sol_newton.f_converged), 0 = :Diverged; p0 [B][n] the converged initial costate; X [B][N][n],
 * U [B][N][m] the recovered trajectory of the :Optimal problems (shooting.jl:26-36).  Any pointer may be NULL. */
int gusto_get_shoot(gusto_handle h, int* status, int* newton_iters, double* resid, double* p0, double* X, double* U);

/* One convex subproblem per problem (what scp_gusto.jl:82-104 builds and solves in one trip), linearised at
 * (Xp,Up)[b] with the given Delta/omega/obstacle_toggle_distance[b].  Used by the parity tests.
 * Outputs: Xn,Un [B][N][.], obj [B] (JuMP.objective_value), status [B] (GUSTO_SOLVER_*), iters [B]. */
int gusto_subproblem(gusto_handle h, int B, const double* Xp, const double* Up, const double* Delta,
                     const double* omega, const double* toggle, double* Xn, double* Un, double* obj, int* status,
                     int* iters);

/* ---- TrajOpt: solve_trajopt_jump!(SCPS, SCPP, solver, max_iter, force) (src/scp/scp_trajopt.jl:33-157), the second SCP
 * algorithm the reference passes through the same `solve_method!` argument of solve_SCP! (src/traj_opt.jl:47-72) ----------
 * FreeflyerSE2, AstrobeeSE3 and AstrobeeSE3Manifold (the models of this library with a SCPParam_TrajOpt: freeflyer_se2.jl:49-64,
 * astrobee_se3.jl:50-65, astrobee_se3_manifold.jl:56-70; the manifold model has no trust region row, and its quaternion
 * norm row -- a convex_state_eq row, hard in TrajOpt, scp_trajopt.jl:200-208 -- is carried as the hard band |h| <= 1e-4).  A TrajOpt handle is a gusto_handle created by gusto_create_trajopt: gusto_set_env,
 * gusto_set_problems(_dev), gusto_get_traj, gusto_get_status, gusto_get_dual, gusto_last_solve_ms work on it unchanged
 * (U has the model's u_dim columns on the host side).  Where the file cannot run as written the math it states is built;
 * the list is in DESIGN.md section 4 (intended L1 dynamics penalty, hard x_1 = x_init, index typos of
 * trust_region_ratio_trajopt, the class evaluate_ctol drops, max_iter as a cap on the subproblem solves). */
/* SCPParam_TrajOpt (scp_trajopt.jl:3-30) */
typedef struct {
    double mu0, s0, c, tau_plus, tau_minus, k, ftol, xtol, ctol;
    int max_penalty_iteration, max_convex_iteration, max_trust_iteration;
} gusto_trajopt_params;
int gusto_default_trajopt_params(int model_id, gusto_trajopt_params* tp);
/* SCPProblem(TOP) + SCPParam_TrajOpt(model) for a batch; hist_cap >= 2 * max_penalty * max_convex * max_trust + 8 entries */
int gusto_create_trajopt(gusto_handle* h, int model_id, int N, int batch_cap, int hist_cap, int device);
int gusto_set_trajopt_params(gusto_handle h, const gusto_trajopt_params* tp);
/* the whole three-loop schedule (penalty mu x k, convex iterations, trust region s x tau+-) for every problem of the batch;
 * max_iter caps the number of convex subproblems per problem (the reference computes iter_cap and never reads it) */
int gusto_solve_trajopt(gusto_handle h, int max_iter);
/* ... enqueued on the handle's stream like gusto_solve_async: returns once the one launch of the batch is queued; gusto_wait
 * or any getter completes it.  One handle per GPU (SURVEY.md 8(b) threading row): the shards of a multi-GPU TrajOpt batch,
 * or two handles on one GPU, run side by side instead of one after the other. */
int gusto_solve_trajopt_async(gusto_handle h, int max_iter);
/* SCPParam_TrajOpt vectors as [B][hist_cap] arrays with their lengths [B]: rho_vec and s_vec have 1 + iterations entries,
 * J_true 1 + iterations, J_full / convergence_measure / solver_status iterations (solver_status, ipm_iters and
 * convergence_measure start at row 1 like the GuSTO histories).  Any pointer may be NULL. */
typedef struct {
    int hist_cap; /* IN: row capacity of the arrays below, >= the handle's */
    int *n_solves, *n_mu, *n_xtol, *n_ftol, *n_ctol;
    double *rho_vec, *s_vec, *mu_vec, *xtol_vec, *ftol_vec, *ctol_vec, *J_true, *J_full, *convergence_measure;
    int *solver_status, *ipm_iters;
} gusto_trajopt_history;
int gusto_get_trajopt_history(gusto_handle h, gusto_trajopt_history* out);
/* One TrajOpt subproblem (:159-279) per problem around (Xp, Up)[b] with penalty mu[b] and trust region s[b] (parity tests).
 * Up, Un: [B][N][u_dim]; Dn (may be NULL): the defect variables of the optimum, [B][N][x_dim]. */
int gusto_subproblem_trajopt(gusto_handle h, int B, const double* Xp, const double* Up, const double* mu, const double* s,
                             double* Xn, double* Un, double* Dn, double* obj, int* status, int* iters);

/* Development hook (libraries built with -DGUSTO_PROFILE only, otherwise GUSTO_ERR_STATE): per-problem cycle counters of
 * the kernel's phases, [B][48] (tools/gpu_prof.py).  Stands in for SCPS.iter_elapsed_times at a finer grain. */
int gusto_dev_get_prof(gusto_handle h, long long* out);
/* Development hook: shape of the last gusto_solve / gusto_subproblem launch -- resident (persistent) workgroups, dynamic
 * LDS bytes per workgroup, workgroups per CU.  GUSTO_ERR_STATE before the first launch.  (The freeflyerSE2 N = 50 kernel
 * is tuned to 4 problems per CU: 40 664 B of the 160 KiB; tests/test_gpu_parity.py guards it.) */
int gusto_dev_launch_info(gusto_handle h, int* slots, int* lds_bytes, int* per_cu);
/* Development hook: bytes of the handle's interior point workspace in HBM (resident workgroups x per-slot workspace; what the
 * kernel re-reads between the phases of a KKT solve -- bench.py sets it against the 256 MiB Infinity Cache next to the
 * L2 <-> fabric traffic it reports).  No counterpart in the reference (JuMP's model memory). */
int gusto_dev_workspace_bytes(gusto_handle h, long long* bytes);

#ifdef __cplusplus
}
#endif
#endif
