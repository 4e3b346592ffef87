/*
 * gusto_oracle.h -- CPU ORACLE (test infrastructure, NOT the product).
 *
 * A plain-C, fp64, single-problem restatement of the GuSTO sequential convex
 * programming path of StanfordASL/GuSTO.jl (reference @ /root/reference):
 *   outer loop          src/scp/scp_gusto.jl:49-176, 316-343
 *   driver / metric     src/traj_opt.jl:47-85
 *   subproblem          src/scp/scp_gusto.jl:178-314 + per-model SCPConstraints
 *   models              src/dynamics/{freeflyer_se2,dubins_car,astrobee_se3,
 *                       astrobee_se3_manifold}.jl, src/dynamics.jl:24-81
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library.  The shipped HIP path never links or calls it.
 *
 * PARITY UNPINNED: the reference cannot run here (no Julia), has no tests and no
 * golden vectors; its two arithmetic dependencies are external and absent:
 *   (1) convex solve  : JuMP 0.19.2 -> Ipopt 0.5.4 / Gurobi 0.6.0 (Manifest.toml)
 *   (2) signed distance: BulletCollision.jl (unpinned)
 * (1) is restated as a primal-dual interior point method (any correct convex
 * solver agrees to tolerance on this strictly convex problem); (2) is restated
 * as analytic signed distances (disc/sphere vs AABB, sphere vs sphere).
 * The oracle is pinned instead by scipy SLSQP on the same subproblem, dense-KKT
 * known answers and KKT certificates (tests/test_oracle_*.py).
 */
#ifndef GUSTO_ORACLE_H
#define GUSTO_ORACLE_H

#ifdef __cplusplus
extern "C" {
#endif

#define GO_MAXN 13
#define GO_MAXM 6

enum { GO_FREEFLYER_SE2 = 0, GO_DUBINS_CAR = 1, GO_ASTROBEE_SE3 = 2, GO_ASTROBEE_SE3_MANIFOLD = 3 };

/* scp_status codes (Symbols in the reference, scp_gusto.jl:126-146) */
enum { GO_SCP_NA = 0, GO_SCP_OK = 1, GO_SCP_INACCURATE_MODEL = 2, GO_SCP_VIOLATES_CONSTRAINTS = 3,
       GO_SCP_TRUST_REGION_VIOLATED = 4 };
/* solver_status codes (MOI termination codes in the reference, scp_gusto.jl:106-111) */
enum { GO_SOLVER_NA = 0, GO_SOLVER_OPTIMAL = 1, GO_SOLVER_ALMOST = 2, GO_SOLVER_FAILED = 3 };
/* why the outer loop stopped */
enum { GO_STOP_MAXITER = 0, GO_STOP_CONVERGED = 1, GO_STOP_SUBPROBLEM_FAILED = 2, GO_STOP_OMEGA_MAX = 3 };

typedef struct {
    double Delta0, omega0, omega_max, eps, rho0, rho1, beta_succ, beta_fail, gamma_fail;
    double convergence_threshold;
} go_scp_params;

typedef struct {
    double mass, Jdiag[3], radius, clearance;
    double hard_limit_vel, hard_limit_accel, hard_limit_omega, hard_limit_alpha;
    double dubins_v, dubins_k, u_max, u_min;
    double x_max[GO_MAXN], x_min[GO_MAXN];
    int n_robot_comp;          /* convex robot components looped in the rho ratio */
    double comp_off[2][3];     /* their offsets in the robot frame */
} go_model_params;

typedef struct {
    double tol;        /* residual tolerance of the scaled subproblem            */
    double tol_acc;    /* acceptable tolerance at the iteration cap             */
    double mu_floor;   /* smallest complementarity target                        */
    double tr_tol;     /* slack on the `max||dx||^2 - Delta <= 0` post-check     */
    double mu_warm;    /* centred start at this mu once a subproblem of the same SCP run has been solved
                          (the iterate starts at the previous optimum); 0 = always the cold start; < 0 = the model's triple */
    int max_iter;
    int acc_iter;        /* ALMOST once tol_acc has held for this many consecutive iterations (0: only at the cap) */
    double mu_warm_gain; /* start level = min(max(mu_warm, mu_warm_max), max(mu_warm, mu_warm_gain * conv[end]^2)): it follows */
    double mu_warm_max;  /* the size of the last trajectory change (gusto_hip.h: gusto_ipm_opts)                              */
    double sigma_max;    /* upper bound of Mehrotra's centring parameter (<= 0: none)                                          */
} go_ipm_opts;

/* Study knob (tests/test_oracle_scp.py): alternative models of what the reference's external
 * collision library returns for the freeflyer.  kind 0 = the analytic planar distance the product uses. */
typedef struct {
    int kind;             /* 0 analytic (default), 1 prism-vs-AABB study model (freeflyer only)          */
    int n_poly;           /* >0: body is a regular n-gon prism with vertices on the circle; 0: exact disc  */
    int vertical_escape;  /* penetration depth may be the vertical separation (3-D minimum translation)    */
    double poly_phase, margin, z_lo, z_hi;
    int pen_mode;         /* 1: a penetrating pair returns pen_value instead of a depth                   */
    double pen_value, pen_band; /* pen_mode 3: |d| < pen_band, 4: -pen_band < d < 0 return pen_value     */
} go_dist_model;

typedef struct {
    double obj;        /* JuMP.objective_value: cost + all slacks (unscaled)     */
    double res_p, res_d, mu;
    int iters, status; /* GO_SOLVER_*                                            */
} go_sub_info;

/* SCPParam_TrajOpt (scp_trajopt.jl:3-30; per-model values freeflyer_se2.jl:49-64, astrobee_se3.jl:50-65) */
typedef struct {
    double mu0, s0, c, tau_plus, tau_minus, k, ftol, xtol, ctol;
    int max_penalty_iteration, max_convex_iteration, max_trust_iteration;
} go_trajopt_params;

typedef struct go_problem go_problem;

void go_default_params(int model, go_scp_params* sp, go_model_params* mp);
void go_default_ipm_opts(go_ipm_opts* o);
int  go_model_dims(int model, int* n, int* m);

go_problem* go_create(int model, int N, const go_scp_params* sp, const go_model_params* mp,
                      int n_box, const double* box_min_max, int n_sph, const double* sph_c_r);
void go_destroy(go_problem* p);
void go_set_ipm_opts(go_problem* p, const go_ipm_opts* o);
void go_set_distance_model(go_problem* p, const go_dist_model* dm);
/* record (traj_prev, subproblem optimum) of up to `cap` trips of the following go_solve calls (lock-step parity tests) */
void go_set_trace(go_problem* p, int cap);
int go_trace_len(const go_problem* p);
int go_get_trace(const go_problem* p, int t, double* Xp, double* Up, double* Xn, double* Un);

/* goal_lo==goal_hi -> hard equality, +-inf -> free, else hard box (BoxGoal).
 * X0/U0 NULL -> straight line init (freeflyer_se2.jl:97-111). Resets histories. */
int go_set_problem(go_problem* p, const double* x_init, const double* goal_lo, const double* goal_hi,
                   double tf, const double* X0, const double* U0);

/* scp_gusto.jl:49-176. Re-entrant: a second call resumes (iter_cap = iterations + max_iter). */
int go_solve(go_problem* p, int max_iter, int force);

/* results */
int go_get_traj(const go_problem* p, double* X, double* U);
int go_get_status(const go_problem* p, int* iterations, int* converged, int* successful, int* stop_reason,
                  int* total_ipm_iters);
int go_hist_len(const go_problem* p);   /* iterations+1 entries in per-iteration vectors */
/* each out array may be NULL; lengths: see SCPSolution (types.jl:150-173). J_true/J_full have one extra
 * leading entry per go_solve call, as in the reference (scp_gusto.jl:73-74). */
int go_get_history(const go_problem* p, double* J_true, int* nJ_true, double* J_full, int* nJ_full,
                   double* conv, double* Delta, double* omega, double* rho, int* n_rho,
                   int* accept, int* scp_status, int* solver_status, int* tr_sat, int* cvx_sat,
                   int* ipm_iters);
int go_get_dual(const go_problem* p, double* dual);

/* Indirect shooting seeded by the SCP dual (shooting.jl:4-66; DubinsCar only, dubins_car.jl:259-280): RK4 with
 * `substeps` steps per knot interval, Newton on F(p0) = x_goal - x(tf; p0) with a forward-difference Jacobian.
 * Returns 1 (:Optimal, |F|_inf <= ftol), 0 (:Diverged), -1 (model without a shooting ODE).  p0 NULL -> SCPS.dual. */
int go_shoot(go_problem* p, const double* p0, int substeps, int max_newton, double ftol, double* p_out, double* X, double* U,
             int* newton_iters, double* resid);

/* pieces exposed for tests --------------------------------------------------------------- */
/* one convex subproblem (scp_gusto.jl:178-314) around (Xp,Up) */
int go_subproblem(go_problem* p, const double* Xp, const double* Up, double Delta, double omega,
                  double toggle_dist, double* Xn, double* Un, double* dual, go_sub_info* info);
/* row dump of the last assembled subproblem: returns number of rows */
int go_rows_count(const go_problem* p);
int go_rows_get(const go_problem* p, int i, int* k, int* isu, int* kind, int* nnz, int* idx,
                double* a, double* v0, double* b, double* c0, double* mul, double* off,
                double* slack, double* lam);
void go_dynamics(const go_problem* p, const double* x, const double* u, double* f, double* A, double* B);
/* signed distance of robot component comp placed at workspace location r to obstacle i */
double go_signed_distance(const go_problem* p, int comp, const double* r, int i, double* nhat);
double go_trust_region_ratio(go_problem* p, const double* X, const double* U, const double* Xp, const double* Up);
double go_cost_true(const go_problem* p, const double* U);
double go_convergence_metric(const go_problem* p, const double* X, const double* Xp);
int go_convex_ineq_satisfied(go_problem* p, const double* X, const double* Xp, const double* Up, double toggle_dist);
void go_init_straightline(const go_problem* p, double* X, double* U);

/* ---- TrajOpt (src/scp/scp_trajopt.jl), the second SCP algorithm behind the same solve_method! seam -------------------
 * FreeflyerSE2 and AstrobeeSE3.  A TrajOpt problem keeps (u_k, d_k) per knot: U arrays have m + n columns, the last n being
 * the defect of the interval (k, k+1) (the L1-penalised dynamics); go_model_dims still reports the model's m.
 * What the file means where it cannot run as written is listed in DESIGN.md section 4 (intended L1 dynamics penalty,
 * hard x_1 = x_init, index typos of trust_region_ratio_trajopt, the dropped last class of evaluate_ctol, max_iter). */
void go_default_trajopt_params(int model, go_trajopt_params* tp);
go_problem* go_create_trajopt(int model, int N, const go_model_params* mp, const go_trajopt_params* tp,
                              int n_box, const double* box_min_max, int n_sph, const double* sph_c_r);
/* one convex subproblem (:159-279) around (Xp, Up) with penalty mu and trust region s */
int go_trajopt_subproblem(go_problem* p, const double* Xp, const double* Up, double mu, double s, double* Xn, double* Un,
                          double* dual, go_sub_info* info);
/* solve_trajopt_jump! (:33-157); max_iter caps the number of subproblem solves (the reference computes iter_cap and never
 * uses it).  Returns the number of solves. */
int go_solve_trajopt(go_problem* p, int max_iter);
/* rho_vec / s_vec (one leading entry + one per solve), mu_vec, xtol_vec, ftol_vec, ctol_vec; any pointer may be NULL */
int go_get_trajopt_history(const go_problem* p, double* s_vec, int* n_s, double* mu_vec, int* n_mu, double* xtol_vec, int* n_xtol,
                           double* ftol_vec, int* n_ftol, double* ctol_vec, int* n_ctol);
double go_trajopt_ratio(go_problem* p, const double* X, const double* U, const double* Xp, const double* Up);
double go_trajopt_ctol(go_problem* p, const double* X, const double* U, const double* Xp, const double* Up);

/* batch driver for the CPU baseline: B independent problems, OpenMP over problems.
 * x_init/goal_lo/goal_hi: [B][n]; tf: [B]; X,U out: [B][N][n], [B][N][m]; flags out [B]. */
int go_solve_batch(int model, int N, const go_scp_params* sp, const go_model_params* mp,
                   int n_box, const double* box_min_max, int n_sph, const double* sph_c_r,
                   int B, const double* x_init, const double* goal_lo, const double* goal_hi,
                   const double* tf, int max_iter, int nthreads,
                   double* X, double* U, int* converged, int* successful, int* iterations, int* ipm_iters);

#ifdef __cplusplus
}
#endif
#endif
