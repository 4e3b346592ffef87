/*
 * gusto_oracle.c -- CPU ORACLE (test infrastructure, NOT the product).  See gusto_oracle.h.
 *
 * Restates, for one problem at a time and in scalar fp64 C:
 *   - the GuSTO outer loop                       (scp_gusto.jl:49-176)
 *   - the convex subproblem it builds each trip  (scp_gusto.jl:178-314)
 *   - the per-model math the loop calls          (src/dynamics/ models, src/dynamics.jl)
 * The JuMP + Ipopt/Gurobi solve (scp_gusto.jl:104) is restated as a Mehrotra
 * predictor-corrector interior point method whose Newton systems are solved by a
 * Riccati recursion over the trapezoid-collocation structure; BulletCollision.distance
 * is restated as analytic signed distances.  "parity unpinned" -- see the header.
 */
#define _GNU_SOURCE
#include "gusto_oracle.h"
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#ifdef _OPENMP
#include <omp.h>
#endif

#define NX GO_MAXN
#define NU (GO_MAXM + GO_MAXN)   /* TrajOpt: controls (u, d), d = the n defect variables of a knot */
#define NZ (GO_MAXN + GO_MAXM + GO_MAXN)
/* TrajOpt: the defect variables carry the vanishing quadratic cost REG * dt * |d_k|^2 next to their L1 penalty mu |d_k|_1: an
 * L1 term alone leaves a defect that is off its kink without curvature, the Newton systems of the interior point method
 * then lose the goal rows at complementarity 1e-9 (LP-like degeneracy).  DESIGN.md section 4. */
#define GO_TRAJOPT_DEFECT_REG 1e-4
/* TrajOpt keeps convex_state_eq rows HARD (`== 0`, scp_trajopt.jl:200-208).  The only such row of the library's models is the
 * manifold model's linearised quaternion norm, one row per knot on x_k alone; the interior point method carries it as the
 * hard band |h_k| <= 1e-4 (two hard inequality rows): the width the notebook itself uses when it writes the equality on the
 * goal quaternion as a BoxGoal of +-1e-4 (examples/astrobeeSE3manifold.ipynb cell 1).  1e-6 was tried: the barrier weights
 * lambda / t ~ 1e8 on the quaternion block then break the Cholesky of the condensed stage Hessians in 2 of 3 problems. */
#define GO_TRAJOPT_EQ_BAND 1e-4   /* (rounds 3-5: the equality as a band; kept for the record, unused) */
/* Round 6: a convex_state_eq row is an EQUALITY ROW of the interior point method -- h(x_k) = 0 with a multiplier eta of either sign
 * and the constraint regularisation delta of a primal-dual method (Ipopt's delta_c): the Newton system carries  grad h' dx -
 * delta d_eta = -h, eliminated per row like every other row (H += grad grad' / delta, coefficient eta + h / delta, d_eta =
 * (h + grad' dx) / delta).  No slack, no barrier, no step-length limit, not part of the complementarity measure; |h| is part of
 * the primal residual, so a solve that stops OPTIMAL holds the equality to the 1e-8 stopping tolerance.  The band had no such
 * property: inside it the row was inactive, at its edge its barrier weights (~1e8) decided 3 % of whole runs in the last digits. */
#define GO_EQ_DELTA 1e-8
#define ROW_HARD 0     /* hard inequality  (scp_gusto.jl:213-221, 236-245)                  */
#define ROW_PEN 1      /* L1-penalised state inequality, part of the post-check (:281-295)  */
#define ROW_PEN_TR 2   /* L1-penalised trust region (:265-279), not part of the post-check  */
#define ROW_PEN_EQ 3   /* j=2 half of a penalised equality (:297-311), |h|<eps post-check   */
#define ROW_HARD_EQ 4  /* j=1 half of a penalised equality: 0 <= s1 <= w*h+eps, s1 -> 0     */
#define ROW_EQ 5       /* hard equality h = 0 (TrajOpt's convex_state_eq rows, scp_trajopt.jl:200-208) */

typedef struct {
    int k, isu, kind, nnz;
    int idx[NX];
    double a[NX], v0[NX], b[NX], c0; /* raw g(v) = sum a_j (v_j - v0_j)^2 + sum b_j v_j + c0 */
    double mul, off;                 /* scaled row used by the IPM: ghat = mul*g - off       */
} go_row;

struct go_problem {
    int model, n, m, N;
    go_scp_params sp;
    go_model_params mp;
    go_ipm_opts io;
    go_dist_model dm;
    int trace_cap, n_trace;      /* go_set_trace: (traj_prev, subproblem optimum) of every trip, for lock-step tests */
    double *trXp, *trUp, *trXn, *trUn;
    int n_box, n_sph, n_obs;
    double *box, *sph;
    double x_init[NX], goal_lo[NX], goal_hi[NX], tf, dt;
    /* current trajectory (SCPS.traj) */
    double *X, *U;
    /* histories (types.jl:150-173, scp_gusto.jl:15-19) */
    int cap, iterations, converged, successful, stop_reason, total_ipm;
    int nJ_true, nJ_full, n_rho, n_hist; /* n_hist = entries of per-iteration vectors */
    double *J_true, *J_full, *conv, *Delta, *omega, *rho;
    int *accept, *scp_status, *solver_status, *tr_sat, *cvx_sat, *ipm_it;
    double toggle, dual[NX];
    /* linearisation (model.f, model.A, model.B of the reference) */
    double *fk, *Ak, *Bk;
    /* rows */
    go_row* rows;
    int nrows, rows_cap;
    int* row_start; /* N+1 */
    /* IPM per-row state */
    double *rt, *rlam, *rlamb, *rs, *rg, *rrp, *rsig, *rrho0, *rD, *rdl, *rds, *rdt, *rka, *rkb, *rw;
    /* IPM per-stage data */
    double *Fm, *Gm, *Mm, *bm, *hk, *Phi, *Gam;      /* [N][n*n] etc. */
    double *Hx, *Hu, *gx, *gu, *rd, *QQ, *qq, *cc;
    double *Ps, *ps, *Pis, *Ks, *Sinv, *Ds, *d0s;
    double *dX, *dU, *nun, *nu, *Xw, *Uw;
    double mug[NX], mugn[NX];
    double Gd[NX * NX];
    int warm; /* the previous subproblem of this problem's SCP run ended GO_SOLVER_OPTIMAL */
    /* TrajOpt variant (scp_trajopt.jl): m = m0 + n, the last n "controls" of a knot are the defect d_k of the interval
     * (k, k+1): x_{k+1} - x_k - dt/2 (a_k + a_{k+1}) = d_k is a hard row, mu |d_k|_1 the L1 penalty of the dynamics */
    int trajopt, m0;
    go_trajopt_params tp;
    double to_mu, to_s;
    int n_solves, n_mu, n_xtol, n_ftol, n_ctol, to_cap;
    double *s_vec, *mu_vec, *xtol_vec, *ftol_vec, *ctol_vec;
};

/* ------------------------------------------------------------------------------------------ */
/* tiny dense helpers, row-major                                                                */
static void mm(double* C, const double* A, const double* B, int p, int q, int r) { /* C=A*B */
    for (int i = 0; i < p; i++)
        for (int j = 0; j < r; j++) {
            double s = 0;
            for (int l = 0; l < q; l++) s += A[i * q + l] * B[l * r + j];
            C[i * r + j] = s;
        }
}
static void mtm(double* C, const double* A, const double* B, int p, int q, int r) { /* C=A^T*B, A is q x p */
    for (int i = 0; i < p; i++)
        for (int j = 0; j < r; j++) {
            double s = 0;
            for (int l = 0; l < q; l++) s += A[l * p + i] * B[l * r + j];
            C[i * r + j] = s;
        }
}
static void mv(double* y, const double* A, const double* x, int p, int q) {
    for (int i = 0; i < p; i++) {
        double s = 0;
        for (int l = 0; l < q; l++) s += A[i * q + l] * x[l];
        y[i] = s;
    }
}
static void mtv(double* y, const double* A, const double* x, int p, int q) { /* y = A^T x, A is q x p */
    for (int i = 0; i < p; i++) {
        double s = 0;
        for (int l = 0; l < q; l++) s += A[l * p + i] * x[l];
        y[i] = s;
    }
}
/* inverse by Gauss-Jordan with partial pivoting; returns 0 ok */
static int inv_gj(double* Ainv, const double* A, int n) {
    double W[NX * 2 * NX];
    int w = 2 * n;
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            W[i * w + j] = A[i * n + j];
            W[i * w + n + j] = (i == j) ? 1.0 : 0.0;
        }
    for (int c = 0; c < n; c++) {
        int piv = c;
        double best = fabs(W[c * w + c]);
        for (int r = c + 1; r < n; r++)
            if (fabs(W[r * w + c]) > best) { best = fabs(W[r * w + c]); piv = r; }
        if (best == 0.0) return -1;
        if (piv != c)
            for (int j = 0; j < w; j++) { double t = W[c * w + j]; W[c * w + j] = W[piv * w + j]; W[piv * w + j] = t; }
        double d = 1.0 / W[c * w + c];
        for (int j = 0; j < w; j++) W[c * w + j] *= d;
        for (int r = 0; r < n; r++) {
            if (r == c) continue;
            double f = W[r * w + c];
            if (f != 0.0)
                for (int j = 0; j < w; j++) W[r * w + j] -= f * W[c * w + j];
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) Ainv[i * n + j] = W[i * w + n + j];
    return 0;
}
/* Cholesky S = L L^T; returns Li = L^{-1} (lower) and Sinv = Li^T Li; 0 ok, -1 not positive definite */
static int chol_inv(double* Sinv, double* Li, const double* S, int n) {
    double L[NU * NU];   /* (n <= NU: the control block of the TrajOpt variant is (m + n) x (m + n)) */
    memset(L, 0, sizeof(L));
    for (int j = 0; j < n; j++) {
        double d = S[j * n + j];
        for (int l = 0; l < j; l++) d -= L[j * n + l] * L[j * n + l];
        if (!(d > 0.0)) return -1;
        d = sqrt(d);
        L[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = S[i * n + j];
            for (int l = 0; l < j; l++) s -= L[i * n + l] * L[j * n + l];
            L[i * n + j] = s / d;
        }
    }
    for (int i = 0; i < n * n; i++) Li[i] = 0;
    for (int j = 0; j < n; j++) {
        Li[j * n + j] = 1.0 / L[j * n + j];
        for (int i = j + 1; i < n; i++) {
            double s = 0;
            for (int l = j; l < i; l++) s -= L[i * n + l] * Li[l * n + j];
            Li[i * n + j] = s / L[i * n + i];
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            int l0 = i > j ? i : j;
            for (int l = l0; l < n; l++) s += Li[l * n + i] * Li[l * n + j];
            Sinv[i * n + j] = s;
        }
    return 0;
}
static int inv_spd(double* Sinv, const double* S, int n) {
    double Li[NU * NU];
    return chol_inv(Sinv, Li, S, n);
}

/* ------------------------------------------------------------------------------------------ */
/* parameters: SCPParam / SCPParam_GuSTO per model and robot constants                          */
void go_default_params(int model, go_scp_params* sp, go_model_params* mp) {
    memset(sp, 0, sizeof(*sp));
    memset(mp, 0, sizeof(*mp));
    sp->omega_max = 1.0e10;
    sp->beta_succ = 2.0;
    sp->beta_fail = 0.5;
    sp->omega0 = 1.0;
    mp->n_robot_comp = 1;
    switch (model) {
    case GO_FREEFLYER_SE2: /* freeflyer_se2.jl:15-39, robot/freeflyer.jl:28-62 */
        sp->Delta0 = 3.0; sp->eps = 1.0e-2; sp->rho0 = 0.1; sp->rho1 = 0.3; sp->gamma_fail = 10.0;
        sp->convergence_threshold = 1.0e-2;
        mp->mass = 0.5 * (15.36 + 18.08);
        mp->Jdiag[0] = mp->Jdiag[1] = mp->Jdiag[2] = 0.184;
        mp->radius = 0.157; mp->clearance = 0.05;
        mp->hard_limit_vel = 0.2;
        mp->hard_limit_accel = 2 * 0.185 / mp->mass;
        mp->hard_limit_omega = 20 * M_PI / 180;
        mp->hard_limit_alpha = (1.0 / (0.184 / 6.43)) * 0.593;
        mp->n_robot_comp = 2; /* body + arm cylinder, freeflyer.jl:53-57 */
        mp->comp_off[1][0] = 0.0; mp->comp_off[1][1] = 0.15; mp->comp_off[1][2] = 0.0;
        break;
    case GO_DUBINS_CAR: /* dubins_car.jl:22-52 */
        sp->Delta0 = 10000.0; sp->eps = 1.0e-6; sp->rho0 = 0.4; sp->rho1 = 1.5; sp->gamma_fail = 5.0;
        sp->convergence_threshold = 1e-4;
        mp->dubins_v = 2.0; mp->dubins_k = 1.0;
        mp->x_max[0] = 100.0; mp->x_max[1] = 100.0; mp->x_max[2] = 2 * M_PI;
        for (int i = 0; i < 3; i++) mp->x_min[i] = -mp->x_max[i];
        mp->u_max = 10.0; mp->u_min = -10.0;
        mp->clearance = 0.01;
        break;
    case GO_ASTROBEE_SE3: /* astrobee_se3.jl:16-40, robot/astrobee3D.jl:15-33 */
    case GO_ASTROBEE_SE3_MANIFOLD: /* astrobee_se3_manifold.jl:18-46 */
        if (model == GO_ASTROBEE_SE3) {
            sp->Delta0 = 10.0; sp->eps = 1.0e-6; sp->rho0 = 0.01; sp->rho1 = 0.05; sp->gamma_fail = 5.0;
            sp->convergence_threshold = 1e-2;
        } else {
            sp->Delta0 = 1000.0; sp->eps = 1.0e-1; sp->rho0 = 0.01; sp->rho1 = 100.0; sp->gamma_fail = 5.0;
            sp->convergence_threshold = 1e-4;
        }
        mp->mass = 7.0;
        mp->Jdiag[0] = mp->Jdiag[1] = mp->Jdiag[2] = 0.1083;
        mp->radius = sqrt(3.0) * 0.5 * 0.305;
        mp->clearance = 0.03;
        mp->hard_limit_vel = 0.5; mp->hard_limit_accel = 0.1;
        mp->hard_limit_omega = 45 * M_PI / 180; mp->hard_limit_alpha = 50 * M_PI / 180;
        break;
    }
}
void go_default_ipm_opts(go_ipm_opts* o) {
    o->tol = 1e-8; o->tol_acc = 1e-5; o->mu_floor = -1.0 /* the model's */; o->tr_tol = 1e-6; o->max_iter = 60; o->acc_iter = 0;
    o->mu_warm = -1.0; o->mu_warm_gain = -1.0; o->mu_warm_max = -1.0; /* the model's warm-start triple */
    o->sigma_max = -1.0; /* the algorithm's: 0.1 for GuSTO, none for TrajOpt */
}
int go_model_dims(int model, int* n, int* m) {
    switch (model) {
    case GO_FREEFLYER_SE2: *n = 6; *m = 3; return 0;
    case GO_DUBINS_CAR: *n = 3; *m = 1; return 0;
    case GO_ASTROBEE_SE3: *n = 12; *m = 6; return 0;
    case GO_ASTROBEE_SE3_MANIFOLD: *n = 13; *m = 6; return 0;
    }
    return -1;
}

/* ------------------------------------------------------------------------------------------ */
/* dynamics: f, A = df/dx, B = df/du                                                            */
static void cross3(double* c, const double* a, const double* b) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}
void go_dynamics(const go_problem* p, const double* x, const double* u, double* f, double* A, double* B) {
    const int n = p->n, m = p->m;   /* (TrajOpt: B has m = m0 + n columns, the defect columns stay zero) */
    const go_model_params* mp = &p->mp;
    if (A) memset(A, 0, sizeof(double) * n * n);
    if (B) memset(B, 0, sizeof(double) * n * m);
    switch (p->model) {
    case GO_FREEFLYER_SE2: { /* freeflyer_se2.jl:182-206 */
        if (f) {
            f[0] = x[3]; f[1] = x[4]; f[2] = x[5];
            f[3] = u[0] / mp->mass; f[4] = u[1] / mp->mass; f[5] = u[2] * (1.0 / mp->Jdiag[2]);
        }
        if (A) for (int i = 0; i < 3; i++) A[i * n + 3 + i] = 1.0;
        if (B) { B[3 * m + 0] = 1.0 / mp->mass; B[4 * m + 1] = 1.0 / mp->mass; B[5 * m + 2] = 1.0 / mp->Jdiag[2]; }
        break;
    }
    case GO_DUBINS_CAR: { /* dubins_car.jl:161-181 */
        if (f) { f[0] = mp->dubins_v * cos(x[2]); f[1] = mp->dubins_v * sin(x[2]); f[2] = mp->dubins_k * u[0]; }
        if (A) { A[0 * n + 2] = -mp->dubins_v * sin(x[2]); A[1 * n + 2] = mp->dubins_v * cos(x[2]); }
        if (B) B[2 * m + 0] = mp->dubins_k;
        break;
    }
    case GO_ASTROBEE_SE3: { /* astrobee_se3.jl:180-241, quat_functions.jl:253-257 */
        const double* pp = x + 6; const double* w = x + 9;
        const double Jx = mp->Jdiag[0], Jy = mp->Jdiag[1], Jz = mp->Jdiag[2];
        if (f) {
            for (int i = 0; i < 3; i++) { f[i] = x[3 + i]; f[3 + i] = u[i] / mp->mass; }
            double p2 = pp[0] * pp[0] + pp[1] * pp[1] + pp[2] * pp[2];
            double wp = w[0] * pp[0] + w[1] * pp[1] + w[2] * pp[2];
            double cr[3]; cross3(cr, w, pp);
            for (int i = 0; i < 3; i++) f[6 + i] = 0.25 * ((1 - p2) * w[i] - 2 * cr[i] + 2 * wp * pp[i]);
            double Jw[3] = {Jx * w[0], Jy * w[1], Jz * w[2]}, c2[3];
            cross3(c2, w, Jw);
            f[9] = (u[3] - c2[0]) / Jx; f[10] = (u[4] - c2[1]) / Jy; f[11] = (u[5] - c2[2]) / Jz;
        }
        if (A) {
            double px = pp[0], py = pp[1], pz = pp[2], wx = w[0], wy = w[1], wz = w[2];
            for (int i = 0; i < 3; i++) A[i * n + 3 + i] = 1.0;
#define AA(i, j) A[((i)-1) * n + ((j)-1)]
            AA(7, 7) = (px * wx) / 2 + (py * wy) / 2 + (pz * wz) / 2;
            AA(7, 8) = wz / 2 + (px * wy) / 2 - (py * wx) / 2;
            AA(7, 9) = (px * wz) / 2 - wy / 2 - (pz * wx) / 2;
            AA(7, 10) = px * px / 4 - py * py / 4 - pz * pz / 4 + 0.25;
            AA(7, 11) = (px * py) / 2 - pz / 2;
            AA(7, 12) = py / 2 + (px * pz) / 2;
            AA(8, 7) = (py * wx) / 2 - (px * wy) / 2 - wz / 2;
            AA(8, 8) = (px * wx) / 2 + (py * wy) / 2 + (pz * wz) / 2;
            AA(8, 9) = wx / 2 + (py * wz) / 2 - (pz * wy) / 2;
            AA(8, 10) = pz / 2 + (px * py) / 2;
            AA(8, 11) = -px * px / 4 + py * py / 4 - pz * pz / 4 + 0.25;
            AA(8, 12) = (py * pz) / 2 - px / 2;
            AA(9, 7) = wy / 2 - (px * wz) / 2 + (pz * wx) / 2;
            AA(9, 8) = (pz * wy) / 2 - (py * wz) / 2 - wx / 2;
            AA(9, 9) = (px * wx) / 2 + (py * wy) / 2 + (pz * wz) / 2;
            AA(9, 10) = (px * pz) / 2 - py / 2;
            AA(9, 11) = px / 2 + (py * pz) / 2;
            AA(9, 12) = -px * px / 4 - py * py / 4 + pz * pz / 4 + 0.25;
            AA(10, 11) = (Jy - Jz) * wz / Jx;
            AA(10, 12) = (Jy - Jz) * wy / Jx;
            AA(11, 10) = -(Jx - Jz) * wz / Jy;
            AA(11, 12) = -(Jx - Jz) * wx / Jy;
            AA(12, 10) = (Jx - Jy) * wy / Jz;
            AA(12, 11) = (Jx - Jy) * wx / Jz;
        }
        if (B) {
            for (int i = 0; i < 3; i++) B[(3 + i) * m + i] = 1.0 / mp->mass;
            B[9 * m + 3] = 1.0 / Jx; B[10 * m + 4] = 1.0 / Jy; B[11 * m + 5] = 1.0 / Jz;
        }
        break;
    }
    case GO_ASTROBEE_SE3_MANIFOLD: { /* astrobee_se3_manifold.jl:231-304 */
        const double qw = x[6], qx = x[7], qy = x[8], qz = x[9];
        const double wx = x[10], wy = x[11], wz = x[12];
        const double Jx = mp->Jdiag[0], Jy = mp->Jdiag[1], Jz = mp->Jdiag[2];
        if (f) {
            for (int i = 0; i < 3; i++) { f[i] = x[3 + i]; f[3 + i] = u[i] / mp->mass; }
            f[6] = 0.5 * (-wx * qx - wy * qy - wz * qz);
            f[7] = 0.5 * (wx * qw - wz * qy + wy * qz);
            f[8] = 0.5 * (wy * qw + wz * qx - wx * qz);
            f[9] = 0.5 * (wz * qw - wy * qx + wx * qy);
            double w[3] = {wx, wy, wz}, Jw[3] = {Jx * wx, Jy * wy, Jz * wz}, c2[3];
            cross3(c2, w, Jw);
            f[10] = (u[3] - c2[0]) / Jx; f[11] = (u[4] - c2[1]) / Jy; f[12] = (u[5] - c2[2]) / Jz;
        }
        if (A) {
            for (int i = 0; i < 3; i++) A[i * n + 3 + i] = 1.0;
            AA(7, 8) = -wx / 2; AA(7, 9) = -wy / 2; AA(7, 10) = -wz / 2;
            AA(7, 11) = -qx / 2; AA(7, 12) = -qy / 2; AA(7, 13) = -qz / 2;
            AA(8, 7) = wx / 2; AA(8, 9) = -wz / 2; AA(8, 10) = wy / 2;
            AA(8, 11) = qw / 2; AA(8, 12) = qz / 2; AA(8, 13) = -qy / 2;
            AA(9, 7) = wy / 2; AA(9, 8) = wz / 2; AA(9, 10) = -wx / 2;
            AA(9, 11) = -qz / 2; AA(9, 12) = qw / 2; AA(9, 13) = qx / 2;
            AA(10, 7) = wz / 2; AA(10, 8) = -wy / 2; AA(10, 9) = wx / 2;
            AA(10, 11) = qy / 2; AA(10, 12) = -qx / 2; AA(10, 13) = qw / 2;
            AA(11, 12) = (Jy - Jz) * wz / Jx;
            AA(11, 13) = (Jy - Jz) * wy / Jx;
            AA(12, 11) = -(Jx - Jz) * wz / Jy;
            AA(12, 13) = -(Jx - Jz) * wx / Jy;
            AA(13, 11) = (Jx - Jy) * wy / Jz;
            AA(13, 12) = (Jx - Jy) * wx / Jz;
#undef AA
        }
        if (B) {
            for (int i = 0; i < 3; i++) B[(3 + i) * m + i] = 1.0 / mp->mass;
            B[10 * m + 3] = 1.0 / Jx; B[11 * m + 4] = 1.0 / Jy; B[12 * m + 5] = 1.0 / Jz;
        }
        break;
    }
    }
}

/* ------------------------------------------------------------------------------------------ */
/* signed distance (restates BulletCollision.distance for the in-scope primitives)              */
static int ws_dim(const go_problem* p) { return p->model == GO_FREEFLYER_SE2 ? 2 : 3; }

/* signed distance from point q to an axis-aligned box in `d` dims + outward unit normal */
static double sdf_box(const double* q, const double* lo, const double* hi, int d, double* nh) {
    double v[3], s2 = 0;
    int outside = 0;
    for (int i = 0; i < d; i++) {
        double e = 0;
        if (q[i] < lo[i]) e = q[i] - lo[i];
        else if (q[i] > hi[i]) e = q[i] - hi[i];
        v[i] = e;
        if (e != 0) outside = 1;
        s2 += e * e;
    }
    if (outside) {
        double dist = sqrt(s2);
        for (int i = 0; i < d; i++) nh[i] = v[i] / dist;
        return dist;
    }
    /* inside: nearest face, fixed tie-break order -x,+x,-y,+y,-z,+z */
    double best = q[0] - lo[0];
    int bi = 0, bs = -1;
    for (int i = 0; i < d; i++) {
        double a = q[i] - lo[i], b = hi[i] - q[i];
        if (a < best) { best = a; bi = i; bs = -1; }
        if (b < best) { best = b; bi = i; bs = +1; }
    }
    for (int i = 0; i < d; i++) nh[i] = 0;
    nh[bi] = bs;
    return -best;
}
/* ---- study model of what a 3-D collision library returns for the freeflyer --------------
 * The robot body is an upright prism over a regular n-gon (n_poly vertices on the circle of radius r; 0 = the
 * exact disc) spanning z in [z_lo, z_hi]; obstacles are 3-D AABBs.  Separated: Euclidean distance (z ranges of
 * every freeflyer obstacle overlap the robot's, so it is the planar distance).  Penetrating: the minimum
 * translation that separates the two bodies, which may be VERTICAL (lifting the robot over a low box) -- then
 * the normal has no planar component.  `margin` is subtracted from every distance (collision margins).     */
static double poly_rect_2d(const double* c, double rad, int np, double phase, const double* lo, const double* hi,
                           double* nh) {
    double V[64][2], W[4][2] = {{lo[0], lo[1]}, {hi[0], lo[1]}, {hi[0], hi[1]}, {lo[0], hi[1]}};
    if (np > 64) np = 64;
    for (int a = 0; a < np; a++) {
        double t = phase + 2.0 * M_PI * a / np;
        V[a][0] = c[0] + rad * cos(t); V[a][1] = c[1] + rad * sin(t);
    }
    /* separating axis test over the edge normals of both polygons: overlap on every axis <=> intersecting */
    double best_ov = 1e300, bax[2] = {0, 0};
    int sep = 0;
    for (int pass = 0; pass < 2; pass++) {
        int ne = pass ? 4 : np;
        for (int e = 0; e < ne; e++) {
            const double *a = pass ? W[e] : V[e], *b = pass ? W[(e + 1) % 4] : V[(e + 1) % np];
            double ax[2] = {b[1] - a[1], -(b[0] - a[0])}, l = hypot(ax[0], ax[1]);
            ax[0] /= l; ax[1] /= l;
            double vmin = 1e300, vmax = -1e300, wmin = 1e300, wmax = -1e300;
            for (int j = 0; j < np; j++) { double t = V[j][0] * ax[0] + V[j][1] * ax[1]; if (t < vmin) vmin = t; if (t > vmax) vmax = t; }
            for (int j = 0; j < 4; j++) { double t = W[j][0] * ax[0] + W[j][1] * ax[1]; if (t < wmin) wmin = t; if (t > wmax) wmax = t; }
            /* translation of the robot along +ax by (wmax - vmin) or along -ax by (vmax - wmin) separates */
            double o1 = wmax - vmin, o2 = vmax - wmin;
            if (o1 <= 0 || o2 <= 0) { sep = 1; break; }
            if (o1 < best_ov) { best_ov = o1; bax[0] = ax[0]; bax[1] = ax[1]; }
            if (o2 < best_ov) { best_ov = o2; bax[0] = -ax[0]; bax[1] = -ax[1]; }
        }
        if (sep) break;
    }
    if (!sep) { nh[0] = bax[0]; nh[1] = bax[1]; return -best_ov; }
    /* disjoint convex polygons: closest vertex-edge pair, both ways */
    double best = 1e300;
    for (int pass = 0; pass < 2; pass++) {
        int nv = pass ? 4 : np, ne = pass ? np : 4;
        for (int j = 0; j < nv; j++) {
            const double* q = pass ? W[j] : V[j];
            for (int e = 0; e < ne; e++) {
                const double *a = pass ? V[e] : W[e], *b = pass ? V[(e + 1) % np] : W[(e + 1) % 4];
                double ab[2] = {b[0] - a[0], b[1] - a[1]}, aq[2] = {q[0] - a[0], q[1] - a[1]};
                double t = (aq[0] * ab[0] + aq[1] * ab[1]) / (ab[0] * ab[0] + ab[1] * ab[1]);
                t = t < 0 ? 0 : (t > 1 ? 1 : t);
                double cx = a[0] + t * ab[0], cy = a[1] + t * ab[1], dx = q[0] - cx, dy = q[1] - cy, dd = hypot(dx, dy);
                if (dd < best) {
                    best = dd; /* normal points from the obstacle to the robot */
                    if (pass) { nh[0] = -dx / dd; nh[1] = -dy / dd; } else { nh[0] = dx / dd; nh[1] = dy / dd; }
                }
            }
        }
    }
    return best;
}
static double dist_study_freeflyer(const go_problem* p, int comp, const double* q, const double* lo, const double* hi,
                                   double* nh) {
    const go_dist_model* dm = &p->dm;
    double dist;
    nh[0] = nh[1] = nh[2] = 0;
    if (dm->n_poly > 0) dist = poly_rect_2d(q, p->mp.radius, dm->n_poly, dm->poly_phase, lo, hi, nh);
    else dist = sdf_box(q, lo, hi, 2, nh) - p->mp.radius;
    if (dist < 0 && dm->vertical_escape) {
        double up = hi[2] - dm->z_lo, down = dm->z_hi - lo[2], v = up < down ? up : down;
        if (v < -dist) { dist = -v; nh[0] = nh[1] = 0; nh[2] = up < down ? 1.0 : -1.0; }
    }
    if (dist < 0 && (dm->pen_mode == 1 || (dm->pen_mode == 2 && comp > 0))) return dm->pen_value; /* "no result" from a detector without penetration solver */
    if (dm->pen_mode == 3 && fabs(dist) < dm->pen_band) return dm->pen_value; /* degenerate near-contact band */
    if (dm->pen_mode == 4 && dist < 0 && dist > -dm->pen_band) return dm->pen_value; /* shallow penetration  */
    return dist - dm->margin;
}
void go_set_distance_model(go_problem* p, const go_dist_model* dm) { p->dm = *dm; }

/* obstacle i: boxes first (keepout_zones then obstacle_set boxes), then spheres (types.jl:19) */
double go_signed_distance(const go_problem* p, int comp, const double* r, int i, double* nhat) {
    const int d = ws_dim(p);
    double q[3] = {0, 0, 0}, nh[3] = {0, 0, 0}, dist;
    for (int j = 0; j < d; j++) q[j] = r[j] + p->mp.comp_off[comp][j];
    if (p->dm.kind == 1 && p->model == GO_FREEFLYER_SE2 && i < p->n_box) {
        const double* bx = p->box + 6 * i;
        dist = dist_study_freeflyer(p, comp, q, bx, bx + 3, nh);
    } else if (i < p->n_box) {
        const double* bx = p->box + 6 * i;
        dist = sdf_box(q, bx, bx + 3, d, nh) - p->mp.radius;
    } else {
        const double* sp = p->sph + 4 * (i - p->n_box);
        double v[3], s2 = 0;
        for (int j = 0; j < d; j++) { v[j] = q[j] - sp[j]; s2 += v[j] * v[j]; }
        double nrm = sqrt(s2);
        for (int j = 0; j < d; j++) nh[j] = v[j] / nrm;
        dist = nrm - sp[3] - p->mp.radius;
    }
    if (nhat) for (int j = 0; j < d; j++) nhat[j] = nh[j];
    return dist;
}

/* ------------------------------------------------------------------------------------------ */
static double row_val(const go_row* r, const double* v) {
    double s = r->c0;
    for (int j = 0; j < r->nnz; j++) {
        double w = v[r->idx[j]], e = w - r->v0[j];
        s += r->a[j] * e * e + r->b[j] * w;
    }
    return s;
}
static go_row* new_row(go_problem* p, int k, int isu, int kind) {
    if (p->nrows >= p->rows_cap) { fprintf(stderr, "gusto_oracle: row capacity exceeded\n"); abort(); }
    go_row* r = &p->rows[p->nrows++];
    memset(r, 0, sizeof(*r));
    r->k = k; r->isu = isu; r->kind = kind; r->mul = 1.0; r->off = 0.0;
    return r;
}
static void row_add(go_row* r, int idx, double a, double v0, double b) {
    r->idx[r->nnz] = idx; r->a[r->nnz] = a; r->v0[r->nnz] = v0; r->b[r->nnz] = b; r->nnz++;
}

/* Rows of the convex subproblem around (Xp,Up): the reference's SCPConstraints(SCPP) registry
 * (freeflyer_se2.jl:338-390, dubins_car.jl:184-226, astrobee_se3.jl:322-379,
 * astrobee_se3_manifold.jl:533-608) flattened into stage-local rows.  `kappa` = 1/max(1,omega)
 * rescales the whole objective so that slack multipliers stay in [0,1]. */
static void assemble_rows(go_problem* p, const double* Xp, double Delta, double omega, double toggle, double kappa) {
    const int n = p->n, N = p->N, d = ws_dim(p);
    const go_model_params* mp = &p->mp;
    const double eps = p->sp.eps;
    p->nrows = 0;
    for (int k = 0; k < N; k++) {
        p->row_start[k] = p->nrows;
        const double* xp = Xp + k * n;
        go_row* r;
        switch (p->model) {
        case GO_FREEFLYER_SE2:
        case GO_ASTROBEE_SE3:
        case GO_ASTROBEE_SE3_MANIFOLD: {
            const int is2 = p->model == GO_FREEFLYER_SE2, man = p->model == GO_ASTROBEE_SE3_MANIFOLD;
            const int nv = is2 ? 2 : 3, iw = is2 ? 5 : (man ? 10 : 9), nw = is2 ? 1 : 3;
            if (!man) { /* stri_state_trust_region: w*||x-xp||^2 - Delta <= s */
                r = new_row(p, k, 0, ROW_PEN_TR);
                for (int j = 0; j < n; j++) row_add(r, j, 1.0, xp[j], 0.0);
                r->mul = kappa * omega; r->off = kappa * Delta;
            }
            if (man) {
                /* cse_quaternion_norm (manifold.jl:308-313): h = |qp| + qp.(q-qp)/|qp| - 1, penalised as a
                 * +-eps pair (scp_gusto.jl:297-311): j=1 is the hard bound w*h+eps >= 0, j=2 the L1 penalty. */
                double qn = sqrt(xp[6] * xp[6] + xp[7] * xp[7] + xp[8] * xp[8] + xp[9] * xp[9]);
                double c0 = qn - 1.0;
                for (int j = 0; j < 4; j++) c0 -= xp[6 + j] * xp[6 + j] / qn;
                r = new_row(p, k, 0, ROW_HARD_EQ); /* -(w*h) - eps <= 0 */
                for (int j = 0; j < 4; j++) row_add(r, 6 + j, 0.0, 0.0, -xp[6 + j] / qn);
                r->c0 = -c0; r->mul = kappa * omega; r->off = kappa * eps;
                r = new_row(p, k, 0, ROW_PEN_EQ); /* w*h - eps <= s */
                for (int j = 0; j < 4; j++) row_add(r, 6 + j, 0.0, 0.0, xp[6 + j] / qn);
                r->c0 = c0; r->mul = kappa * omega; r->off = kappa * eps;
                /* csi_orientation_sign: -qw <= s/w */
                r = new_row(p, k, 0, ROW_PEN);
                row_add(r, 6, 0.0, 0.0, -1.0);
                r->mul = kappa * omega;
            }
            /* csi_translational_velocity_bound */
            r = new_row(p, k, 0, ROW_PEN);
            for (int j = 0; j < nv; j++) row_add(r, 3 + j, 1.0, 0.0, 0.0);
            r->c0 = -mp->hard_limit_vel * mp->hard_limit_vel; r->mul = kappa * omega;
            /* csi_angular_velocity_bound */
            r = new_row(p, k, 0, ROW_PEN);
            for (int j = 0; j < nw; j++) row_add(r, iw + j, 1.0, 0.0, 0.0);
            r->c0 = -mp->hard_limit_omega * mp->hard_limit_omega; r->mul = kappa * omega;
            /* ncsi_*_convexified: clearance - (dist + nhat.(r - r0)) if dist < toggle else 0 */
            for (int i = 0; i < p->n_obs; i++) {
                double nh[3];
                double dist = go_signed_distance(p, 0, xp, i, nh);
                if (dist < toggle) {
                    r = new_row(p, k, 0, ROW_PEN);
                    double c0 = mp->clearance - dist;
                    for (int j = 0; j < d; j++) { row_add(r, j, 0.0, 0.0, -nh[j]); c0 += nh[j] * xp[j]; }
                    r->c0 = c0; r->mul = kappa * omega;
                }
            }
            if (k < N - 1) { /* cci_*_accel_bound, k = 1..N-1 only (freeflyer_se2.jl:380-381) */
                const int nf = is2 ? 2 : 3, im = is2 ? 2 : 3, nm = is2 ? 1 : 3;
                r = new_row(p, k, 1, ROW_HARD);
                for (int j = 0; j < nf; j++) row_add(r, j, 1.0 / (mp->mass * mp->mass), 0.0, 0.0);
                r->c0 = -mp->hard_limit_accel * mp->hard_limit_accel;
                r->mul = 1.0 / (mp->hard_limit_accel * mp->hard_limit_accel);
                r = new_row(p, k, 1, ROW_HARD);
                for (int j = 0; j < nm; j++) {
                    double ji = 1.0 / mp->Jdiag[is2 ? 2 : j];
                    row_add(r, im + j, ji * ji, 0.0, 0.0);
                }
                r->c0 = -mp->hard_limit_alpha * mp->hard_limit_alpha;
                r->mul = 1.0 / (mp->hard_limit_alpha * mp->hard_limit_alpha);
            }
            break;
        }
        case GO_DUBINS_CAR: {
            for (int i = 0; i < n; i++) { /* csi_max_bound_constraints, dynamics.jl:56-59 */
                r = new_row(p, k, 0, ROW_PEN);
                row_add(r, i, 0.0, 0.0, 1.0); r->c0 = -mp->x_max[i]; r->mul = kappa * omega;
            }
            for (int i = 0; i < n; i++) { /* csi_min_bound_constraints, dynamics.jl:61-64 */
                r = new_row(p, k, 0, ROW_PEN);
                row_add(r, i, 0.0, 0.0, -1.0); r->c0 = mp->x_min[i]; r->mul = kappa * omega;
            }
            if (k < N - 1) { /* cci_max/min_bound_constraints, dynamics.jl:73-81 */
                r = new_row(p, k, 1, ROW_HARD);
                row_add(r, 0, 0.0, 0.0, 1.0); r->c0 = -mp->u_max; r->mul = 1.0 / fabs(mp->u_max);
                r = new_row(p, k, 1, ROW_HARD);
                row_add(r, 0, 0.0, 0.0, -1.0); r->c0 = mp->u_min; r->mul = 1.0 / fabs(mp->u_min);
            }
            break;
        }
        }
        if (k == N - 1) { /* csbci_goal_constraints (BoxGoal), dynamics.jl:37-42: hard */
            for (int i = 0; i < n; i++) {
                double lo = p->goal_lo[i], hi = p->goal_hi[i];
                if (lo == hi) continue;
                double sc = 1.0 / fmax(1e-3, fmin(1.0, (isfinite(hi) && isfinite(lo)) ? 0.5 * (hi - lo) : 1.0));
                if (isfinite(hi)) { r = new_row(p, k, 0, ROW_HARD); row_add(r, i, 0.0, 0.0, 1.0); r->c0 = -hi; r->mul = sc; }
                if (isfinite(lo)) { r = new_row(p, k, 0, ROW_HARD); row_add(r, i, 0.0, 0.0, -1.0); r->c0 = lo; r->mul = sc; }
            }
        }
    }
    p->row_start[N] = p->nrows;
}

/* Rows of the TrajOpt subproblem around Xp (scp_trajopt.jl:159-279), same registry, different treatment:
 *   hard       : state trust region  ||x_k - xp_k||^2 - s <= 0                      (:165-173)
 *   penalised  : mu * g <= v, v >= 0, cost += v  for convex_state_ineq, nonconvex_state_convexified_ineq AND
 *                convex_control_ineq (hard in GuSTO)                                (:222-235)
 *   dynamics   : mu * |d_kj| as the pair  +-mu d_kj <= v  on the defect variables    (:257-275; as written there the pair of
 *                auxiliary variables is free and the subproblem unbounded below: the L1 form of the equality branch :238-246
 *                is what it means -- DESIGN.md section 4)
 * Boundary rows (x_1 = x_init, goals) stay hard as in GuSTO.  `kappa` = 1/max(1, mu) rescales the objective. */
static void assemble_rows_trajopt(go_problem* p, const double* Xp, double s_tr, double mu, double toggle, double kappa) {
    const int n = p->n, N = p->N, d = ws_dim(p), m0 = p->m0;
    const go_model_params* mp = &p->mp;
    const int is2 = p->model == GO_FREEFLYER_SE2, man = p->model == GO_ASTROBEE_SE3_MANIFOLD;
    const int nv = is2 ? 2 : 3, iw = is2 ? 5 : (man ? 10 : 9), nw = is2 ? 1 : 3;
    p->nrows = 0;
    for (int k = 0; k < N; k++) {
        p->row_start[k] = p->nrows;
        const double* xp = Xp + k * n;
        go_row* r;
        if (!man) {   /* (the manifold model registers no state_trust_region_ineq row: astrobee_se3_manifold.jl:601) */
            r = new_row(p, k, 0, ROW_HARD);   /* stri_state_trust_region - s <= 0, normalised by s */
            for (int j = 0; j < n; j++) row_add(r, j, 1.0, xp[j], 0.0);
            r->c0 = -s_tr; r->mul = 1.0 / s_tr;
        } else {
            /* cse_quaternion_norm (manifold.jl:308-313), a convex_state_eq row: hard `== 0` here (:200-208), as the band
             * ROW_EQ row (GO_EQ_DELTA above);  csi_orientation_sign (:316-319) penalised like every convex_state_ineq row */
            double qn = sqrt(xp[6] * xp[6] + xp[7] * xp[7] + xp[8] * xp[8] + xp[9] * xp[9]);
            double c0 = qn - 1.0;
            for (int j = 0; j < 4; j++) c0 -= xp[6 + j] * xp[6 + j] / qn;
            r = new_row(p, k, 0, ROW_EQ);
            for (int j = 0; j < 4; j++) row_add(r, 6 + j, 0.0, 0.0, xp[6 + j] / qn);
            r->c0 = c0; r->mul = 1.0;
            r = new_row(p, k, 0, ROW_PEN);
            row_add(r, 6, 0.0, 0.0, -1.0);
            r->mul = kappa * mu;
        }
        r = new_row(p, k, 0, ROW_PEN);    /* csi_translational_velocity_bound */
        for (int j = 0; j < nv; j++) row_add(r, 3 + j, 1.0, 0.0, 0.0);
        r->c0 = -mp->hard_limit_vel * mp->hard_limit_vel; r->mul = kappa * mu;
        r = new_row(p, k, 0, ROW_PEN);    /* csi_angular_velocity_bound */
        for (int j = 0; j < nw; j++) row_add(r, iw + j, 1.0, 0.0, 0.0);
        r->c0 = -mp->hard_limit_omega * mp->hard_limit_omega; r->mul = kappa * mu;
        for (int i = 0; i < p->n_obs; i++) {   /* ncsi_*_convexified, active inside obstacle_toggle_distance = clearance + 1 (:65) */
            double nh[3];
            double dist = go_signed_distance(p, 0, xp, i, nh);
            if (dist < toggle) {
                r = new_row(p, k, 0, ROW_PEN);
                double c0 = mp->clearance - dist;
                for (int j = 0; j < d; j++) { row_add(r, j, 0.0, 0.0, -nh[j]); c0 += nh[j] * xp[j]; }
                r->c0 = c0; r->mul = kappa * mu;
            }
        }
        if (k < N - 1) {   /* cci_*_accel_bound, penalised here */
            const int nf = is2 ? 2 : 3, im = is2 ? 2 : 3, nm = is2 ? 1 : 3;
            r = new_row(p, k, 1, ROW_PEN);
            for (int j = 0; j < nf; j++) row_add(r, j, 1.0 / (mp->mass * mp->mass), 0.0, 0.0);
            r->c0 = -mp->hard_limit_accel * mp->hard_limit_accel; r->mul = kappa * mu;
            r = new_row(p, k, 1, ROW_PEN);
            for (int j = 0; j < nm; j++) { double ji = 1.0 / mp->Jdiag[is2 ? 2 : j]; row_add(r, im + j, ji * ji, 0.0, 0.0); }
            r->c0 = -mp->hard_limit_alpha * mp->hard_limit_alpha; r->mul = kappa * mu;
        }
        for (int j = 0; j < n; j++) {   /* mu |d_kj|: d_k of the last knot moves nothing and is driven to zero */
            r = new_row(p, k, 1, ROW_PEN); row_add(r, m0 + j, 0.0, 0.0, 1.0); r->mul = kappa * mu;
            r = new_row(p, k, 1, ROW_PEN); row_add(r, m0 + j, 0.0, 0.0, -1.0); r->mul = kappa * mu;
        }
        if (k == N - 1) {   /* csbci_goal_constraints (BoxGoal): hard, as in GuSTO */
            for (int i = 0; i < n; i++) {
                double lo = p->goal_lo[i], hi = p->goal_hi[i];
                if (lo == hi) continue;
                double sc = 1.0 / fmax(1e-3, fmin(1.0, (isfinite(hi) && isfinite(lo)) ? 0.5 * (hi - lo) : 1.0));
                if (isfinite(hi)) { r = new_row(p, k, 0, ROW_HARD); row_add(r, i, 0.0, 0.0, 1.0); r->c0 = -hi; r->mul = sc; }
                if (isfinite(lo)) { r = new_row(p, k, 0, ROW_HARD); row_add(r, i, 0.0, 0.0, -1.0); r->c0 = lo; r->mul = sc; }
            }
        }
    }
    p->row_start[N] = p->nrows;
}

/* ------------------------------------------------------------------------------------------ */
/* linearisation: initialize_model_params!/update_model_params! (freeflyer_se2.jl:116-147 etc.)  */
static void linearize(go_problem* p, const double* Xp, const double* Up) {
    const int n = p->n, m = p->m, N = p->N;
    const double hdt = 0.5 * p->dt;
    double e[NX * 64];
    double* ek = (N <= 64) ? e : (double*)malloc(sizeof(double) * n * N);
    for (int k = 0; k < N; k++) {
        double *f = p->fk + k * n, *A = p->Ak + k * n * n, *B = p->Bk + k * n * m;
        go_dynamics(p, Xp + k * n, Up + k * m, f, A, B);
        double *F = p->Fm + k * n * n, *G = p->Gm + k * n * n, *b = p->bm + k * n * m;
        for (int i = 0; i < n; i++) {
            double s = f[i];
            for (int j = 0; j < n; j++) {
                s -= A[i * n + j] * Xp[k * n + j];
                F[i * n + j] = (i == j ? 1.0 : 0.0) + hdt * A[i * n + j];
                G[i * n + j] = (i == j ? 1.0 : 0.0) - hdt * A[i * n + j];
            }
            for (int j = 0; j < m; j++) { s -= B[i * m + j] * Up[k * m + j]; b[i * m + j] = hdt * B[i * m + j]; }
            ek[k * n + i] = s;
        }
        if (inv_gj(p->Mm + k * n * n, G, n)) { fprintf(stderr, "gusto_oracle: singular I - dt/2 A\n"); abort(); }
    }
    for (int k = 0; k < N; k++) {
        double* Phi = p->Phi + k * n * n, *Gam = p->Gam + k * n * m;
        for (int i = 0; i < n; i++) p->hk[k * n + i] = (k == 0) ? 0.0 : hdt * (ek[(k - 1) * n + i] + ek[k * n + i]);
        if (k == 0) { /* x_1 is fixed: dy_0 = F_0 dx_0 + b_0 du_0 */
            memset(Phi, 0, sizeof(double) * n * n);
            memcpy(Gam, p->bm, sizeof(double) * n * m);
        } else {
            mm(Phi, p->Fm + k * n * n, p->Mm + k * n * n, n, n, n);
            mm(Gam, Phi, p->bm + k * n * m, n, n, m);
            for (int i = 0; i < n * m; i++) Gam[i] += p->bm[k * n * m + i];
        }
        /* TrajOpt: y_k := F_k x_k + b_k u_k + d_k, so the defect moves dy_k directly (Gam_d = I) and not x_k (b_d = 0) */
        if (p->trajopt)
            for (int i = 0; i < n; i++)
                for (int j = 0; j < n; j++) Gam[i * m + p->m0 + j] = (i == j) ? 1.0 : 0.0;
    }
    if (ek != e) free(ek);
}

/* ------------------------------------------------------------------------------------------ */
/* Newton system of the IPM.
 * Unknowns per stage k: dx_k, du_k; trapezoid rows (freeflyer_se2.jl:160-172) in Newton form
 *   F_{k-1} dx_{k-1} + b_{k-1} du_{k-1} - G_k dx_k + b_k du_k = -rd_k,   k = 1..N-1
 * With y_k := F_k x_k + b_k u_k they read  dx_k = M_k (dy_{k-1} + b_k du_k + rd_k), M_k = G_k^{-1}, i.e.
 * a standard LQR in the state dy (dim n) and control du_k:
 *   dy_k = Phi_k dy_{k-1} + Gam_k du_k + c_k,  Phi = F M, Gam = Phi b + b, c = Phi rd.
 * Goal point rows C x_N = g are adjoined with multiplier mu_g: the backward sweep also carries
 * Pi = d p / d mu_g, so mu_g is known after the backward sweep alone.                             */
static int riccati_factor(go_problem* p, int ng, const int* gidx) {
    const int n = p->n, m = p->m, N = p->N, nz = n + m;
    double P[NX * NX], Pi[NX * NX], T[NX * NZ], PG[NX * NZ], Hh[NZ * NZ], Z[NZ * NX];
    double Qt[NX * NX], Qb[NX * NU], tmp[NX * NX], S[NU * NU];
    memset(P, 0, sizeof(P));
    memset(Pi, 0, sizeof(Pi));
    memset(p->Gd, 0, sizeof(double) * NX * NX);
    for (int k = N - 1; k >= 0; k--) {
        const double *Phi = p->Phi + k * n * n, *Gam = p->Gam + k * n * m, *M = p->Mm + k * n * n, *b = p->bm + k * n * m;
        double* QQ = p->QQ + k * nz * nz;
        memcpy(p->Ps + k * n * n, P, sizeof(double) * n * n);
        memcpy(p->Pis + k * n * ng, Pi, sizeof(double) * n * ng);
        /* stage cost in (dy', du): QQ = [Qt, Qt b; ., Hu + b^T Qt b], Qt = M^T Hx M */
        memset(QQ, 0, sizeof(double) * nz * nz);
        if (k > 0) {
            mm(tmp, p->Hx + k * n * n, M, n, n, n);
            mtm(Qt, M, tmp, n, n, n);
            mm(Qb, Qt, b, n, n, m);
            for (int i = 0; i < n; i++) {
                for (int j = 0; j < n; j++) QQ[i * nz + j] = Qt[i * n + j];
                for (int j = 0; j < m; j++) { QQ[i * nz + n + j] = Qb[i * m + j]; QQ[(n + j) * nz + i] = Qb[i * m + j]; }
            }
            mtm(S, b, Qb, m, n, m);
        } else {
            memset(S, 0, sizeof(S));
        }
        for (int i = 0; i < m; i++)
            for (int j = 0; j < m; j++) QQ[(n + i) * nz + n + j] = p->Hu[k * m * m + i * m + j] + S[i * m + j];
        /* PG = [Phi Gam] (n x nz), T = P PG, Hh = QQ + PG^T T, Z = PG^T Pi */
        for (int i = 0; i < n; i++) {
            for (int j = 0; j < n; j++) PG[i * nz + j] = Phi[i * n + j];
            for (int j = 0; j < m; j++) PG[i * nz + n + j] = Gam[i * m + j];
        }
        mm(T, P, PG, n, n, nz);
        mtm(Hh, PG, T, nz, n, nz);
        for (int i = 0; i < nz * nz; i++) Hh[i] += QQ[i];
        mtm(Z, PG, Pi, nz, n, ng);
        if (k == N - 1 && ng > 0) { /* E = [M^T C^T; b^T M^T C^T] */
            for (int j = 0; j < ng; j++) {
                double ey[NX];
                for (int i = 0; i < n; i++) { ey[i] = M[gidx[j] * n + i]; Z[i * ng + j] += ey[i]; }
                for (int i = 0; i < m; i++) {
                    double s = 0;
                    for (int l = 0; l < n; l++) s += b[l * m + i] * ey[l];
                    Z[(n + i) * ng + j] += s;
                }
            }
        }
        for (int i = 0; i < m; i++)
            for (int j = 0; j < m; j++) S[i * m + j] = 0.5 * (Hh[(n + i) * nz + n + j] + Hh[(n + j) * nz + n + i]);
        double* Sinv = p->Sinv + k * m * m;
        double Li[NU * NU], Wm[NU * NX], Vm[NU * NX];
        if (chol_inv(Sinv, Li, S, m)) {
            if (getenv("GO_DEBUG")) fprintf(stderr, "gusto_oracle: S not PD at k=%d\n", k);
            return -1;
        }
        /* block Cholesky of [S Hyu^T; Hyu Hyy]: W = L^-1 Hyu^T, V = L^-1 Zu; K = L^-T W, D = L^-T V.
         * The Schur complements are formed as Hyy - W^T W (never through an explicit S^-1): with barrier
         * weights ~1/mu in Hyy the explicit form loses all digits. */
        for (int i = 0; i < m; i++) {
            for (int j = 0; j < n; j++) {
                double s = 0;
                for (int l = 0; l <= i; l++) s += Li[i * m + l] * Hh[j * nz + n + l];
                Wm[i * n + j] = s;
            }
            for (int j = 0; j < ng; j++) {
                double s = 0;
                for (int l = 0; l <= i; l++) s += Li[i * m + l] * Z[(n + l) * ng + j];
                Vm[i * ng + j] = s;
            }
        }
        double* K = p->Ks + k * m * n, *D = p->Ds + k * m * ng;
        for (int i = 0; i < m; i++) {
            for (int j = 0; j < n; j++) {
                double s = 0;
                for (int l = i; l < m; l++) s += Li[l * m + i] * Wm[l * n + j];
                K[i * n + j] = s;
            }
            for (int j = 0; j < ng; j++) {
                double s = 0;
                for (int l = i; l < m; l++) s += Li[l * m + i] * Vm[l * ng + j];
                D[i * ng + j] = s;
            }
        }
        /* Gd += V^T V ; P' = Hyy - W^T W ; Pi' = Zy - W^T V */
        for (int i = 0; i < ng; i++)
            for (int j = 0; j < ng; j++) {
                double s = 0;
                for (int l = 0; l < m; l++) s += Vm[l * ng + i] * Vm[l * ng + j];
                p->Gd[i * ng + j] += s;
            }
        for (int i = 0; i < n; i++) {
            for (int j = 0; j < n; j++) {
                double s = 0.5 * (Hh[i * nz + j] + Hh[j * nz + i]);
                for (int l = 0; l < m; l++) s -= Wm[l * n + i] * Wm[l * n + j];
                P[i * n + j] = s;
            }
            for (int j = 0; j < ng; j++) {
                double s = Z[i * ng + j];
                for (int l = 0; l < m; l++) s -= Wm[l * n + i] * Vm[l * ng + j];
                Pi[i * ng + j] = s;
            }
        }
    }
    if (ng > 0) {
        double Gi[NX * NX];
        if (inv_spd(Gi, p->Gd, ng)) { if (getenv("GO_DEBUG")) { fprintf(stderr,"Gd not PD:"); for(int i=0;i<ng*ng;i++) fprintf(stderr," %g",p->Gd[i]); fprintf(stderr,"\n"); } return -2; }
        memcpy(p->Gd, Gi, sizeof(double) * ng * ng); /* Gd now holds its inverse */
    }
    return 0;
}

/* one right-hand side: gradients gx,gu (stage cost linear terms), residuals rd (dynamics), r0 = 0, rg (goal) */
static void riccati_solve(go_problem* p, int ng, const int* gidx, const double* rg) {
    const int n = p->n, m = p->m, N = p->N, nz = n + m;
    double pv[NX], tp[NX], l[NZ], gy[NX], t1[NX], th[NX], lu[NU];
    memset(pv, 0, sizeof(pv));
    memset(th, 0, sizeof(th));
    for (int k = N - 1; k >= 0; k--) {
        const double *Phi = p->Phi + k * n * n, *Gam = p->Gam + k * n * m, *M = p->Mm + k * n * n, *b = p->bm + k * n * m;
        const double* QQ = p->QQ + k * nz * nz;
        const double* rd = p->rd + k * n;
        double* c = p->cc + k * n;
        memcpy(p->ps + k * n, pv, sizeof(double) * n);
        if (k > 0) {
            mv(c, Phi, rd, n, n);
            /* gy = Qt rd + M^T gx ; q = [gy; gu + b^T gy] */
            mtv(t1, M, p->gx + k * n, n, n);
            for (int i = 0; i < n; i++) {
                double s = t1[i];
                for (int j = 0; j < n; j++) s += QQ[i * nz + j] * rd[j];
                gy[i] = s;
            }
        } else {
            memset(c, 0, sizeof(double) * n); /* c_0 = F_0 r0, r0 = 0 */
            memset(gy, 0, sizeof(gy));
        }
        for (int i = 0; i < n; i++) l[i] = gy[i];
        for (int i = 0; i < m; i++) {
            double s = p->gu[k * m + i];
            for (int j = 0; j < n; j++) s += b[j * m + i] * gy[j];
            l[n + i] = s;
        }
        /* tp = p + P c ; l += [Phi Gam]^T tp */
        mv(t1, p->Ps + k * n * n, c, n, n);
        for (int i = 0; i < n; i++) tp[i] = pv[i] + t1[i];
        for (int i = 0; i < n; i++) {
            double s = 0;
            for (int j = 0; j < n; j++) s += Phi[j * n + i] * tp[j];
            l[i] += s;
        }
        for (int i = 0; i < m; i++) {
            double s = 0;
            for (int j = 0; j < n; j++) s += Gam[j * m + i] * tp[j];
            l[n + i] += s;
            lu[i] = l[n + i];
        }
        /* theta += Pi_k^T c - D_k^T lu ; d0 = Sinv lu ; p' = ly - K^T lu */
        const double *Pi = p->Pis + k * n * ng, *D = p->Ds + k * m * ng, *K = p->Ks + k * m * n;
        for (int j = 0; j < ng; j++) {
            double s = 0;
            for (int i = 0; i < n; i++) s += Pi[i * ng + j] * c[i];
            for (int i = 0; i < m; i++) s -= D[i * ng + j] * lu[i];
            th[j] += s;
        }
        mv(p->d0s + k * m, p->Sinv + k * m * m, lu, m, m);
        for (int i = 0; i < n; i++) {
            double s = l[i];
            for (int j = 0; j < m; j++) s -= K[j * n + i] * lu[j];
            pv[i] = s;
        }
    }
    /* mu_g = Gd^{-1} (theta + C M rd_{N-1} - rg) */
    if (ng > 0) {
        double rhs[NX];
        const double* M = p->Mm + (N - 1) * n * n;
        for (int j = 0; j < ng; j++) {
            double s = th[j] - rg[j];
            for (int i = 0; i < n; i++) s += M[gidx[j] * n + i] * p->rd[(N - 1) * n + i];
            rhs[j] = s;
        }
        mv(p->mugn, p->Gd, rhs, ng, ng);
    }
    /* forward sweep */
    double dy[NX], dyn[NX], du[NU], a[NX];
    memset(dy, 0, sizeof(dy));
    for (int k = 0; k < N; k++) {
        const double *Phi = p->Phi + k * n * n, *Gam = p->Gam + k * n * m, *M = p->Mm + k * n * n, *b = p->bm + k * n * m;
        const double *D = p->Ds + k * m * ng, *K = p->Ks + k * m * n;
        for (int i = 0; i < m; i++) {
            double s = p->d0s[k * m + i];
            for (int j = 0; j < ng; j++) s += D[i * ng + j] * p->mugn[j];
            for (int j = 0; j < n; j++) s += K[i * n + j] * dy[j];
            du[i] = -s;
            p->dU[k * m + i] = du[i];
        }
        if (k == 0) {
            memset(p->dX, 0, sizeof(double) * n);
        } else {
            for (int i = 0; i < n; i++) {
                double s = dy[i] + p->rd[k * n + i];
                for (int j = 0; j < m; j++) s += b[i * m + j] * du[j];
                a[i] = s;
            }
            mv(p->dX + k * n, M, a, n, n);
        }
        for (int i = 0; i < n; i++) {
            double s = p->cc[k * n + i];
            for (int j = 0; j < n; j++) s += Phi[i * n + j] * dy[j];
            for (int j = 0; j < m; j++) s += Gam[i * m + j] * du[j];
            dyn[i] = s;
        }
        if (k + 1 < N) { /* costate of the trapezoid row k+1: nu = P dy + p + Pi mu_g */
            const double *P = p->Ps + k * n * n, *Pi = p->Pis + k * n * ng;
            for (int i = 0; i < n; i++) {
                double s = p->ps[k * n + i];
                for (int j = 0; j < n; j++) s += P[i * n + j] * dyn[j];
                for (int j = 0; j < ng; j++) s += Pi[i * ng + j] * p->mugn[j];
                p->nun[(k + 1) * n + i] = s;
            }
        }
        memcpy(dy, dyn, sizeof(double) * n);
    }
    /* multiplier of x_1 = x_init from the stationarity of x_1: gx_0 + nu_0 + F_0^T nu_1 = 0 (dx_0 = 0) */
    for (int i = 0; i < n; i++) {
        double s = p->gx[i];
        if (N > 1)
            for (int j = 0; j < n; j++) s += p->Fm[j * n + i] * p->nun[n + j];
        p->nun[i] = -s;
    }
}

/* ------------------------------------------------------------------------------------------ */
/* The convex subproblem (scp_gusto.jl:178-314) solved by a primal-dual interior point method.
 *   min  kappa * sum_k w_k |u_k|^2 + sum_pen s_i
 *   s.t. x_1 = x_init, C x_N = goal, trapezoid rows (hard);  mul*g_i - off <= 0 (hard rows);
 *        mul*g_i - off <= s_i, s_i >= 0 (penalised rows)                                          */
#define GO_IPM_DIVERGED 1e3
static double max_step(double a, double v, double dv, double tau) {
    if (dv < 0) { double c = -tau * v / dv; if (c < a) a = c; }
    return a;
}

static int ipm_solve(go_problem* p, const double* Xp, const double* Up, double Delta, double omega, double toggle,
                     go_sub_info* info) {
    const int n = p->n, m = p->m, N = p->N;
    const double kappa = 1.0 / fmax(1.0, omega);
    go_ipm_opts io_resolved = p->io;
    /* complementarity floor of the model (gusto_hip.h: mu_floor < 0; common.hpp: warm_defaults): the manifold model's solves sit
     * at the noise floor of their dual residual for ten iterations when mu is driven to 1e-11 */
    if (io_resolved.mu_floor < 0) io_resolved.mu_floor = (p->model == GO_ASTROBEE_SE3_MANIFOLD && !p->trajopt) ? 1e-10 : 1e-11;
    if (io_resolved.sigma_max < 0) io_resolved.sigma_max = p->trajopt ? 0.0 : 0.1;
    const go_ipm_opts* io = &io_resolved;
    int gidx[NX], ng = 0;
    double gval[NX];
    for (int i = 0; i < n; i++)
        if (p->goal_lo[i] == p->goal_hi[i]) { gidx[ng] = i; gval[ng] = p->goal_lo[i]; ng++; }

    linearize(p, Xp, Up);
    if (p->trajopt) assemble_rows_trajopt(p, Xp, Delta /* s */, omega /* mu */, toggle, kappa);
    else assemble_rows(p, Xp, Delta, omega, toggle, kappa);
    const int nr = p->nrows;
    double *X = p->Xw, *U = p->Uw;
    memcpy(X, Xp, sizeof(double) * n * N);
    memcpy(U, Up, sizeof(double) * m * N);
    memcpy(X, p->x_init, sizeof(double) * n); /* warm start at traj_prev (scp_gusto.jl:100-102) with x_1 pinned */
    memset(p->nu, 0, sizeof(double) * n * N);
    memset(p->mug, 0, sizeof(p->mug));
    memset(p->mugn, 0, sizeof(p->mugn));

    /* Start point.  Cold (first subproblem after go_set_problem, or after a solver failure): slacks just inside, multipliers small (tuned on the
     * freeflyer batch).  Warm (the iterate starts at the optimum of the previous subproblem): every penalised row is
     * put ON the central path at mu_warm for its value g at the start point -- s - t = g, t*lam_a = s*lam_b = mu,
     * lam_a + lam_b = 1  <=>  s,t = mu + (sqrt(g^2 + 4 mu^2) +- g)/2 -- and hard rows get lam = mu/t. */
    double muw = 0.0;
    if (p->warm) { /* start level from the size of the last trajectory change (gusto_hip.h: gusto_ipm_opts, common.hpp: warm_mu) */
        double lo = io->mu_warm, gain = io->mu_warm_gain, hi = io->mu_warm_max;
        if (lo < 0) { /* the model's triple, as common.hpp: warm_defaults */
            if (p->model == GO_DUBINS_CAR) { lo = 1e-9; gain = 0.0; hi = 1e-9; }
            else if (p->model == GO_FREEFLYER_SE2) { lo = 1e-4; gain = 0.1; hi = 1e-2; }
            else if (p->model == GO_ASTROBEE_SE3) { lo = 1e-6; gain = 1.0; hi = 1e-2; }
            else { lo = 1e-4; gain = 1.0; hi = 1e-2; }
        } else if (gain < 0) gain = 0.0;
        const double c = p->n_hist >= 1 ? p->conv[p->n_hist - 1] : 0.0;
        muw = (lo == 0.0) ? 0.0 : fmin(fmax(lo, hi), fmax(lo, gain * c * c));
    }
    p->warm = 0;
    int ncomp = 0;
    for (int i = 0; i < nr; i++) {
        go_row* r = &p->rows[i];
        const double* v = (r->isu ? U + r->k * m : X + r->k * n);
        double g = r->mul * row_val(r, v) - r->off;
        if (r->kind == ROW_EQ) {
            p->rt[i] = 1.0; p->rlam[i] = 0.0; p->rlamb[i] = 0; p->rs[i] = 0;   /* (eta = 0; t, s, lamb unused) */
        } else if (r->kind == ROW_HARD || r->kind == ROW_HARD_EQ) {
            const double mu0 = (muw > 0) ? muw : 0.01;
            p->rt[i] = fmax(-g, 1e-2); p->rlam[i] = mu0 / p->rt[i]; p->rlamb[i] = 0; p->rs[i] = 0;
            ncomp += 1;
        } else if (muw > 0) {
            const double ag = fabs(g), rr = sqrt(g * g + 4 * muw * muw);
            const double big = muw + 0.5 * (rr + ag), small = muw + 2 * muw * muw / (rr + ag);
            p->rs[i] = (g >= 0) ? big : small; p->rt[i] = (g >= 0) ? small : big;
            p->rlam[i] = muw / p->rt[i]; p->rlamb[i] = muw / p->rs[i];
            ncomp += 2;
        } else {
            p->rs[i] = fmax(g, 0.0) + 0.01; p->rt[i] = p->rs[i] - g; p->rlam[i] = 0.5; p->rlamb[i] = 0.5;
            ncomp += 2;
        }
    }
    double wk[256];
    for (int k = 0; k < N; k++) wk[k] = kappa * ((k == 0 || k == N - 1) ? 0.5 * p->dt : p->dt);

    int status = GO_SOLVER_FAILED, it, n_acc = 0;
    double res_p = 0, res_d = 0, mu = 0, mu_start = 0, rg[NX];
    for (it = 0;; it++) {
        /* residuals ----------------------------------------------------------------------- */
        res_p = 0;
        memset(p->rd, 0, sizeof(double) * n);
        for (int k = 1; k < N; k++) {
            const double *F = p->Fm + (k - 1) * n * n, *G = p->Gm + k * n * n, *b0 = p->bm + (k - 1) * n * m, *b1 = p->bm + k * n * m;
            for (int i = 0; i < n; i++) {
                double s = p->hk[k * n + i];
                for (int j = 0; j < n; j++) s += F[i * n + j] * X[(k - 1) * n + j] - G[i * n + j] * X[k * n + j];
                for (int j = 0; j < m; j++) s += b0[i * m + j] * U[(k - 1) * m + j] + b1[i * m + j] * U[k * m + j];
                if (p->trajopt) s += U[(k - 1) * m + p->m0 + i];   /* + d_{k-1}: the defect of the interval (k-1, k) */
                p->rd[k * n + i] = s;
                if (fabs(s) > res_p) res_p = fabs(s);
            }
        }
        for (int j = 0; j < ng; j++) {
            rg[j] = gval[j] - X[(N - 1) * n + gidx[j]];
            if (fabs(rg[j]) > res_p) res_p = fabs(rg[j]);
        }
        double comp = 0;
        for (int i = 0; i < nr; i++) {
            go_row* r = &p->rows[i];
            const double* v = (r->isu ? U + r->k * m : X + r->k * n);
            double g = r->mul * row_val(r, v) - r->off;
            p->rg[i] = g;
            if (r->kind == ROW_EQ) {
                p->rrp[i] = g;
            } else if (r->kind == ROW_HARD || r->kind == ROW_HARD_EQ) {
                p->rrp[i] = g + p->rt[i];
                comp += p->rt[i] * p->rlam[i];
            } else {
                p->rrp[i] = g - p->rs[i] + p->rt[i];
                comp += p->rt[i] * p->rlam[i] + p->rs[i] * p->rlamb[i];
            }
            if (fabs(p->rrp[i]) > res_p) res_p = fabs(p->rrp[i]);
        }
        mu = ncomp ? comp / ncomp : 0.0;
        if (getenv("GO_DEBUG_IPM")) {
            double mrd = 0, mrp = 0; int arg = -1;
            for (int i = n; i < n * N; i++) if (fabs(p->rd[i]) > mrd) mrd = fabs(p->rd[i]);
            for (int i = 0; i < nr; i++) if (fabs(p->rrp[i]) > mrp) { mrp = fabs(p->rrp[i]); arg = i; }
            fprintf(stderr, "it %d: rd %.3e rp %.3e (row %d k %d kind %d isu %d g %.3e t %.3e s %.3e lam %.3e lamb %.3e) mu %.3e\n", it, mrd, mrp, arg,
                    arg >= 0 ? p->rows[arg].k : -1, arg >= 0 ? p->rows[arg].kind : -1, arg >= 0 ? p->rows[arg].isu : -1, arg >= 0 ? p->rg[arg] : 0,
                    arg >= 0 ? p->rt[arg] : 0, arg >= 0 ? p->rs[arg] : 0, arg >= 0 ? p->rlam[arg] : 0, arg >= 0 ? p->rlamb[arg] : 0, mu);
        }
        /* dual residual: grad f + J^T lam + E^T nu ------------------------------------------ */
        res_d = 0;
        double numax = 0;
        for (int k = 0; k < N; k++) {
            double dx[NX], du[NU];
            memset(dx, 0, sizeof(dx));
            for (int i = 0; i < m; i++) du[i] = 2 * wk[k] * ((i < p->m0) ? 1.0 : GO_TRAJOPT_DEFECT_REG) * U[k * m + i];
            if (p->trajopt && k + 1 < N)
                for (int i = 0; i < n; i++) du[p->m0 + i] += p->nu[(k + 1) * n + i];
            for (int i = p->row_start[k]; i < p->row_start[k + 1]; i++) {
                go_row* r = &p->rows[i];
                const double* v = (r->isu ? U + k * m : X + k * n);
                double* dst = r->isu ? du : dx;
                for (int j = 0; j < r->nnz; j++)
                    dst[r->idx[j]] += p->rlam[i] * r->mul * (2 * r->a[j] * (v[r->idx[j]] - r->v0[j]) + r->b[j]);
            }
            if (k + 1 < N) {
                const double *F = p->Fm + k * n * n, *b = p->bm + k * n * m, *nu1 = p->nu + (k + 1) * n;
                for (int i = 0; i < n; i++) { double s = 0; for (int j = 0; j < n; j++) s += F[j * n + i] * nu1[j]; dx[i] += s; }
                for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < n; j++) s += b[j * m + i] * nu1[j]; du[i] += s; }
            }
            if (k >= 1) {
                const double *G = p->Gm + k * n * n, *b = p->bm + k * n * m, *nu0 = p->nu + k * n;
                for (int i = 0; i < n; i++) { double s = 0; for (int j = 0; j < n; j++) s += G[j * n + i] * nu0[j]; dx[i] -= s; }
                for (int i = 0; i < m; i++) { double s = 0; for (int j = 0; j < n; j++) s += b[j * m + i] * nu0[j]; du[i] += s; }
            }
            if (k == N - 1)
                for (int j = 0; j < ng; j++) dx[gidx[j]] += p->mug[j];
            if (getenv("GO_DEBUG_IPM")) {
                static double mx, mu_, md; static int kx, ku, kd;
                if (k == 0) { mx = mu_ = md = 0; kx = ku = kd = -1; }
                if (k > 0) for (int i = 0; i < n; i++) if (fabs(dx[i]) > mx) { mx = fabs(dx[i]); kx = k; }
                for (int i = 0; i < p->m0; i++) if (fabs(du[i]) > mu_) { mu_ = fabs(du[i]); ku = k; }
                for (int i = p->m0; i < m; i++) if (fabs(du[i]) > md) { md = fabs(du[i]); kd = k; }
                if (k == N - 1) fprintf(stderr, "   res_d: x %.3e (k %d)  u %.3e (k %d)  d %.3e (k %d)\n", mx, kx, mu_, ku, md, kd);
            }
            if (k > 0)
                for (int i = 0; i < n; i++) if (fabs(dx[i]) > res_d) res_d = fabs(dx[i]);
            for (int i = 0; i < m; i++) if (fabs(du[i]) > res_d) res_d = fabs(du[i]);
            for (int i = 0; i < n; i++) if (fabs(p->nu[k * n + i]) > numax) numax = fabs(p->nu[k * n + i]);
        }
        if (getenv("GO_DEBUG_IPM")) fprintf(stderr, "   test: res_p %.3e res_d %.3e numax %.3e mu %.3e tol %.3e\n", res_p, res_d, numax, mu, io->tol);
        if (res_p <= io->tol && res_d <= io->tol * (1 + numax) && mu <= 0.1 * io->tol) { status = GO_SOLVER_OPTIMAL; break; }
        const int acceptable = res_p <= io->tol_acc && res_d <= io->tol_acc * (1 + numax) && mu <= io->tol_acc;
        n_acc = acceptable ? n_acc + 1 : 0;
        if (it >= io->max_iter || (io->acc_iter > 0 && n_acc >= io->acc_iter)) {
            if (acceptable) status = GO_SOLVER_ALMOST;
            break;
        }
        if (!isfinite(res_p) || !isfinite(res_d) || !isfinite(mu)) break;
        if (it == 0) mu_start = mu;
        if (mu > GO_IPM_DIVERGED * fmax(1.0, mu_start)) break; /* diverging: an infeasible subproblem (common.hpp: IPM_DIVERGED) */

        /* Hessian blocks (shared by predictor and corrector) ----------------------------------- */
        memset(p->Hx, 0, sizeof(double) * n * n * N);
        memset(p->Hu, 0, sizeof(double) * m * m * N);
        for (int k = 0; k < N; k++)
            for (int i = 0; i < m; i++) p->Hu[k * m * m + i * m + i] = 2 * wk[k] * ((i < p->m0) ? 1.0 : GO_TRAJOPT_DEFECT_REG);
        for (int i = 0; i < nr; i++) {
            go_row* r = &p->rows[i];
            const int k = r->k, dim = r->isu ? m : n;
            const double* v = (r->isu ? U + k * m : X + k * n);
            double* H = r->isu ? p->Hu + k * m * m : p->Hx + k * n * n;
            double gr[NX];
            for (int j = 0; j < r->nnz; j++) gr[j] = r->mul * (2 * r->a[j] * (v[r->idx[j]] - r->v0[j]) + r->b[j]);
            if (r->kind == ROW_EQ) {
                p->rsig[i] = 1.0 / GO_EQ_DELTA;
            } else if (r->kind == ROW_HARD || r->kind == ROW_HARD_EQ) {
                p->rsig[i] = p->rlam[i] / p->rt[i];
            } else {
                p->rD[i] = p->rt[i] + p->rlam[i] * p->rs[i] / p->rlamb[i];
                p->rsig[i] = p->rlam[i] / p->rD[i];
            }
            for (int a = 0; a < r->nnz; a++) {
                for (int b = 0; b < r->nnz; b++) H[r->idx[a] * dim + r->idx[b]] += p->rsig[i] * gr[a] * gr[b];
                H[r->idx[a] * dim + r->idx[a]] += p->rlam[i] * r->mul * 2 * r->a[a];
            }
        }
        if (riccati_factor(p, ng, gidx)) break;

        double sigma = 0, mu_t = 0, alpha = 1.0;
        for (int pass = 0; pass < 2; pass++) {
            /* pass 0: affine-scaling predictor (mu_t = 0); pass 1: centred corrector */
            memset(p->gx, 0, sizeof(double) * n * N);
            for (int k = 0; k < N; k++)
                for (int i = 0; i < m; i++) p->gu[k * m + i] = 2 * wk[k] * ((i < p->m0) ? 1.0 : GO_TRAJOPT_DEFECT_REG) * U[k * m + i];
            for (int i = 0; i < nr; i++) {
                go_row* r = &p->rows[i];
                const int k = r->k;
                const double* v = (r->isu ? U + k * m : X + k * n);
                double* g = r->isu ? p->gu + k * m : p->gx + k * n;
                double ka = pass ? p->rka[i] : 0.0, kb = pass ? p->rkb[i] : 0.0, coef;
                if (r->kind == ROW_EQ) {
                    coef = p->rlam[i] + p->rrp[i] / GO_EQ_DELTA;
                } else if (r->kind == ROW_HARD || r->kind == ROW_HARD_EQ) {
                    coef = (mu_t - ka + p->rlam[i] * p->rrp[i]) / p->rt[i];
                } else {
                    double lb = p->rlamb[i], la = p->rlam[i];
                    p->rrho0[i] = mu_t - p->rt[i] * la - ka + la * p->rrp[i] - (la / lb) * (mu_t - p->rs[i] * lb - kb);
                    coef = la + p->rrho0[i] / p->rD[i];
                }
                p->rw[i] = coef; /* reused below as scratch */
                for (int j = 0; j < r->nnz; j++)
                    g[r->idx[j]] += coef * r->mul * (2 * r->a[j] * (v[r->idx[j]] - r->v0[j]) + r->b[j]);
            }
            riccati_solve(p, ng, gidx, rg);
            /* row steps */
            double a_max = 1.0;
            const double tau = pass ? fmax(0.995, 1.0 - mu) : 1.0;
            for (int i = 0; i < nr; i++) {
                go_row* r = &p->rows[i];
                const int k = r->k;
                const double* v = (r->isu ? U + k * m : X + k * n);
                const double* dv = (r->isu ? p->dU + k * m : p->dX + k * n);
                double ka = pass ? p->rka[i] : 0.0, kb = pass ? p->rkb[i] : 0.0, w = 0;
                for (int j = 0; j < r->nnz; j++)
                    w += r->mul * (2 * r->a[j] * (v[r->idx[j]] - r->v0[j]) + r->b[j]) * dv[r->idx[j]];
                if (r->kind == ROW_EQ) {   /* d_eta = (h + grad' dx) / delta; no positivity anywhere: no step-length test */
                    p->rdt[i] = 0; p->rds[i] = 0;
                    p->rdl[i] = (p->rrp[i] + w) / GO_EQ_DELTA;
                    continue;
                }
                if (r->kind == ROW_HARD || r->kind == ROW_HARD_EQ) {
                    p->rdt[i] = -p->rrp[i] - w;
                    p->rdl[i] = (mu_t - p->rt[i] * p->rlam[i] - ka - p->rlam[i] * p->rdt[i]) / p->rt[i];
                    p->rds[i] = 0;
                } else {
                    p->rdl[i] = (p->rrho0[i] + p->rlam[i] * w) / p->rD[i];
                    p->rds[i] = (mu_t - p->rs[i] * p->rlamb[i] - kb + p->rs[i] * p->rdl[i]) / p->rlamb[i];
                    p->rdt[i] = -p->rrp[i] - w + p->rds[i];
                    a_max = max_step(a_max, p->rs[i], p->rds[i], tau);
                    a_max = max_step(a_max, p->rlamb[i], -p->rdl[i], tau);
                }
                a_max = max_step(a_max, p->rt[i], p->rdt[i], tau);
                a_max = max_step(a_max, p->rlam[i], p->rdl[i], tau);
            }
            if (pass == 0) {
                double ca = 0;
                for (int i = 0; i < nr; i++) {
                    go_row* r = &p->rows[i];
                    if (r->kind == ROW_EQ) { p->rka[i] = 0; p->rkb[i] = 0; continue; }
                    double ta = p->rt[i] + a_max * p->rdt[i], la = p->rlam[i] + a_max * p->rdl[i];
                    ca += ta * la;
                    if (!(r->kind == ROW_HARD || r->kind == ROW_HARD_EQ))
                        ca += (p->rs[i] + a_max * p->rds[i]) * (p->rlamb[i] - a_max * p->rdl[i]);
                    p->rka[i] = p->rdt[i] * p->rdl[i];
                    p->rkb[i] = -p->rds[i] * p->rdl[i];
                }
                double mu_aff = ncomp ? ca / ncomp : 0.0;
                sigma = (mu > 0) ? pow(mu_aff / mu, 3.0) : 0.0;
                if (io->sigma_max > 0) sigma = fmin(sigma, io->sigma_max); /* (gusto_hip.h: gusto_ipm_opts.sigma_max) */
                mu_t = fmax(sigma * mu, io->mu_floor);
                if (ncomp == 0) break; /* equality-constrained QP: the predictor is the Newton step */
            }
            alpha = a_max;
        }
        if (getenv("GO_DEBUG_IPM")) fprintf(stderr, "      alpha %.3e sigma %.3e\n", alpha, sigma);
        /* update */
        for (int i = 0; i < n * N; i++) X[i] += alpha * p->dX[i];
        for (int i = 0; i < m * N; i++) U[i] += alpha * p->dU[i];
        for (int i = 0; i < n * N; i++) p->nu[i] += alpha * (p->nun[i] - p->nu[i]);
        for (int j = 0; j < ng; j++) p->mug[j] += alpha * (p->mugn[j] - p->mug[j]);
        for (int i = 0; i < nr; i++) {
            go_row* r = &p->rows[i];
            p->rt[i] += alpha * p->rdt[i];
            p->rlam[i] += alpha * p->rdl[i];
            if (!(r->kind == ROW_HARD || r->kind == ROW_HARD_EQ || r->kind == ROW_EQ)) { p->rs[i] += alpha * p->rds[i]; p->rlamb[i] -= alpha * p->rdl[i]; }
        }
    }
    double obj = 0;
    for (int k = 0; k < N; k++)
        for (int i = 0; i < m; i++) obj += wk[k] * ((i < p->m0) ? 1.0 : GO_TRAJOPT_DEFECT_REG) * U[k * m + i] * U[k * m + i];
    for (int i = 0; i < nr; i++)
        if (!(p->rows[i].kind == ROW_HARD || p->rows[i].kind == ROW_HARD_EQ || p->rows[i].kind == ROW_EQ)) obj += p->rs[i];
    info->obj = obj / kappa;
    info->res_p = res_p; info->res_d = res_d; info->mu = mu; info->iters = it; info->status = status;
    p->warm = (status == GO_SOLVER_OPTIMAL);
    for (int i = 0; i < n; i++) p->dual[i] = p->nu[i] / kappa; /* get_dual_jump: -dual(init rows) */
    return status;
}

/* ------------------------------------------------------------------------------------------ */
double go_cost_true(const go_problem* p, const double* U) { /* freeflyer_se2.jl:66-76 */
    double J = 0;
    for (int k = 1; k < p->N; k++)
        for (int j = 0; j < p->m0; j++)   /* (TrajOpt: U rows hold (u, d); only u is costed) */
            J += 0.5 * p->dt * (U[(k - 1) * p->m + j] * U[(k - 1) * p->m + j] + U[k * p->m + j] * U[k * p->m + j]);
    return J;
}
double go_convergence_metric(const go_problem* p, const double* X, const double* Xp) { /* traj_opt.jl:74-85 */
    double mn = -INFINITY, md = -INFINITY;
    for (int k = 0; k < p->N; k++) {
        double a = 0, b = 0;
        for (int i = 0; i < p->n; i++) {
            double e = X[k * p->n + i] - Xp[k * p->n + i];
            a += e * e; b += X[k * p->n + i] * X[k * p->n + i];
        }
        a = sqrt(a); b = sqrt(b);
        if (a > mn) mn = a;
        if (b > md) md = b;
    }
    return mn / md;
}
/* trust_region_ratio_gusto (freeflyer_se2.jl:392-427, dubins_car.jl:229-241, astrobee_se3.jl:383-417,
 * astrobee_se3_manifold.jl:610-642). Linearisation f,A taken at (Xp,Up); B*du deliberately absent.  */
double go_trust_region_ratio(go_problem* p, const double* X, const double* U, const double* Xp, const double* Up) {
    const int n = p->n, m = p->m, N = p->N, d = ws_dim(p);
    double num = 0, den = 0, f[NX], fp[NX], A[NX * NX];
    for (int k = 0; k < N - 1; k++) {
        go_dynamics(p, Xp + k * n, Up + k * m, fp, A, NULL);
        go_dynamics(p, X + k * n, U + k * m, f, NULL, NULL);
        double a = 0, b = 0;
        for (int i = 0; i < n; i++) {
            double lin = fp[i];
            for (int j = 0; j < n; j++) lin += A[i * n + j] * (X[k * n + j] - Xp[k * n + j]);
            a += (f[i] - lin) * (f[i] - lin); b += lin * lin;
        }
        num += sqrt(a); den += sqrt(b);
    }
    if (p->model != GO_DUBINS_CAR) {
        for (int k = 0; k < N; k++) {
            const double *r0 = Xp + k * n, *r = X + k * n;
            for (int c = 0; c < p->mp.n_robot_comp; c++)
                for (int i = 0; i < p->n_obs; i++) {
                    double nh[3];
                    double d0 = go_signed_distance(p, c, r0, i, nh);
                    double lin = p->mp.clearance - d0;
                    for (int j = 0; j < d; j++) lin -= nh[j] * (r[j] - r0[j]);
                    double d1 = go_signed_distance(p, c, r, i, NULL);
                    num += fabs((p->mp.clearance - d1) - lin);
                    den += fabs(lin);
                }
        }
    }
    return num / den;
}
/* convex_ineq_satisfied_gusto_jump (scp_gusto.jl:316-343): raw row values against eps */
int go_convex_ineq_satisfied(go_problem* p, const double* X, const double* Xp, const double* Up, double toggle) {
    (void)Up;
    assemble_rows(p, Xp, 1.0, 1.0, toggle, 1.0);
    for (int i = 0; i < p->nrows; i++) {
        const go_row* r = &p->rows[i];
        if (r->isu) continue;
        double g = row_val(r, X + r->k * p->n);
        if (r->kind == ROW_PEN && g >= p->sp.eps) return 0;
        if (r->kind == ROW_PEN_EQ && (g <= -p->sp.eps || g >= p->sp.eps)) return 0;
    }
    return 1;
}
static int trust_region_satisfied(const go_problem* p, const double* X, const double* Xp, double Delta) {
    /* scp_gusto.jl:34-44; the literal `<= 0` is evaluated with the solver's accuracy as slack */
    double mx = -INFINITY;
    for (int k = 0; k < p->N; k++) {
        double a = 0;
        for (int i = 0; i < p->n; i++) { double e = X[k * p->n + i] - Xp[k * p->n + i]; a += e * e; }
        if (a > mx) mx = a;
    }
    return mx - Delta <= p->io.tr_tol * fmax(1.0, Delta);
}
void go_init_straightline(const go_problem* p, double* X, double* U) { /* freeflyer_se2.jl:97-111 */
    const int n = p->n, N = p->N;
    double xg[NX];
    for (int i = 0; i < n; i++) {
        double lo = p->goal_lo[i], hi = p->goal_hi[i];
        xg[i] = (isfinite(lo) && isfinite(hi)) ? 0.5 * (lo + hi) : 0.0; /* center(goal), zeros elsewhere */
    }
    for (int k = 0; k < N; k++) { /* Julia LinRange element: lerpi(j,d,a,b) = (1-t)*a + t*b, t = j/d */
        const double t = (double)k / (double)(N - 1);
        for (int i = 0; i < n; i++) X[k * n + i] = (1 - t) * p->x_init[i] + t * xg[i];
    }
    memset(U, 0, sizeof(double) * p->m * N);
}

/* ------------------------------------------------------------------------------------------ */
#define ALLOC(ptr, count) ptr = calloc((size_t)(count), sizeof(*(ptr)))
static go_problem* create_impl(int model, int N, const go_scp_params* sp, const go_model_params* mp, int n_box,
                               const double* box, int n_sph, const double* sph, int trajopt) {
    int n, m;
    if (go_model_dims(model, &n, &m) || N < 3 || N > 256) return NULL;
    go_problem* p = calloc(1, sizeof(*p));
    p->m0 = m; p->trajopt = trajopt;
    if (trajopt) m += n;   /* controls (u, d) */
    p->model = model; p->n = n; p->m = m; p->N = N;
    p->sp = *sp; p->mp = *mp;
    go_default_ipm_opts(&p->io);
    p->n_box = n_box; p->n_sph = n_sph; p->n_obs = (model == GO_DUBINS_CAR) ? 0 : n_box + n_sph;
    ALLOC(p->box, 6 * n_box + 1); ALLOC(p->sph, 4 * n_sph + 1);
    if (n_box) memcpy(p->box, box, sizeof(double) * 6 * n_box);
    if (n_sph) memcpy(p->sph, sph, sizeof(double) * 4 * n_sph);
    ALLOC(p->X, n * N); ALLOC(p->U, m * N);
    p->cap = 0;
    ALLOC(p->fk, n * N); ALLOC(p->Ak, n * n * N); ALLOC(p->Bk, n * m * N);
    p->rows_cap = N * (8 + p->n_obs + 4 * n);
    ALLOC(p->rows, p->rows_cap); ALLOC(p->row_start, N + 1);
    const int rc = p->rows_cap;
    ALLOC(p->rt, rc); ALLOC(p->rlam, rc); ALLOC(p->rlamb, rc); ALLOC(p->rs, rc); ALLOC(p->rg, rc); ALLOC(p->rrp, rc);
    ALLOC(p->rsig, rc); ALLOC(p->rrho0, rc); ALLOC(p->rD, rc); ALLOC(p->rdl, rc); ALLOC(p->rds, rc); ALLOC(p->rdt, rc);
    ALLOC(p->rka, rc); ALLOC(p->rkb, rc); ALLOC(p->rw, rc);
    const int nz = n + m;
    ALLOC(p->Fm, n * n * N); ALLOC(p->Gm, n * n * N); ALLOC(p->Mm, n * n * N); ALLOC(p->bm, n * m * N);
    ALLOC(p->hk, n * N); ALLOC(p->Phi, n * n * N); ALLOC(p->Gam, n * m * N);
    ALLOC(p->Hx, n * n * N); ALLOC(p->Hu, m * m * N); ALLOC(p->gx, n * N); ALLOC(p->gu, m * N); ALLOC(p->rd, n * N);
    ALLOC(p->QQ, nz * nz * N); ALLOC(p->qq, nz * N); ALLOC(p->cc, n * N);
    ALLOC(p->Ps, n * n * N); ALLOC(p->ps, n * N); ALLOC(p->Pis, n * n * N); ALLOC(p->Ks, m * n * N);
    ALLOC(p->Sinv, m * m * N); ALLOC(p->Ds, m * n * N); ALLOC(p->d0s, m * N);
    ALLOC(p->dX, n * N); ALLOC(p->dU, m * N); ALLOC(p->nun, n * N); ALLOC(p->nu, n * N); ALLOC(p->Xw, n * N); ALLOC(p->Uw, m * N);
    return p;
}
go_problem* go_create(int model, int N, const go_scp_params* sp, const go_model_params* mp, int n_box,
                      const double* box, int n_sph, const double* sph) {
    return create_impl(model, N, sp, mp, n_box, box, n_sph, sph, 0);
}
static void free_hist(go_problem* p) {
    free(p->J_true); free(p->J_full); free(p->conv); free(p->Delta); free(p->omega); free(p->rho);
    free(p->accept); free(p->scp_status); free(p->solver_status); free(p->tr_sat); free(p->cvx_sat); free(p->ipm_it);
    p->J_true = p->J_full = p->conv = p->Delta = p->omega = p->rho = NULL;
    p->accept = p->scp_status = p->solver_status = p->tr_sat = p->cvx_sat = p->ipm_it = NULL;
    p->cap = 0;
}
void go_destroy(go_problem* p) {
    if (!p) return;
    free_hist(p);
    free(p->box); free(p->sph); free(p->X); free(p->U); free(p->fk); free(p->Ak); free(p->Bk); free(p->rows); free(p->row_start);
    free(p->rt); free(p->rlam); free(p->rlamb); free(p->rs); free(p->rg); free(p->rrp); free(p->rsig); free(p->rrho0); free(p->rD);
    free(p->rdl); free(p->rds); free(p->rdt); free(p->rka); free(p->rkb); free(p->rw);
    free(p->Fm); free(p->Gm); free(p->Mm); free(p->bm); free(p->hk); free(p->Phi); free(p->Gam);
    free(p->Hx); free(p->Hu); free(p->gx); free(p->gu); free(p->rd); free(p->QQ); free(p->qq); free(p->cc);
    free(p->Ps); free(p->ps); free(p->Pis); free(p->Ks); free(p->Sinv); free(p->Ds); free(p->d0s);
    free(p->dX); free(p->dU); free(p->nun); free(p->nu); free(p->Xw); free(p->Uw);
    free(p->trXp); free(p->trUp); free(p->trXn); free(p->trUn);
    free(p->s_vec); free(p->mu_vec); free(p->xtol_vec); free(p->ftol_vec); free(p->ctol_vec);
    free(p);
}
void go_set_ipm_opts(go_problem* p, const go_ipm_opts* o) { p->io = *o; }
void go_set_trace(go_problem* p, int cap) {
    free(p->trXp); free(p->trUp); free(p->trXn); free(p->trUn);
    p->trace_cap = cap; p->n_trace = 0;
    p->trXp = malloc(sizeof(double) * (size_t)(cap ? cap : 1) * p->n * p->N); p->trXn = malloc(sizeof(double) * (size_t)(cap ? cap : 1) * p->n * p->N);
    p->trUp = malloc(sizeof(double) * (size_t)(cap ? cap : 1) * p->m * p->N); p->trUn = malloc(sizeof(double) * (size_t)(cap ? cap : 1) * p->m * p->N);
}
int go_get_trace(const go_problem* p, int t, double* Xp, double* Up, double* Xn, double* Un) {
    if (t < 0 || t >= p->n_trace) return -1;
    const size_t nx = (size_t)p->n * p->N, nu = (size_t)p->m * p->N;
    memcpy(Xp, p->trXp + t * nx, sizeof(double) * nx); memcpy(Up, p->trUp + t * nu, sizeof(double) * nu);
    memcpy(Xn, p->trXn + t * nx, sizeof(double) * nx); memcpy(Un, p->trUn + t * nu, sizeof(double) * nu);
    return 0;
}
int go_trace_len(const go_problem* p) { return p->n_trace; }

static void grow_hist(go_problem* p, int need) {
    if (need <= p->cap) return;
    int c = need + 64;
#define GROW(ptr) ptr = realloc(ptr, sizeof(*(ptr)) * (size_t)c)
    GROW(p->J_true); GROW(p->J_full); GROW(p->conv); GROW(p->Delta); GROW(p->omega); GROW(p->rho);
    GROW(p->accept); GROW(p->scp_status); GROW(p->solver_status); GROW(p->tr_sat); GROW(p->cvx_sat); GROW(p->ipm_it);
    p->cap = c;
}

int go_set_problem(go_problem* p, const double* x_init, const double* goal_lo, const double* goal_hi, double tf,
                   const double* X0, const double* U0) {
    const int n = p->n, m = p->m, N = p->N;
    memcpy(p->x_init, x_init, sizeof(double) * n);
    memcpy(p->goal_lo, goal_lo, sizeof(double) * n);
    memcpy(p->goal_hi, goal_hi, sizeof(double) * n);
    p->tf = tf; p->dt = tf / (N - 1); /* types.jl:235 */
    if (X0 && U0) { memcpy(p->X, X0, sizeof(double) * n * N); memcpy(p->U, U0, sizeof(double) * m * N); }
    else go_init_straightline(p, p->X, p->U);
    /* SCPSolution(SCPP, traj_init) (types.jl:233) and SCPParam_GuSTO ctor (scp_gusto.jl:21-23) */
    free_hist(p);
    grow_hist(p, 8);
    p->iterations = 0; p->converged = 0; p->successful = 0; p->stop_reason = GO_STOP_MAXITER; p->total_ipm = 0; p->warm = 0;
    p->nJ_true = 0; p->nJ_full = 0; p->n_trace = 0;
    p->n_hist = 1;
    p->solver_status[0] = GO_SOLVER_NA; p->scp_status[0] = GO_SCP_NA; p->accept[0] = 1; p->conv[0] = 0.0; p->ipm_it[0] = 0;
    p->Delta[0] = p->sp.Delta0; p->omega[0] = p->sp.omega0; p->tr_sat[0] = 0; p->cvx_sat[0] = 0;
    p->n_rho = 1; p->rho[0] = 0.0;
    memset(p->dual, 0, sizeof(p->dual));
    return 0;
}

int go_solve(go_problem* p, int max_iter, int force) {
    const int n = p->n, m = p->m, N = p->N;
    const go_scp_params* sp = &p->sp;
    const int iter_cap = p->iterations + max_iter; /* scp_gusto.jl:67 */
    grow_hist(p, p->n_hist + max_iter + p->nJ_true + 8);
    double* Xn = malloc(sizeof(double) * n * N), *Un = malloc(sizeof(double) * m * N);
    p->J_true[p->nJ_true++] = go_cost_true(p, p->U);         /* :73 */
    p->J_full[p->nJ_full++] = p->J_true[p->nJ_true - 1];     /* :74 */
    p->rho[p->n_rho++] = go_trust_region_ratio(p, p->X, p->U, p->X, p->U); /* :75 */
    p->toggle = p->Delta[p->n_hist - 1] / 8 + p->mp.clearance; /* :76 */
    p->stop_reason = GO_STOP_MAXITER;
    while (p->iterations < iter_cap) {
        grow_hist(p, p->n_hist + p->nJ_true + 8);
        const int h = p->n_hist; /* index of the entries pushed this trip */
        const double Delta = p->Delta[h - 1], omega = p->omega[h - 1];
        go_sub_info info;
        int st = ipm_solve(p, p->X, p->U, Delta, omega, p->toggle, &info); /* :82-104 */
        p->total_ipm += info.iters;
        p->solver_status[h] = st; p->ipm_it[h] = info.iters;
        if (st != GO_SOLVER_OPTIMAL && st != GO_SOLVER_ALMOST) { /* :106-111 */
            p->n_hist = h; /* only solver_status/iter_elapsed_times are pushed in the reference */
            p->stop_reason = GO_STOP_SUBPROBLEM_FAILED;
            break;
        }
        memcpy(Xn, p->Xw, sizeof(double) * n * N);
        memcpy(Un, p->Uw, sizeof(double) * m * N);
        if (p->n_trace < p->trace_cap) {
            const size_t t = (size_t)p->n_trace++;
            memcpy(p->trXp + t * n * N, p->X, sizeof(double) * n * N); memcpy(p->trUp + t * m * N, p->U, sizeof(double) * m * N);
            memcpy(p->trXn + t * n * N, Xn, sizeof(double) * n * N); memcpy(p->trUn + t * m * N, Un, sizeof(double) * m * N);
        }
        p->conv[h] = go_convergence_metric(p, Xn, p->X);                 /* :115 */
        p->J_full[p->nJ_full++] = info.obj;                               /* :116 */
        p->tr_sat[h] = trust_region_satisfied(p, Xn, p->X, Delta);        /* :120 */
        p->cvx_sat[h] = go_convex_ineq_satisfied(p, Xn, p->X, p->U, p->toggle); /* :121 */
        if (p->tr_sat[h]) {
            double rho = go_trust_region_ratio(p, Xn, Un, p->X, p->U);    /* :124 */
            p->rho[p->n_rho++] = rho;
            if (rho > sp->rho1) {
                p->scp_status[h] = GO_SCP_INACCURATE_MODEL; p->accept[h] = 0;
                p->Delta[h] = sp->beta_fail * Delta; p->omega[h] = omega;
            } else {
                p->accept[h] = 1;
                p->Delta[h] = (rho < sp->rho0) ? fmin(sp->beta_succ * Delta, sp->Delta0) : Delta;
                if (!p->cvx_sat[h]) { p->scp_status[h] = GO_SCP_VIOLATES_CONSTRAINTS; p->omega[h] = sp->gamma_fail * omega; }
                else { p->scp_status[h] = GO_SCP_OK; p->omega[h] = omega; }
            }
        } else {
            p->scp_status[h] = GO_SCP_TRUST_REGION_VIOLATED; p->accept[h] = 0;
            p->Delta[h] = Delta; p->omega[h] = sp->gamma_fail * omega;
        }
        if (p->accept[h]) {                                               /* :149-154 */
            p->J_true[p->nJ_true++] = go_cost_true(p, Un);
            memcpy(p->X, Xn, sizeof(double) * n * N);
            memcpy(p->U, Un, sizeof(double) * m * N);
        } else {
            p->J_true[p->nJ_true] = p->J_true[p->nJ_true - 1];
            p->nJ_true++;
        }
        p->toggle = p->Delta[h] / 8 + p->mp.clearance;                    /* :156 */
        p->n_hist = h + 1;
        p->iterations += 1;
        if (p->omega[h] > sp->omega_max) { p->stop_reason = GO_STOP_OMEGA_MAX; break; } /* :163-166 */
        if (!p->accept[h]) continue;
        if (p->iterations > 2 && p->conv[h - 1] + p->conv[h] <= sp->convergence_threshold) { /* :169-175 */
            p->converged = 1;
            if (p->cvx_sat[h]) p->successful = 1;
            if (!force) { p->stop_reason = GO_STOP_CONVERGED; break; }
        }
    }
    free(Xn); free(Un);
    return 0;
}

int go_get_traj(const go_problem* p, double* X, double* U) {
    if (X) memcpy(X, p->X, sizeof(double) * p->n * p->N);
    if (U) memcpy(U, p->U, sizeof(double) * p->m * p->N);
    return 0;
}
int go_get_status(const go_problem* p, int* iterations, int* converged, int* successful, int* stop_reason, int* total_ipm) {
    if (iterations) *iterations = p->iterations;
    if (converged) *converged = p->converged;
    if (successful) *successful = p->successful;
    if (stop_reason) *stop_reason = p->stop_reason;
    if (total_ipm) *total_ipm = p->total_ipm;
    return 0;
}
int go_hist_len(const go_problem* p) { return p->n_hist; }
int go_get_history(const go_problem* p, double* J_true, int* nJ_true, double* J_full, int* nJ_full, double* conv,
                   double* Delta, double* omega, double* rho, int* n_rho, int* accept, int* scp_status,
                   int* solver_status, int* tr_sat, int* cvx_sat, int* ipm_iters) {
    const int h = p->n_hist;
    if (J_true) memcpy(J_true, p->J_true, sizeof(double) * p->nJ_true);
    if (nJ_true) *nJ_true = p->nJ_true;
    if (J_full) memcpy(J_full, p->J_full, sizeof(double) * p->nJ_full);
    if (nJ_full) *nJ_full = p->nJ_full;
    if (conv) memcpy(conv, p->conv, sizeof(double) * h);
    if (Delta) memcpy(Delta, p->Delta, sizeof(double) * h);
    if (omega) memcpy(omega, p->omega, sizeof(double) * h);
    if (rho) memcpy(rho, p->rho, sizeof(double) * p->n_rho);
    if (n_rho) *n_rho = p->n_rho;
    if (accept) memcpy(accept, p->accept, sizeof(int) * h);
    if (scp_status) memcpy(scp_status, p->scp_status, sizeof(int) * h);
    if (solver_status) memcpy(solver_status, p->solver_status, sizeof(int) * h);
    if (tr_sat) memcpy(tr_sat, p->tr_sat, sizeof(int) * h);
    if (cvx_sat) memcpy(cvx_sat, p->cvx_sat, sizeof(int) * h);
    if (ipm_iters) memcpy(ipm_iters, p->ipm_it, sizeof(int) * h);
    return 0;
}
int go_get_dual(const go_problem* p, double* dual) { memcpy(dual, p->dual, sizeof(double) * p->n); return 0; }

/* ------------------------------------------------------------------------------------------ */
/* Indirect shooting seeded by the SCP dual (src/shooting.jl:4-66, traj_opt.jl:4-45), DubinsCar:  */
/*   shooting_ode! / get_control  dubins_car.jl:259-280                                          */
/* The reference integrates with DifferentialEquations' default adaptive method and solves       */
/* F(p0) = x_goal - x(tf; p0) = 0 with NLsolve (trust region, finite-difference Jacobian,        */
/* ftol = 1e-3, 100 iterations); neither library is available, so both sides of the parity test  */
/* use the SAME stated scheme: classical RK4 with `substeps` steps per knot interval, Newton with */
/* a forward-difference Jacobian (h_j = 1e-6 max(1,|p_j|)) and halving line search on |F|_inf.    */
static void dubins_shoot_rhs(const go_problem* p, const double* z, double* dz) {
    const double v = p->mp.dubins_v, kk = p->mp.dubins_k, u = 0.5 * kk * z[5];
    dz[0] = v * cos(z[2]); dz[1] = v * sin(z[2]); dz[2] = kk * u;
    dz[3] = 0; dz[4] = 0; dz[5] = z[3] * v * sin(z[2]) - z[4] * v * cos(z[2]);
}
static void dubins_shoot_integrate(const go_problem* p, const double* p0, int substeps, double* xT, double* X, double* U) {
    const int N = p->N;
    double z[6], k1[6], k2[6], k3[6], k4[6], w[6];
    const double h = p->tf / ((N - 1) * (double)substeps);
    for (int i = 0; i < 3; i++) { z[i] = p->x_init[i]; z[3 + i] = p0[i]; }
    for (int k = 0; k < N; k++) {
        if (X) { for (int i = 0; i < 3; i++) X[k * 3 + i] = z[i]; U[k] = 0.5 * p->mp.dubins_k * z[5]; }
        if (k == N - 1) break;
        for (int s = 0; s < substeps; s++) {
            dubins_shoot_rhs(p, z, k1);
            for (int i = 0; i < 6; i++) w[i] = z[i] + 0.5 * h * k1[i];
            dubins_shoot_rhs(p, w, k2);
            for (int i = 0; i < 6; i++) w[i] = z[i] + 0.5 * h * k2[i];
            dubins_shoot_rhs(p, w, k3);
            for (int i = 0; i < 6; i++) w[i] = z[i] + h * k3[i];
            dubins_shoot_rhs(p, w, k4);
            for (int i = 0; i < 6; i++) z[i] += h / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
        }
    }
    for (int i = 0; i < 3; i++) xT[i] = z[i];
}
/* AstrobeeSE3Manifold: dynamics_shooting! / shooting_ode! / get_control, astrobee_se3_manifold.jl:831-895 (the row      */
/* contributions of :897-1006 are commented out in shooting_ode! at HEAD).  z = (r v q w | pr pv pq pw), 26 values;     */
/* get_control: F = pv / (2 mass), M = Jinv' pw / 2 (J diagonal).                                                      */
static void manifold_shoot_ctrl(const go_problem* p, const double* z, double* u) {
    for (int i = 0; i < 3; i++) { u[i] = z[13 + 3 + i] / (2.0 * p->mp.mass); u[3 + i] = z[13 + 10 + i] / p->mp.Jdiag[i] / 2.0; }
}
static void manifold_shoot_rhs(const go_problem* p, const double* z, double* dz) {
    const double *v = z + 3, *w = z + 10, *pr = z + 13, *pq = z + 13 + 6;
    const double qw = z[6], qx = z[7], qy = z[8], qz = z[9], wx = w[0], wy = w[1], wz = w[2];
    const double pqw = pq[0], pqx = pq[1], pqy = pq[2], pqz = pq[3];
    const double Jx = p->mp.Jdiag[0], Jy = p->mp.Jdiag[1], Jz = p->mp.Jdiag[2];
    double u[6];
    manifold_shoot_ctrl(p, z, u);
    for (int i = 0; i < 3; i++) { dz[i] = v[i]; dz[3 + i] = u[i] / p->mp.mass; }
    dz[6] = 0.5 * (-wx * qx - wy * qy - wz * qz);
    dz[7] = 0.5 * (wx * qw - wz * qy + wy * qz);
    dz[8] = 0.5 * (wy * qw + wz * qx - wx * qz);
    dz[9] = 0.5 * (wz * qw - wy * qx + wx * qy);
    {   /* Jinv (M - w x J w) */
        const double Jw[3] = {Jx * wx, Jy * wy, Jz * wz};
        const double c[3] = {wy * Jw[2] - wz * Jw[1], wz * Jw[0] - wx * Jw[2], wx * Jw[1] - wy * Jw[0]};
        dz[10] = (u[3] - c[0]) / Jx; dz[11] = (u[4] - c[1]) / Jy; dz[12] = (u[5] - c[2]) / Jz;
    }
    dz[13] = 0; dz[14] = 0; dz[15] = 0;
    for (int i = 0; i < 3; i++) dz[16 + i] = -pr[i];
    dz[19] = -0.5 * (pqx * wx + pqy * wy + pqz * wz);
    dz[20] = -0.5 * (-pqw * wx + pqy * wz - pqz * wy);
    dz[21] = -0.5 * (-pqw * wy - pqx * wz + pqz * wx);
    dz[22] = -0.5 * (-pqw * wz + pqx * wy - pqy * wx);
    dz[23] = -0.5 * (-pqw * qx + pqx * qw - pqy * qz + pqz * qy);
    dz[24] = -0.5 * (-pqw * qy + pqx * qz + pqy * qw - pqz * qx);
    dz[25] = -0.5 * (-pqw * qz - pqx * qy + pqy * qx + pqz * qw);
}
static void manifold_shoot_integrate(const go_problem* p, const double* p0, int substeps, double* xT, double* X, double* U) {
    const int N = p->N;
    double z[26], k1[26], k2[26], k3[26], k4[26], w[26];
    const double h = p->tf / ((N - 1) * (double)substeps);
    for (int i = 0; i < 13; i++) { z[i] = p->x_init[i]; z[13 + i] = p0[i]; }
    for (int k = 0; k < N; k++) {
        if (X) { for (int i = 0; i < 13; i++) X[k * 13 + i] = z[i]; manifold_shoot_ctrl(p, z, U + k * 6); }
        if (k == N - 1) break;
        for (int s = 0; s < substeps; s++) {
            manifold_shoot_rhs(p, z, k1);
            for (int i = 0; i < 26; i++) w[i] = z[i] + 0.5 * h * k1[i];
            manifold_shoot_rhs(p, w, k2);
            for (int i = 0; i < 26; i++) w[i] = z[i] + 0.5 * h * k2[i];
            manifold_shoot_rhs(p, w, k3);
            for (int i = 0; i < 26; i++) w[i] = z[i] + h * k3[i];
            manifold_shoot_rhs(p, w, k4);
            for (int i = 0; i < 26; i++) z[i] += h / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
        }
    }
    for (int i = 0; i < 13; i++) xT[i] = z[i];
}
static void shoot_integrate(const go_problem* p, const double* p0, int substeps, double* xT, double* X, double* U) {
    if (p->model == GO_DUBINS_CAR) dubins_shoot_integrate(p, p0, substeps, xT, X, U);
    else manifold_shoot_integrate(p, p0, substeps, xT, X, U);
}
/* dp = -J^-1 F: Cramer's rule for n = 3 (DubinsCar), Gaussian elimination with partial pivoting otherwise; 0 if singular */
static int shoot_newton_step(int n, double* J, const double* F, double* dp) {
    if (n == 3) {
        const double det = J[0] * (J[4] * J[8] - J[5] * J[7]) - J[1] * (J[3] * J[8] - J[5] * J[6]) + J[2] * (J[3] * J[7] - J[4] * J[6]);
        if (!(fabs(det) > 1e-300) || !isfinite(det)) return 0;
        dp[0] = -(F[0] * (J[4] * J[8] - J[5] * J[7]) - J[1] * (F[1] * J[8] - J[5] * F[2]) + J[2] * (F[1] * J[7] - J[4] * F[2])) / det;
        dp[1] = -(J[0] * (F[1] * J[8] - J[5] * F[2]) - F[0] * (J[3] * J[8] - J[5] * J[6]) + J[2] * (J[3] * F[2] - F[1] * J[6])) / det;
        dp[2] = -(J[0] * (J[4] * F[2] - F[1] * J[7]) - J[1] * (J[3] * F[2] - F[1] * J[6]) + F[0] * (J[3] * J[7] - J[4] * J[6])) / det;
        return 1;
    }
    /* Gaussian elimination with COMPLETE pivoting, stopped at the numerical rank: pivots below 1e-6 max|J| -- the accuracy
     * of a forward-difference Jacobian with h = 1e-6 -- are noise.  The costate of the quaternion along q itself does not
     * move the state (the flow keeps |q|), so dF/dp0 is rank deficient by one; the free component of the step is left
     * at 0 (NLsolve's trust region bounds it in the reference) instead of being divided by noise. */
    double r[GO_MAXN], jmax = 0;
    int perm[GO_MAXN], rank = 0;
    for (int i = 0; i < n; i++) { r[i] = -F[i]; perm[i] = i; dp[i] = 0.0; }
    for (int i = 0; i < n * n; i++) jmax = fmax(jmax, fabs(J[i]));
    if (!(jmax > 0) || !isfinite(jmax)) return 0;
    for (int c = 0; c < n; c++) {
        int pi = c, pj = c;
        double best = -1.0;
        for (int i = c; i < n; i++)
            for (int j = c; j < n; j++) {
                const double a = fabs(J[i * n + j]);
                if (!(a == a)) return 0;
                if (a > best) { best = a; pi = i; pj = j; }
            }
        if (!(best > 1e-6 * jmax)) break;
        if (pi != c) {
            for (int j = 0; j < n; j++) { const double t = J[c * n + j]; J[c * n + j] = J[pi * n + j]; J[pi * n + j] = t; }
            const double t = r[c]; r[c] = r[pi]; r[pi] = t;
        }
        if (pj != c) {
            for (int i = 0; i < n; i++) { const double t = J[i * n + c]; J[i * n + c] = J[i * n + pj]; J[i * n + pj] = t; }
            const int t = perm[c]; perm[c] = perm[pj]; perm[pj] = t;
        }
        for (int i = c + 1; i < n; i++) {
            const double f = J[i * n + c] / J[c * n + c];
            for (int j = c; j < n; j++) J[i * n + j] -= f * J[c * n + j];
            r[i] -= f * r[c];
        }
        rank = c + 1;
    }
    if (rank == 0) return 0;
    double y[GO_MAXN];
    for (int c = rank - 1; c >= 0; c--) {
        double sacc = r[c];
        for (int j = c + 1; j < rank; j++) sacc -= J[c * n + j] * y[j];
        y[c] = sacc / J[c * n + c];
    }
    for (int c = 0; c < rank; c++) dp[perm[c]] = y[c];
    return 1;
}
/* returns 1 = :Optimal (|F|_inf <= ftol), 0 = :Diverged, -1 = the model has no shooting ODE.  p0 NULL -> SCPS.dual.   */
/* X [N][n], U [N][m] may be NULL.                                                                                      */
int go_shoot(go_problem* p, const double* p0, int substeps, int max_newton, double ftol, double* p_out, double* X, double* U,
             int* newton_iters, double* resid) {
    if (p->model != GO_DUBINS_CAR && p->model != GO_ASTROBEE_SE3_MANIFOLD) return -1;
    const int n = p->n;
    double pv[GO_MAXN], xg[GO_MAXN], F[GO_MAXN], xT[GO_MAXN], nf = 0;
    for (int i = 0; i < n; i++) {
        pv[i] = p0 ? p0[i] : p->dual[i];
        const double lo = p->goal_lo[i], hi = p->goal_hi[i];
        xg[i] = (isfinite(lo) && isfinite(hi)) ? 0.5 * (lo + hi) : 0.0;      /* ShootingProblem ctor, types.jl:219-226 */
    }
    int it = 0, ok = 0;
    shoot_integrate(p, pv, substeps, xT, NULL, NULL);
    for (int i = 0; i < n; i++) { F[i] = xg[i] - xT[i]; nf = (F[i] != F[i] || nf != nf) ? NAN : fmax(nf, fabs(F[i])); }
    for (;; it++) {
        if (!(nf == nf) || !isfinite(nf)) break;
        if (nf <= ftol) { ok = 1; break; }
        if (it >= max_newton) break;
        double J[GO_MAXN * GO_MAXN], Fj[GO_MAXN], pj[GO_MAXN];
        for (int j = 0; j < n; j++) {
            const double h = 1e-6 * fmax(1.0, fabs(pv[j]));
            for (int i = 0; i < n; i++) pj[i] = pv[i];
            pj[j] += h;
            shoot_integrate(p, pj, substeps, xT, NULL, NULL);
            for (int i = 0; i < n; i++) { Fj[i] = xg[i] - xT[i]; J[i * n + j] = (Fj[i] - F[i]) / h; }
        }
        double dp[GO_MAXN];
        if (!shoot_newton_step(n, J, F, dp)) break;
        double a = 1.0, nn = 0, Fn[GO_MAXN], pn[GO_MAXN];
        int dec = 0;
        while (a > 1e-4) {
            for (int i = 0; i < n; i++) pn[i] = pv[i] + a * dp[i];
            shoot_integrate(p, pn, substeps, xT, NULL, NULL);
            nn = 0;
            for (int i = 0; i < n; i++) { Fn[i] = xg[i] - xT[i]; nn = (Fn[i] != Fn[i] || nn != nn) ? NAN : fmax(nn, fabs(Fn[i])); }
            if (nn < nf) { dec = 1; break; }
            a *= 0.5;
        }
        if (!dec) break;
        for (int i = 0; i < n; i++) { pv[i] = pn[i]; F[i] = Fn[i]; }
        nf = nn;
    }
    if (newton_iters) *newton_iters = it;
    if (resid) *resid = nf;
    if (p_out) for (int i = 0; i < n; i++) p_out[i] = pv[i];
    if (ok && X) shoot_integrate(p, pv, substeps, xT, X, U);
    return ok;
}

int go_subproblem(go_problem* p, const double* Xp, const double* Up, double Delta, double omega, double toggle,
                  double* Xn, double* Un, double* dual, go_sub_info* info) {
    const int warm_saved = p->warm; /* the hook always starts cold and leaves the SCP run's state alone */
    p->warm = 0;
    int st = ipm_solve(p, Xp, Up, Delta, omega, toggle, info);
    p->warm = warm_saved;
    if (Xn) memcpy(Xn, p->Xw, sizeof(double) * p->n * p->N);
    if (Un) memcpy(Un, p->Uw, sizeof(double) * p->m * p->N);
    if (dual) memcpy(dual, p->dual, sizeof(double) * p->n);
    return st;
}

/* ------------------------------------------------------------------------------------------ */
/* TrajOpt (src/scp/scp_trajopt.jl)                                                               */
void go_default_trajopt_params(int model, go_trajopt_params* tp) {
    memset(tp, 0, sizeof(*tp));
    tp->mu0 = 1.0; tp->c = 10.0; tp->tau_plus = 2.0; tp->tau_minus = 0.5; tp->k = 5.0; tp->ftol = 0.01; tp->ctol = 0.01;
    tp->max_penalty_iteration = 5; tp->max_convex_iteration = 5; tp->max_trust_iteration = 5;
    if (model == GO_FREEFLYER_SE2) { tp->s0 = 1.0; tp->xtol = 0.1; }   /* freeflyer_se2.jl:49-64 */
    else { tp->s0 = 10.0; tp->xtol = 0.01; }                            /* astrobee_se3.jl:50-65, astrobee_se3_manifold.jl:56-70 */
}
go_problem* go_create_trajopt(int model, int N, const go_model_params* mp, const go_trajopt_params* tp, int n_box,
                              const double* box, int n_sph, const double* sph) {
    if (model != GO_FREEFLYER_SE2 && model != GO_ASTROBEE_SE3 && model != GO_ASTROBEE_SE3_MANIFOLD) return NULL;
    go_scp_params sp; go_model_params dmp;
    go_default_params(model, &sp, &dmp);
    go_problem* p = create_impl(model, N, &sp, mp ? mp : &dmp, n_box, box, n_sph, sph, 1);
    if (!p) return NULL;
    if (tp) p->tp = *tp; else go_default_trajopt_params(model, &p->tp);
    return p;
}
int go_trajopt_subproblem(go_problem* p, const double* Xp, const double* Up, double mu, double s, double* Xn, double* Un,
                          double* dual, go_sub_info* info) {
    if (!p->trajopt) return -1;
    p->warm = 0;
    int st = ipm_solve(p, Xp, Up, s, mu, p->mp.clearance + 1.0, info);
    memcpy(Xn, p->Xw, sizeof(double) * p->n * p->N);
    memcpy(Un, p->Uw, sizeof(double) * p->m * p->N);
    if (dual) memcpy(dual, p->dual, sizeof(double) * p->n);
    return st;
}
/* trust_region_ratio_trajopt (freeflyer_se2.jl:429-467, astrobee_se3.jl:419-460).  As written the dynamics terms read
 * (Xp[:,k]-Xp[:,k])/dtp -- zero -- and call a 5-argument dynamics_constraints that does not exist; what is meant is the
 * forward difference (X[:,k+1]-X[:,k])/dt and the linearised trapezoid defect of interval k.  The obstacle terms take the
 * linearisation at traj_prev, as in trust_region_ratio_gusto (the file evaluates distance and normal at the new point). */
double go_trajopt_ratio(go_problem* p, const double* X, const double* U, const double* Xp, const double* Up) {
    const int n = p->n, m = p->m, N = p->N, d = ws_dim(p);
    double num = 0, den = 0;
    double f[NX], fp[NX], fp1[NX], A[NX * NX], A1[NX * NX], B[NX * NU], B1[NX * NU], a0[NX], a1[NX];
    for (int k = 0; k < N - 1; k++) {
        go_dynamics(p, Xp + k * n, Up + k * m, fp, A, B);
        go_dynamics(p, Xp + (k + 1) * n, Up + (k + 1) * m, fp1, A1, B1);
        go_dynamics(p, X + k * n, U + k * m, f, NULL, NULL);
        double po = 0, pn = 0, ph = 0;
        for (int i = 0; i < n; i++) {
            a0[i] = fp[i]; a1[i] = fp1[i];
            for (int j = 0; j < n; j++) {
                a0[i] += A[i * n + j] * (X[k * n + j] - Xp[k * n + j]);
                a1[i] += A1[i * n + j] * (X[(k + 1) * n + j] - Xp[(k + 1) * n + j]);
            }
            for (int j = 0; j < p->m0; j++) {
                a0[i] += B[i * m + j] * (U[k * m + j] - Up[k * m + j]);
                a1[i] += B1[i * m + j] * (U[(k + 1) * m + j] - Up[(k + 1) * m + j]);
            }
        }
        for (int i = 0; i < n; i++) {
            po += fabs(fp[i] - (Xp[(k + 1) * n + i] - Xp[k * n + i]) / p->dt);
            pn += fabs(f[i] - (X[(k + 1) * n + i] - X[k * n + i]) / p->dt);
            ph += fabs(X[(k + 1) * n + i] - X[k * n + i] - 0.5 * p->dt * (a0[i] + a1[i]));
        }
        num += po - pn; den += po - ph;
    }
    for (int k = 0; k < N; k++) {
        const double *r0 = Xp + k * n, *r = X + k * n;
        for (int c = 0; c < p->mp.n_robot_comp; c++)
            for (int i = 0; i < p->n_obs; i++) {
                double nh[3];
                const double d0 = go_signed_distance(p, c, r0, i, nh), d1 = go_signed_distance(p, c, r, i, NULL);
                double lin = d0;
                for (int j = 0; j < d; j++) lin += nh[j] * (r[j] - r0[j]);
                const double po = p->mp.clearance - d0, pn = p->mp.clearance - d1, ph = p->mp.clearance - lin;
                num += po - pn; den += po - ph;
            }
    }
    return num / den;
}
/* evaluate_ctol (scp_trajopt.jl:289-312): per class of constraints the largest change and the largest value over its
 * members, summed over the classes.  (As written the class entered last is never added; every class counts here.) */
static void defect_true(go_problem* p, const double* X, const double* U, int k, double* F) {
    double f0[NX], f1[NX];
    go_dynamics(p, X + k * p->n, U + k * p->m, f0, NULL, NULL);
    go_dynamics(p, X + (k + 1) * p->n, U + (k + 1) * p->m, f1, NULL, NULL);
    for (int i = 0; i < p->n; i++) F[i] = X[(k + 1) * p->n + i] - X[k * p->n + i] - 0.5 * p->dt * (f0[i] + f1[i]);
}
double go_trajopt_ctol(go_problem* p, const double* X, const double* U, const double* Xp, const double* Up) {
    const int n = p->n, N = p->N;
    const int is2 = p->model == GO_FREEFLYER_SE2, man = p->model == GO_ASTROBEE_SE3_MANIFOLD;
    const int nv = is2 ? 2 : 3, iw = is2 ? 5 : (man ? 10 : 9), nw = is2 ? 1 : 3;
    double JN = 0, JD = 0, tn, td;
    if (man) {   /* csi_orientation_sign (-qw) and cse_quaternion_norm(traj, traj) = |q_k| - 1 (manifold.jl:308-319) */
        tn = td = 0;
        for (int k = 0; k < N; k++) { tn = fmax(tn, fabs(-X[k * n + 6] + Xp[k * n + 6])); td = fmax(td, fabs(-X[k * n + 6])); }
        JN += tn; JD += td;
        tn = td = 0;
        for (int k = 0; k < N; k++) {
            double a = 0, b = 0;
            for (int j = 0; j < 4; j++) { a += X[k * n + 6 + j] * X[k * n + 6 + j]; b += Xp[k * n + 6 + j] * Xp[k * n + 6 + j]; }
            const double g = sqrt(a) - 1.0, gp = sqrt(b) - 1.0;
            tn = fmax(tn, fabs(g - gp)); td = fmax(td, fabs(g));
        }
        JN += tn; JD += td;
    }
    /* csi_translational_velocity_bound, csi_angular_velocity_bound */
    for (int cls = 0; cls < 2; cls++) {
        tn = td = 0;
        const int i0 = cls ? iw : 3, cnt = cls ? nw : nv;
        const double lim = cls ? p->mp.hard_limit_omega : p->mp.hard_limit_vel;
        for (int k = 0; k < N; k++) {
            double g = -lim * lim, gp = -lim * lim;
            for (int j = 0; j < cnt; j++) { g += X[k * n + i0 + j] * X[k * n + i0 + j]; gp += Xp[k * n + i0 + j] * Xp[k * n + i0 + j]; }
            tn = fmax(tn, fabs(g - gp)); td = fmax(td, fabs(g));
        }
        JN += tn; JD += td;
    }
    /* ncsi_body_obstacle_avoidance_constraints: clearance - dist */
    if (p->n_obs > 0) {
        tn = td = 0;
        for (int k = 0; k < N; k++)
            for (int i = 0; i < p->n_obs; i++) {
                const double g = p->mp.clearance - go_signed_distance(p, 0, X + k * n, i, NULL);
                const double gp = p->mp.clearance - go_signed_distance(p, 0, Xp + k * n, i, NULL);
                tn = fmax(tn, fabs(g - gp)); td = fmax(td, fabs(g));
            }
        JN += tn; JD += td;
    }
    /* csbci_goal_constraints (BoxGoal rows, :array) */
    {
        double a = 0, b = 0; int any = 0;
        for (int i = 0; i < n; i++) {
            const double lo = p->goal_lo[i], hi = p->goal_hi[i];
            if (lo == hi) continue;
            const double x = X[(N - 1) * n + i], xq = Xp[(N - 1) * n + i];
            if (isfinite(hi)) { a += (x - xq) * (x - xq); b += (x - hi) * (x - hi); any = 1; }
            if (isfinite(lo)) { a += (x - xq) * (x - xq); b += (lo - x) * (lo - x); any = 1; }
        }
        if (any) { JN += sqrt(a); JD += sqrt(b); }
    }
    /* dynamics_constraints(traj, traj, k): the trapezoid defect of the trajectory itself */
    tn = td = 0;
    for (int k = 0; k < N - 1; k++) {
        double F[NX], Fp[NX], a = 0, b = 0;
        defect_true(p, X, U, k, F); defect_true(p, Xp, Up, k, Fp);
        for (int i = 0; i < n; i++) { a += (F[i] - Fp[i]) * (F[i] - Fp[i]); b += F[i] * F[i]; }
        tn = fmax(tn, sqrt(a)); td = fmax(td, sqrt(b));
    }
    JN += tn; JD += td;
    return JN / JD;
}
static void to_grow(go_problem* p, int need) {
    if (need <= p->to_cap) return;
    const int c = need + 64;
    p->s_vec = realloc(p->s_vec, sizeof(double) * c); p->mu_vec = realloc(p->mu_vec, sizeof(double) * c);
    p->xtol_vec = realloc(p->xtol_vec, sizeof(double) * c); p->ftol_vec = realloc(p->ftol_vec, sizeof(double) * c);
    p->ctol_vec = realloc(p->ctol_vec, sizeof(double) * c);
    p->to_cap = c;
}
int go_solve_trajopt(go_problem* p, int max_iter) {
    if (!p->trajopt) return -1;
    const int n = p->n, m = p->m, N = p->N;
    const go_trajopt_params* tp = &p->tp;
    const int total = tp->max_penalty_iteration * tp->max_convex_iteration * tp->max_trust_iteration;
    const size_t nx = sizeof(double) * n * N, nu = sizeof(double) * m * N;
    grow_hist(p, p->n_hist + p->nJ_true + total + 8);
    to_grow(p, 2 * total + 16);
    /* SCPParam_TrajOpt ctor (:25-28): rho_vec = [0.], mu_vec = [mu0], s_vec = [s0], xtol_vec = ftol_vec = ctol_vec = [0.] */
    p->n_solves = 0; p->n_mu = 1; p->n_xtol = 1; p->n_ftol = 1; p->n_ctol = 1;
    p->mu_vec[0] = tp->mu0; p->s_vec[0] = tp->s0; p->xtol_vec[0] = p->ftol_vec[0] = p->ctol_vec[0] = 0.0;
    p->n_rho = 1; p->rho[0] = 0.0;
    double *Xn = malloc(nx), *Un = malloc(nu), *Xpen = malloc(nx), *Upen = malloc(nu), *Xcvx = malloc(nx), *Ucvx = malloc(nu);
    p->J_true[p->nJ_true++] = go_cost_true(p, p->U);                                   /* :63 */
    p->toggle = p->mp.clearance + 1.0;                                                  /* :64 */
    int constraints_satisfied = 0, xtol_satisfied = 0, solves = 0, stop = 0;
    p->stop_reason = GO_STOP_MAXITER;
    for (int pi = 0; pi < tp->max_penalty_iteration && !stop; pi++) {
        if (constraints_satisfied) break;
        memcpy(Xpen, p->X, nx); memcpy(Upen, p->U, nu);                                /* :73 (old_penalty_traj: a copy, not the alias of :67) */
        for (int ci = 0; ci < tp->max_convex_iteration && !stop; ci++) {
            memcpy(Xcvx, p->X, nx); memcpy(Ucvx, p->U, nu);                            /* :76 */
            if (constraints_satisfied) break;
            if (xtol_satisfied) { xtol_satisfied = 0; break; }
            for (int ti = 0; ti < tp->max_trust_iteration; ti++) {
                if (solves >= max_iter) { stop = 1; break; }
                const double mu = p->mu_vec[p->n_mu - 1], s = p->s_vec[p->n_solves];
                go_sub_info info;
                p->warm = 0;
                const int st = ipm_solve(p, p->X, p->U, s, mu, p->toggle, &info);      /* :98-110 */
                const int h = p->n_hist;
                p->solver_status[h] = st; p->ipm_it[h] = info.iters; p->total_ipm += info.iters;
                p->Delta[h] = s; p->omega[h] = mu;   /* (the trip's own s and mu, for the lock-step tests) */
                p->accept[h] = 1; p->scp_status[h] = GO_SCP_NA; p->tr_sat[h] = 1; p->cvx_sat[h] = 0;
                if (p->n_trace < p->trace_cap) {     /* go_set_trace: (traj before the trip, its optimum) */
                    const size_t t = p->n_trace++;
                    memcpy(p->trXp + t * n * N, p->X, nx); memcpy(p->trUp + t * m * N, p->U, nu);
                    memcpy(p->trXn + t * n * N, p->Xw, nx); memcpy(p->trUn + t * m * N, p->Uw, nu);
                }
                if (st != GO_SOLVER_OPTIMAL && st != GO_SOLVER_ALMOST) {                /* (:107-110 warns and goes on with the values) */
                    p->stop_reason = GO_STOP_SUBPROBLEM_FAILED; stop = 1; break;
                }
                memcpy(Xn, p->Xw, nx); memcpy(Un, p->Uw, nu);
                const double xt = go_convergence_metric(p, Xn, Xcvx);                   /* evaluate_xtol :120-121 */
                p->xtol_vec[p->n_xtol++] = xt; p->conv[h] = xt;
                p->J_full[p->nJ_full++] = info.obj;                                     /* :122 */
                const double rho = go_trajopt_ratio(p, Xn, Un, Xcvx, Ucvx);             /* :127 */
                p->rho[p->n_rho++] = rho;
                p->s_vec[p->n_solves + 1] = (rho > tp->c) ? tp->tau_plus * s : tp->tau_minus * s;   /* :128-132 */
                memcpy(p->X, Xn, nx); memcpy(p->U, Un, nu);                             /* :134: every step is taken */
                p->J_true[p->nJ_true++] = go_cost_true(p, p->U);
                p->n_hist = h + 1; p->n_solves++; p->iterations++; solves++;
                if (p->s_vec[p->n_solves] < tp->xtol) { xtol_satisfied = 1; break; }    /* :140-143 */
            }
            if (stop) break;
            const double Jn = go_cost_true(p, p->U), Jo = go_cost_true(p, Ucvx);
            p->ftol_vec[p->n_ftol++] = fabs(Jn - Jo) / fabs(Jn);                        /* evaluate_ftol :146 */
            p->xtol_vec[p->n_xtol++] = go_convergence_metric(p, p->X, Xcvx);            /* :147 */
            if (p->ftol_vec[p->n_ftol - 1] < tp->ftol || p->xtol_vec[p->n_xtol - 1] < tp->xtol) {   /* :148 (`xtol[end]` means xtol_vec[end]) */
                constraints_satisfied = 1; break;
            }
        }
        if (stop) break;
        const double ct = go_trajopt_ctol(p, p->X, p->U, Xpen, Upen);                   /* :155 */
        p->ctol_vec[p->n_ctol++] = ct;
        if (ct < tp->ctol) { constraints_satisfied = 1; p->converged = 1; p->stop_reason = GO_STOP_CONVERGED; break; }
        p->mu_vec[p->n_mu] = p->mu_vec[p->n_mu - 1] * tp->k; p->n_mu++;                 /* :161 */
    }
    free(Xn); free(Un); free(Xpen); free(Upen); free(Xcvx); free(Ucvx);
    return solves;
}
int go_get_trajopt_history(const go_problem* p, double* s_vec, int* n_s, double* mu_vec, int* n_mu, double* xtol_vec, int* n_xtol,
                           double* ftol_vec, int* n_ftol, double* ctol_vec, int* n_ctol) {
    if (!p->trajopt) return -1;
    if (n_s) *n_s = p->n_solves + 1;
    if (n_mu) *n_mu = p->n_mu;
    if (n_xtol) *n_xtol = p->n_xtol;
    if (n_ftol) *n_ftol = p->n_ftol;
    if (n_ctol) *n_ctol = p->n_ctol;
    if (s_vec) memcpy(s_vec, p->s_vec, sizeof(double) * (p->n_solves + 1));
    if (mu_vec) memcpy(mu_vec, p->mu_vec, sizeof(double) * p->n_mu);
    if (xtol_vec) memcpy(xtol_vec, p->xtol_vec, sizeof(double) * p->n_xtol);
    if (ftol_vec) memcpy(ftol_vec, p->ftol_vec, sizeof(double) * p->n_ftol);
    if (ctol_vec) memcpy(ctol_vec, p->ctol_vec, sizeof(double) * p->n_ctol);
    return 0;
}

int go_rows_count(const go_problem* p) { return p->nrows; }
int go_rows_get(const go_problem* p, int i, int* k, int* isu, int* kind, int* nnz, int* idx, double* a, double* v0,
                double* b, double* c0, double* mul, double* off, double* slack, double* lam) {
    if (i < 0 || i >= p->nrows) return -1;
    const go_row* r = &p->rows[i];
    *k = r->k; *isu = r->isu; *kind = r->kind; *nnz = r->nnz;
    for (int j = 0; j < r->nnz; j++) { idx[j] = r->idx[j]; a[j] = r->a[j]; v0[j] = r->v0[j]; b[j] = r->b[j]; }
    *c0 = r->c0; *mul = r->mul; *off = r->off;
    if (slack) *slack = p->rs[i];
    if (lam) *lam = p->rlam[i];
    return 0;
}

int go_solve_batch(int model, int N, const go_scp_params* sp, const go_model_params* mp, int n_box, const double* box,
                   int n_sph, const double* sph, int B, const double* x_init, const double* goal_lo,
                   const double* goal_hi, const double* tf, int max_iter, int nthreads, double* X, double* U,
                   int* converged, int* successful, int* iterations, int* ipm_iters) {
    int n, m;
    if (go_model_dims(model, &n, &m)) return -1;
#ifdef _OPENMP
    if (nthreads > 0) omp_set_num_threads(nthreads);
#else
    (void)nthreads;
#endif
#pragma omp parallel
    {
        go_problem* p = go_create(model, N, sp, mp, n_box, box, n_sph, sph);
#pragma omp for schedule(dynamic, 1)
        for (int b = 0; b < B; b++) {
            go_set_problem(p, x_init + (size_t)b * n, goal_lo + (size_t)b * n, goal_hi + (size_t)b * n, tf[b], NULL, NULL);
            go_solve(p, max_iter, 0);
            if (X) memcpy(X + (size_t)b * n * N, p->X, sizeof(double) * n * N);
            if (U) memcpy(U + (size_t)b * m * N, p->U, sizeof(double) * m * N);
            if (converged) converged[b] = p->converged;
            if (successful) successful[b] = p->successful;
            if (iterations) iterations[b] = p->iterations;
            if (ipm_iters) ipm_iters[b] = p->total_ipm;
        }
        go_destroy(p);
    }
    return 0;
}
