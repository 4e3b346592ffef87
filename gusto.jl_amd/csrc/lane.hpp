// lane.hpp -- the GuSTO solve with ONE LANE PER PROBLEM (64 problems per wavefront), for the models whose blocks are
// so small that a wave per problem leaves the machine idle: dubins_car (n = 3, m = 1, N = 30; BASELINE.json configs[2],
// "tiny state, stresses wavefront occupancy").  The wave-per-problem kernel (scp.hpp / ipm.hpp) gives such a problem 30 of
// 64 lanes for its stage-parallel work and ONE useful lane-entry per instruction of its sequential sweeps (a 3 x 3 Riccati
// stage costs it ~2.4 k cycles).  Here a lane owns a whole problem: the sweeps are plain loops over the knots with the
// recursion state (P, Pi, Gd, p, theta, dy) in that lane's registers, every arithmetic instruction does 64 problems' worth
// of work, and nothing crosses lanes -- no LDS, no reductions, no barriers.
//
// Same algorithm, statement for statement, as ipm.hpp / scp.hpp: scp_gusto.jl:49-176 around a
// Mehrotra predictor-corrector on the convex subproblem of scp_gusto.jl:178-314, the Newton system solved by the Riccati
// recursion on the trapezoid rows (ipm.hpp header).  What changes is the order of evaluation -- a lane walks its knots one
// after the other, so the phases that the wave kernel runs knot-parallel are FUSED into the sweeps:
//   pass A (backward)  previous step applied + residuals + row Hessians + stage cost + factor stage + the predictor's
//                      backward vector recurrence, all of knot k before knot k-1;
//   pass B (forward)   predictor forward recurrence + row steps (affine step length, Mehrotra terms, corrector row sums);
//   pass C (backward)  corrector backward recurrence;
//   pass D (forward)   corrector forward recurrence + costates + row steps + step length.
// Four passes over the knots per interior point iteration instead of eleven phases; the per-knot data a later pass needs
// goes through the lane workspace in HBM.
//
// Workspace layout: [wave][knot][entry][lane] -- one load / store instruction of a wave touches 64 consecutive doubles
// (512 B, fully coalesced), the entries of a knot are adjacent (DRAM pages, TLB), and the address of an access is a
// wave-uniform base (SGPRs) plus the lane's fixed 32-bit offset.  The kernel is bound by this traffic (~2.5 KB per knot and
// iteration), not by arithmetic: DESIGN.md section 3.
//
// Lanes of a wave run DIFFERENT problems with different trip and iteration counts: the kernel body is a per-lane state
// machine (trip start | one interior point iteration | trip end) under one wave-level loop, so a lane never waits for a
// neighbour's interior point method to finish -- only for the instruction stream both share.
#pragma once
#include "ipm.hpp"

namespace gusto {

// per-knot linearisation data the passes re-read: a cache of A_k (and of f_k's state part) instead of the trigonometric
// functions of the linearisation point (Dyn::lin_cache), and e_k = f_k - A_k xp_k - B up_k, so that the linearised xdot
// of an iterate is a_k(x, u) = e_k + A_k x + B u
template <int MODEL> struct LinK;
template <> struct LinK<GUSTO_DUBINS_CAR> {
    static constexpr int NC = 2, n = 3, m = 1;
    GD static void make(const gusto_model_params& mp, const double* xp, const double* up, double* c, double* e) {
        Dyn<GUSTO_DUBINS_CAR>::lin_cache(mp, xp, c);
        double f[n], A[n * n], B[n * m];
        Dyn<GUSTO_DUBINS_CAR>::f_cached(mp, c, up, f);
        Dyn<GUSTO_DUBINS_CAR>::A_cached(c, A);
        Dyn<GUSTO_DUBINS_CAR>::B(mp, B);
#pragma unroll
        for (int i = 0; i < n; i++) {
            double s = f[i];
#pragma unroll
            for (int j = 0; j < n; j++) s -= A[i * n + j] * xp[j];
#pragma unroll
            for (int j = 0; j < m; j++) s -= B[i * m + j] * up[j];
            e[i] = s;
        }
    }
    GD static void A(const gusto_model_params&, const double* c, double* A) { Dyn<GUSTO_DUBINS_CAR>::A_cached(c, A); }
    // M = (I - h A)^-1.  A has its only entries in column 2 of rows 0, 1 (A^2 = 0): the inverse is I + h A, exactly what the
    // Gauss-Jordan elimination of stage_M() returns for this matrix (unit pivots, one update per row)
    GD static void M(const double* A, double h, double* M) {
#pragma unroll
        for (int i = 0; i < n; i++)
#pragma unroll
            for (int j = 0; j < n; j++) M[i * n + j] = (i == j ? 1.0 : 0.0) + h * A[i * n + j];
    }
    // f(x, u) from the state's own sin / cos (trust_region_ratio evaluates the true dynamics at the new trajectory)
    GD static void f_true(const gusto_model_params& mp, const double* x, const double* u, double* f) {
        Dyn<GUSTO_DUBINS_CAR>::f(mp, x, u, f);
    }
};

// entries of one knot in the lane workspace (doubles per lane)
template <int MODEL> struct LaneLay {
    using T = MT<MODEL>;
    static_assert(!T::HAS_OBS && T::NDEF == 0, "lane-per-problem kernel: models without obstacle rows");
    static constexpr int n = T::n, m = T::m, NZ = n + m, NHX = n * (n + 1) / 2, NHM = m * (m + 1) / 2, NC = LinK<MODEL>::NC;
    static constexpr int NS = T::NFIX + 2 * n + T::NHU;   // row slots of a knot: fixed state rows, BoxGoal pairs, control rows
    static constexpr int eXW = 0, eUW = eXW + n, eNU = eUW + m, eNUN = eNU + n, eDX = eNUN + n, eDU = eDX + n, eXP = eDU + m,
                         eUP = eXP + n, eEE = eUP + m, eLC = eEE + n, eRD = eLC + NC, eQRD = eRD + n, eRV = eQRD + n,
                         ePIC = eRV + n, ePS = ePIC + n, eD0 = ePS + n, eKK = eD0 + m, eDD = eKK + m * n, eSI = eDD + m * n,
                         ePP = eSI + NHM, ePI = ePP + NHX, eGA = ePI + n * n, eGB = eGA + NZ, eRS = eGB + NZ,
                         EK = eRS + RS_NVAR * NS;
};

// the per-row interior point state of knot k of this lane's problem (the RowState of rows.hpp in the lane layout)
template <int NS> struct LaneRS {
    GPtr<double> kb;   // wave-uniform: row state of knot k, entry 0, lane 0
    int lane;
    GD auto& at(int var, int slot) const { return (kb + (size_t)((var * NS + slot) * 64))[lane]; }
};

template <int MODEL> struct LaneSolver {
    using T = MT<MODEL>;
    using Y = LaneLay<MODEL>;
    using LK = LinK<MODEL>;
    static constexpr int n = T::n, m = T::m, NZ = n + m, NHX = Y::NHX, NHM = Y::NHM, NC = Y::NC, NS = Y::NS, NP = T::NFIX + T::NHU;
    using RS_t = LaneRS<NS>;

    const KParams& P;
    GPtr<double> wb;   // wave-uniform base of this wave's workspace
    int lane, b, N;
    double dt, hdt;
    unsigned goalmask, boxmask;
    double xinit[n], gval[n];

    // ---- interior point state of the running subproblem (registers of the lane) ----
    double Delta, omega, kappa, muw;
    int it, status, ncomp, n_acc;
    double alpha_prev, mu, resp, resd, obj, mu_start;
    double mug[n], nu0[n];

    GD LaneSolver(const KParams& P_, double* ws, int wave, int lane_, int b_) : P(P_), lane(lane_), b(b_), N(P_.N) {
        wb = ws + (size_t)wave * (size_t)N * (size_t)(Y::EK * 64);
    }
    GD auto& W(int k, int e) const { return (wb + (size_t)((k * Y::EK + e) * 64))[lane]; }
    GD bool is_goal(int i) const { return (goalmask >> i) & 1u; }
    GD double wk_of(int k) const { return kappa * ((k == 0 || k == N - 1) ? hdt : dt); }

    GD void bind_problem() {
        dt = P.tf[b] / (N - 1);   // Trajectory(X,U,Tf): dt = Tf/(N-1), types.jl:235
        hdt = 0.5 * dt;
        goalmask = 0; boxmask = 0;
#pragma unroll
        for (int i = 0; i < n; i++) {
            const double lo = P.goal_lo[(size_t)b * n + i], hi = P.goal_hi[(size_t)b * n + i];
            xinit[i] = P.x_init[(size_t)b * n + i];
            gval[i] = lo;
            if (lo == hi) goalmask |= 1u << i;
            else if (isfinite(lo) || isfinite(hi)) boxmask |= 1u << i;
        }
    }
    GD void make_ctx(int k, double kap, double om, double De, RowCtx<MODEL>& c) const {
        c.P = &P; c.N = N; c.k = k; c.nslot = NS; c.kappa = kap; c.omega = om; c.Delta = De;
        c.xp = GPtr<const double>((const double*)nullptr); c.mask = 0;
        c.obs_nh = GPtr<const double>((const double*)nullptr); c.obs_c0 = GPtr<const double>((const double*)nullptr);
        c.goal_lo = P.goal_lo + (size_t)b * n; c.goal_hi = P.goal_hi + (size_t)b * n; c.boxmask = boxmask;
    }
    GD RS_t rs_of(int k) const { return RS_t{wb + (size_t)((k * Y::EK + Y::eRS) * 64), lane}; }

    // stage matrices of knot k from its linearisation cache: A, M = (I - h A)^-1, hb = h B, [Phi Gam] (knot 0: [0 | hb])
    struct Stage { double A[n * n], M[n * n], hb[n * m], Phi[n * n], Gam[n * m]; };
    GD void stage_of(int k, const double* lc, Stage& S) const {
        double B[n * m];
        LK::A(P.mp, lc, S.A);
        LK::M(S.A, hdt, S.M);
        Dyn<MODEL>::B(P.mp, B);
#pragma unroll
        for (int i = 0; i < n * m; i++) S.hb[i] = hdt * B[i];
#pragma unroll
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int j = 0; j < n; j++) S.Phi[i * n + j] = (k >= 1) ? 2.0 * S.M[i * n + j] - (i == j ? 1.0 : 0.0) : 0.0;
#pragma unroll
            for (int j = 0; j < m; j++) {
                double s = 0;
#pragma unroll
                for (int l = 0; l < n; l++) s += S.M[i * n + l] * S.hb[l * m + j];
                S.Gam[i * m + j] = (k >= 1) ? 2.0 * s : S.hb[i * m + j];
            }
        }
    }

    // ---- trajectory in / out of the ABI layout ([B][N][n], a lane's rows are 8 n N bytes apart: once per call) ----
    GD void load_traj() {
        const double* Xg = P.X + (size_t)b * N * n;
        const double* Ug = P.U + (size_t)b * N * m;
        for (int k = 0; k < N; k++) {
#pragma unroll
            for (int i = 0; i < n; i++) W(k, Y::eXP + i) = Xg[k * n + i];
#pragma unroll
            for (int i = 0; i < m; i++) W(k, Y::eUP + i) = Ug[k * m + i];
        }
    }
    GD void store_traj(double* Xg, double* Ug, int ex, int eu) const {
        for (int k = 0; k < N; k++) {
#pragma unroll
            for (int i = 0; i < n; i++) Xg[k * n + i] = W(k, ex + i);
#pragma unroll
            for (int i = 0; i < m; i++) Ug[k * m + i] = W(k, eu + i);
        }
    }
    // cost_true: trapezoid control effort (freeflyer_se2.jl:66-76)
    GD double cost_true(int eu) const {
        double J = 0, up[m];
#pragma unroll
        for (int j = 0; j < m; j++) up[j] = W(0, eu + j);
        for (int k = 1; k < N; k++) {
#pragma unroll
            for (int j = 0; j < m; j++) {
                const double u = W(k, eu + j);
                J += 0.5 * dt * (up[j] * up[j] + u * u);
                up[j] = u;
            }
        }
        return J;
    }
    // trust_region_ratio_gusto (dubins_car.jl:229-241): the linearised dynamics deliberately lack B du
    GD double trust_region_ratio(int ex, int eu) const {
        double num = 0, den = 0;
        for (int k = 0; k < N - 1; k++) {
            double x[n], u[m], xp[n], up[m], lc[NC], f[n], fp[n], A[n * n];
#pragma unroll
            for (int i = 0; i < n; i++) { x[i] = W(k, ex + i); xp[i] = W(k, Y::eXP + i); }
#pragma unroll
            for (int i = 0; i < m; i++) { u[i] = W(k, eu + i); up[i] = W(k, Y::eUP + i); }
            Dyn<MODEL>::lin_cache(P.mp, xp, lc);
            Dyn<MODEL>::f_cached(P.mp, lc, up, fp);
            LK::A(P.mp, lc, A);
            LK::f_true(P.mp, x, u, f);
            double a = 0, bb = 0;
#pragma unroll
            for (int i = 0; i < n; i++) {
                double lin = fp[i];
#pragma unroll
                for (int j = 0; j < n; j++) lin += A[i * n + j] * (x[j] - xp[j]);
                a += (f[i] - lin) * (f[i] - lin);
                bb += lin * lin;
            }
            num += sqrt(a); den += sqrt(bb);
        }
        return num / den;
    }

    // ---- trip start: linearise at (XP, UP) (scp_gusto.jl:95) and put the interior point method on its start point ----
    GD void lin_init(double Delta_, double omega_, double muw_) {
        Delta = Delta_; omega = omega_; muw = muw_;
        kappa = 1.0 / fmax(1.0, omega);
        it = 0; n_acc = 0; status = GUSTO_SOLVER_FAILED; alpha_prev = 0.0; mu = 0; resp = 0; resd = 0; obj = 0;
        int nc = 0;
#pragma unroll
        for (int i = 0; i < n; i++) { mug[i] = 0; nu0[i] = 0; }
        for (int k = 0; k < N; k++) {
            double xp[n], up[m], lc[NC], e[n], xs[n], us[m];
#pragma unroll
            for (int i = 0; i < n; i++) xp[i] = W(k, Y::eXP + i);
#pragma unroll
            for (int i = 0; i < m; i++) up[i] = W(k, Y::eUP + i);
            LK::make(P.mp, xp, up, lc, e);
#pragma unroll
            for (int i = 0; i < NC; i++) W(k, Y::eLC + i) = lc[i];
#pragma unroll
            for (int i = 0; i < n; i++) {
                W(k, Y::eEE + i) = e[i];
                xs[i] = (k == 0) ? xinit[i] : xp[i];   // warm start at traj_prev (scp_gusto.jl:100-102) with x_1 pinned
                W(k, Y::eXW + i) = xs[i];
                W(k, Y::eNU + i) = 0.0;
            }
#pragma unroll
            for (int i = 0; i < m; i++) { us[i] = up[i]; W(k, Y::eUW + i) = us[i]; }
            RowCtx<MODEL> ctx;
            make_ctx(k, kappa, omega, Delta, ctx);
            OpInitT<RS_t> op{rs_of(k), muw};
            visit_rows<MODEL>(ctx, xs, us, op);
            nc += op.ncomp;
        }
        ncomp = nc;
    }

    // knot k's iterate with the previous step applied (the update of an interior point iteration is folded into the next
    // pass A, as the wave kernel folds the row update into its residual pass), and its linearised xdot
    struct Knot { double x[n], u[m], nu[n], a[n], lc[NC]; };
    GD void load_knot(int k, Knot& q) const {
        const double ap = alpha_prev;
        const bool upd = ap != 0.0;
        double e[n];
#pragma unroll
        for (int i = 0; i < n; i++) {
            double x = W(k, Y::eXW + i), nu = W(k, Y::eNU + i);
            const double dx = W(k, Y::eDX + i), nn = W(k, Y::eNUN + i);
            x = upd ? x + ap * dx : x;
            nu = upd ? nu + ap * (nn - nu) : nu;
            q.x[i] = x; q.nu[i] = nu;
            W(k, Y::eXW + i) = x; W(k, Y::eNU + i) = nu;
            e[i] = W(k, Y::eEE + i);
        }
#pragma unroll
        for (int i = 0; i < m; i++) {
            double u = W(k, Y::eUW + i);
            const double du = W(k, Y::eDU + i);
            u = upd ? u + ap * du : u;
            q.u[i] = u;
            W(k, Y::eUW + i) = u;
        }
#pragma unroll
        for (int i = 0; i < NC; i++) q.lc[i] = W(k, Y::eLC + i);
        double A[n * n], B[n * m];
        LK::A(P.mp, q.lc, A);
        Dyn<MODEL>::B(P.mp, B);
#pragma unroll
        for (int i = 0; i < n; i++) {
            double s = e[i];
#pragma unroll
            for (int j = 0; j < n; j++) s += A[i * n + j] * q.x[j];
#pragma unroll
            for (int j = 0; j < m; j++) s += B[i * m + j] * q.u[j];
            q.a[i] = s;
        }
    }

    // what pass A leaves in registers for the rest of the iteration
    struct Fact {
        double Ginv[n * n];   // goal system inverse (identity on the coordinates without a point goal)
        double gterm[n];      // C M rd_{N-1} - rg of the goal rows
        double mugn[n];       // goal multipliers of the current Newton step
        double gxs[n];        // gx of knot 0 (stationarity of the pinned x_1)
        bool fail;
    };

    // ---- pass A ----
    GD void passA(Fact& F) {
        double Pm[n * n], Pi[n * n], Gd[n * n], pv[n], th[n], nu1[n];
#pragma unroll
        for (int i = 0; i < n * n; i++) { Pm[i] = 0; Pi[i] = 0; Gd[i] = 0; }
#pragma unroll
        for (int i = 0; i < n; i++) { pv[i] = 0; th[i] = 0; nu1[i] = 0; F.gterm[i] = 0; F.gxs[i] = 0; }
        double l_resp = 0, l_resd = 0, l_comp = 0, l_numax = 0, l_obj = 0;
        bool fail = false;
        Knot q, qm;
        load_knot(N - 1, q);
        for (int k = N - 1; k >= 0; k--) {
            if (k >= 1) load_knot(k - 1, qm);
            const double wk = wk_of(k);
            Stage S;
            stage_of(k, q.lc, S);
            // trapezoid residual of row k (freeflyer_se2.jl:160-172 in Newton form)
            double rdk[n];
#pragma unroll
            for (int i = 0; i < n; i++) {
                rdk[i] = (k >= 1) ? qm.x[i] - q.x[i] + hdt * (qm.a[i] + q.a[i]) : 0.0;
                l_resp = nanmax(l_resp, fabs(rdk[i]));
            }
            // rows of this knot: update by the previous step, residuals, Hessian blocks, predictor row sums
            double Hx[NHX], Hu[NHM], rdx[n], rdu[m], gx0[n], gu0[m];
#pragma unroll
            for (int i = 0; i < NHX; i++) Hx[i] = 0;
#pragma unroll
            for (int i = 0; i < NHM; i++) Hu[i] = 0;
#pragma unroll
            for (int i = 0; i < n; i++) { rdx[i] = 0; gx0[i] = 0; }
#pragma unroll
            for (int i = 0; i < m; i++) { rdu[i] = 0; gu0[i] = 0; }
            {
                RowCtx<MODEL> ctx;
                make_ctx(k, kappa, omega, Delta, ctx);
                const RS_t rs = rs_of(k);
                RowPre<NP> pre;
                pre.load(rs, T::NFIX, T::NFIX + 2 * n, [&](int var) {
                    return var == RS_T || var == RS_LAM || var == RS_S || var == RS_LAMB || var == RS_DT || var == RS_DL || var == RS_DS;
                });
                OpResidHess<n, m, NP, false, RS_t> op{rs, Hx, Hu, rdx, rdu, gx0, gu0, alpha_prev, &pre};
                visit_rows<MODEL>(ctx, q.x, q.u, op);
                l_comp += op.comp;
                l_resp = nanmax(l_resp, op.maxrp);
                l_obj += op.ssum;
            }
#pragma unroll
            for (int i = 0; i < m; i++) {
                Hu[sidx(i, i, m)] += 2 * wk; rdu[i] += 2 * wk * q.u[i];
                l_obj += wk * q.u[i] * q.u[i];
            }
            // + E^T nu: F_k^T nu_{k+1} - G_k^T nu_k on x, b_k^T (nu_{k+1} + nu_k) on u
            {
                double vs[n], vd[n];
#pragma unroll
                for (int i = 0; i < n; i++) {
                    const double n1 = (k + 1 < N) ? nu1[i] : 0.0, n0 = (k >= 1) ? q.nu[i] : 0.0;
                    vs[i] = n1 + n0; vd[i] = n1 - n0;
                    l_numax = fmax(l_numax, fabs(q.nu[i]));
                }
#pragma unroll
                for (int i = 0; i < n; i++) {
                    double s = vd[i];
#pragma unroll
                    for (int j = 0; j < n; j++) s += hdt * S.A[j * n + i] * vs[j];
                    rdx[i] += s;
                }
#pragma unroll
                for (int i = 0; i < m; i++) {
                    double s = 0;
#pragma unroll
                    for (int j = 0; j < n; j++) s += S.hb[j * m + i] * vs[j];
                    rdu[i] += s;
                }
            }
            if (k == N - 1) {
#pragma unroll
                for (int i = 0; i < n; i++) {
                    if (is_goal(i)) {
                        rdx[i] += mug[i];
                        l_resp = nanmax(l_resp, fabs(gval[i] - q.x[i]));
                    }
                }
            }
            if (k >= 1) {
#pragma unroll
                for (int i = 0; i < n; i++) l_resd = nanmax(l_resd, fabs(rdx[i]));
            }
#pragma unroll
            for (int i = 0; i < m; i++) l_resd = nanmax(l_resd, fabs(rdu[i]));

            // stage cost in (dy, du): QQ = [Qt, Qt b; ., Hu + b^T Qt b], Qt = M^T Hx M (knot 0, x_1 pinned: only H_u survives)
            double Qt[NHX], Qb[n * m], Suu[NHM];
            {
                double HM[n * n];
#pragma unroll
                for (int i = 0; i < n; i++)
#pragma unroll
                    for (int j = 0; j < n; j++) {
                        double s = 0;
#pragma unroll
                        for (int l = 0; l < n; l++) s += Hx[sidx(i, l, n)] * S.M[l * n + j];
                        HM[i * n + j] = s;
                    }
#pragma unroll
                for (int i = 0; i < n; i++)
#pragma unroll
                    for (int j = i; j < n; j++) {
                        double s = 0;
#pragma unroll
                        for (int l = 0; l < n; l++) s += S.M[l * n + i] * HM[l * n + j];
                        Qt[sidx(i, j, n)] = (k >= 1) ? s : 0.0;
                    }
#pragma unroll
                for (int i = 0; i < n; i++)
#pragma unroll
                    for (int j = 0; j < m; j++) {
                        double s = 0;
#pragma unroll
                        for (int l = 0; l < n; l++) s += Qt[sidx(i, l, n)] * S.hb[l * m + j];
                        Qb[i * m + j] = s;
                    }
#pragma unroll
                for (int i = 0; i < m; i++)
#pragma unroll
                    for (int j = i; j < m; j++) {
                        double s = Hu[sidx(i, j, m)];
#pragma unroll
                        for (int l = 0; l < n; l++) s += S.hb[l * m + i] * Qb[l * m + j];
                        Suu[sidx(i, j, m)] = s;
                    }
            }
            // ---- factor stage k (ipm.hpp header; scp_gusto.jl:104 JuMP.optimize!) ----
            // the records of P_k and Pi_k (the value function BEHIND knot k) for the costates of pass D
#pragma unroll
            for (int i = 0; i < n; i++) {
#pragma unroll
                for (int j = i; j < n; j++) W(k, Y::ePP + sidx(i, j, n)) = Pm[i * n + j];
#pragma unroll
                for (int j = 0; j < n; j++) W(k, Y::ePI + i * n + j) = Pi[i * n + j];
            }
            double Tph[n * n], Tga[n * m], Hyy[NHX], Hyu[n * m], Huu[m * m], Zy[n * n], Zu[m * n];
#pragma unroll
            for (int i = 0; i < n; i++) {
#pragma unroll
                for (int j = 0; j < n; j++) {
                    double s = 0;
#pragma unroll
                    for (int l = 0; l < n; l++) s += Pm[i * n + l] * S.Phi[l * n + j];
                    Tph[i * n + j] = s;
                }
#pragma unroll
                for (int j = 0; j < m; j++) {
                    double s = 0;
#pragma unroll
                    for (int l = 0; l < n; l++) s += Pm[i * n + l] * S.Gam[l * m + j];
                    Tga[i * m + j] = s;
                }
            }
#pragma unroll
            for (int i = 0; i < n; i++) {
#pragma unroll
                for (int j = i; j < n; j++) {
                    double s = Qt[sidx(i, j, n)];
#pragma unroll
                    for (int l = 0; l < n; l++) s += S.Phi[l * n + i] * Tph[l * n + j];
                    Hyy[sidx(i, j, n)] = s;
                }
#pragma unroll
                for (int j = 0; j < m; j++) {
                    double s = Qb[i * m + j];
#pragma unroll
                    for (int l = 0; l < n; l++) s += S.Phi[l * n + i] * Tga[l * m + j];
                    Hyu[i * m + j] = s;
                }
#pragma unroll
                for (int j = 0; j < n; j++) {
                    double s = 0;
#pragma unroll
                    for (int l = 0; l < n; l++) s += S.Phi[l * n + i] * Pi[l * n + j];
                    Zy[i * n + j] = s;
                }
            }
#pragma unroll
            for (int i = 0; i < m; i++) {
#pragma unroll
                for (int j = 0; j < m; j++) {
                    const int a = i < j ? i : j, c = i < j ? j : i;
                    double s = Suu[sidx(a, c, m)];
#pragma unroll
                    for (int l = 0; l < n; l++) s += S.Gam[l * m + a] * Tga[l * m + c];
                    Huu[i * m + j] = s;
                }
#pragma unroll
                for (int j = 0; j < n; j++) {
                    double s = 0;
#pragma unroll
                    for (int l = 0; l < n; l++) s += S.Gam[l * m + i] * Pi[l * n + j];
                    Zu[i * n + j] = s;
                }
            }
            if (k == N - 1) {   // + E = [M^T C^T; b^T M^T C^T]: column j for the goal coordinates only
#pragma unroll
                for (int j = 0; j < n; j++) {
                    const double g = is_goal(j) ? 1.0 : 0.0;
#pragma unroll
                    for (int i = 0; i < n; i++) Zy[i * n + j] += g * S.M[j * n + i];
#pragma unroll
                    for (int i = 0; i < m; i++) {
                        double s = 0;
#pragma unroll
                        for (int l = 0; l < n; l++) s += S.hb[l * m + i] * S.M[j * n + l];
                        Zu[i * n + j] += g * s;
                    }
                }
            }
            double Li[m * m], Wm[m * n], Vm[m * n], Kk[m * n], Dk[m * n], Si[NHM];
            if (!chol_inv<m>(Huu, Li)) fail = true;
#pragma unroll
            for (int i = 0; i < m; i++)
#pragma unroll
                for (int j = 0; j < n; j++) {
                    double s = 0, v = 0;
#pragma unroll
                    for (int l = 0; l <= i; l++) { s += Li[i * m + l] * Hyu[j * m + l]; v += Li[i * m + l] * Zu[l * n + j]; }
                    Wm[i * n + j] = s; Vm[i * n + j] = v;
                }
#pragma unroll
            for (int i = 0; i < m; i++) {
#pragma unroll
                for (int j = 0; j < n; j++) {
                    double s = 0, v = 0;
#pragma unroll
                    for (int l = i; l < m; l++) { s += Li[l * m + i] * Wm[l * n + j]; v += Li[l * m + i] * Vm[l * n + j]; }
                    Kk[i * n + j] = s; Dk[i * n + j] = v;
                }
#pragma unroll
                for (int j = i; j < m; j++) {
                    double s = 0;
#pragma unroll
                    for (int l = j; l < m; l++) s += Li[l * m + i] * Li[l * m + j];
                    Si[sidx(i, j, m)] = s;
                }
            }
            // ---- the predictor's backward vector recurrence at knot k (ipm.hpp: backward_sweep) ----
            // right-hand side: gy = Qt rd + M^T gx, lu = gu + b^T gy (+ [Phi Gam]^T (p + P c))
            double ck[n], qrd[n], rv[n], pic[n], gy[n], lu[m];
#pragma unroll
            for (int i = 0; i < n; i++) {
                double c = 0, s = 0;
#pragma unroll
                for (int l = 0; l < n; l++) { c += S.Phi[i * n + l] * rdk[l]; s += Qt[sidx(i, l, n)] * rdk[l]; }
                ck[i] = c; qrd[i] = s;
            }
#pragma unroll
            for (int i = 0; i < n; i++) {
                double r = 0, pc = 0, g = qrd[i];
#pragma unroll
                for (int l = 0; l < n; l++) { r += Pm[i * n + l] * ck[l]; pc += Pi[l * n + i] * ck[l]; g += S.M[l * n + i] * gx0[l]; }
                rv[i] = r; pic[i] = pc; gy[i] = (k >= 1) ? g : 0.0;
            }
            if (k == 0) {
#pragma unroll
                for (int i = 0; i < n; i++) F.gxs[i] = gx0[i];
            }
#pragma unroll
            for (int i = 0; i < m; i++) {
                double s = 2 * wk * q.u[i] + gu0[i];
#pragma unroll
                for (int l = 0; l < n; l++) s += S.hb[l * m + i] * gy[l] + S.Gam[l * m + i] * (pv[l] + rv[l]);
                lu[i] = s;
            }
            double pn[n], d0[m];
#pragma unroll
            for (int i = 0; i < n; i++) {
                double s = gy[i];
#pragma unroll
                for (int l = 0; l < n; l++) s += S.Phi[l * n + i] * (pv[l] + rv[l]);
#pragma unroll
                for (int l = 0; l < m; l++) s -= Kk[l * n + i] * lu[l];
                pn[i] = s;
                double t = pic[i];
#pragma unroll
                for (int l = 0; l < m; l++) t -= Dk[l * n + i] * lu[l];
                th[i] += t;
            }
#pragma unroll
            for (int i = 0; i < m; i++) {
                double s = 0;
#pragma unroll
                for (int l = 0; l < m; l++) s += Si[sidx(i, l, m)] * lu[l];
                d0[i] = s;
            }
            if (k == N - 1) {   // C M rd_{N-1} - rg, rg = goal - x_N
#pragma unroll
                for (int j = 0; j < n; j++) {
                    double g = 0;
#pragma unroll
                    for (int i = 0; i < n; i++) g += S.M[j * n + i] * rdk[i];
                    F.gterm[j] = is_goal(j) ? g - (gval[j] - q.x[j]) : 0.0;
                }
            }
            // what the later passes read
#pragma unroll
            for (int i = 0; i < n; i++) {
                W(k, Y::eRD + i) = rdk[i]; W(k, Y::eQRD + i) = qrd[i]; W(k, Y::eRV + i) = rv[i]; W(k, Y::ePIC + i) = pic[i];
            }
#pragma unroll
            for (int i = 0; i < m; i++) {
                W(k, Y::eD0 + i) = d0[i];
#pragma unroll
                for (int j = 0; j < n; j++) { W(k, Y::eKK + i * n + j) = Kk[i * n + j]; W(k, Y::eDD + i * n + j) = Dk[i * n + j]; }
#pragma unroll
                for (int j = i; j < m; j++) W(k, Y::eSI + sidx(i, j, m)) = Si[sidx(i, j, m)];
            }
            if (ncomp == 0) {   // (no rows at all: the predictor is the Newton step and pass D takes p_k from here)
#pragma unroll
                for (int i = 0; i < n; i++) W(k, Y::ePS + i) = pv[i];
            }
            // P' = Hyy - W^T W, Pi' = Zy - W^T V, Gd += V^T V
#pragma unroll
            for (int i = 0; i < n; i++) {
#pragma unroll
                for (int j = 0; j < n; j++) {
                    double s = Hyy[sidx(i, j, n)], z = Zy[i * n + j], g = 0;
#pragma unroll
                    for (int l = 0; l < m; l++) { s -= Wm[l * n + i] * Wm[l * n + j]; z -= Wm[l * n + i] * Vm[l * n + j]; g += Vm[l * n + i] * Vm[l * n + j]; }
                    Pm[i * n + j] = s; Pi[i * n + j] = z; Gd[i * n + j] += g;
                }
                pv[i] = pn[i];
            }
            // next knot
#pragma unroll
            for (int i = 0; i < n; i++) nu1[i] = q.nu[i];
            if (k == 0) {
#pragma unroll
                for (int i = 0; i < n; i++) nu0[i] = q.nu[i];
            }
            q = qm;
        }
        // Gd^-1 with an identity block on the coordinates without a point goal
        {
            double G[n * n], Li[n * n];
#pragma unroll
            for (int i = 0; i < n; i++)
#pragma unroll
                for (int j = 0; j < n; j++) G[i * n + j] = (i == j && !is_goal(i)) ? 1.0 : Gd[i * n + j];
            if (!chol_inv<n>(G, Li)) fail = true;
#pragma unroll
            for (int i = 0; i < n; i++)
#pragma unroll
                for (int j = 0; j < n; j++) {
                    double s = 0;
#pragma unroll
                    for (int l = 0; l < n; l++) if (l >= i && l >= j) s += Li[l * n + i] * Li[l * n + j];
                    F.Ginv[i * n + j] = s;
                }
        }
        goal_multipliers(F, th);
        F.fail = fail;
        resp = l_resp; resd = l_resd;
        mu = ncomp > 0 ? l_comp / ncomp : 0.0;
        obj = l_obj / kappa;   // JuMP.objective_value: cost + all slacks, in unscaled units
        numax_ = l_numax;
    }
    double numax_;
    GD void goal_multipliers(Fact& F, const double* th) const {
#pragma unroll
        for (int j = 0; j < n; j++) {
            double s = 0;
#pragma unroll
            for (int l = 0; l < n; l++) s += F.Ginv[j * n + l] * (is_goal(l) ? th[l] + F.gterm[l] : 0.0);
            F.mugn[j] = is_goal(j) ? s : 0.0;
        }
    }

    // ---- pass C: the corrector's backward vector recurrence (row sums gA + mu_t gB of pass B) ----
    GD void passC(Fact& F, double mu_t) {
        double pv[n], th[n];
#pragma unroll
        for (int i = 0; i < n; i++) { pv[i] = 0; th[i] = 0; }
        for (int k = N - 1; k >= 0; k--) {
            const double wk = wk_of(k);
            double lc[NC], gx[n], gu[m], qrd[n], rv[n], pic[n], Kk[m * n], Dk[m * n], Si[NHM];
#pragma unroll
            for (int i = 0; i < NC; i++) lc[i] = W(k, Y::eLC + i);
#pragma unroll
            for (int i = 0; i < n; i++) {
                gx[i] = W(k, Y::eGA + i) + mu_t * W(k, Y::eGB + i);
                qrd[i] = W(k, Y::eQRD + i); rv[i] = W(k, Y::eRV + i); pic[i] = W(k, Y::ePIC + i);
            }
#pragma unroll
            for (int i = 0; i < m; i++) {
                gu[i] = 2 * wk * W(k, Y::eUW + i) + W(k, Y::eGA + n + i) + mu_t * W(k, Y::eGB + n + i);
#pragma unroll
                for (int j = 0; j < n; j++) { Kk[i * n + j] = W(k, Y::eKK + i * n + j); Dk[i * n + j] = W(k, Y::eDD + i * n + j); }
#pragma unroll
                for (int j = i; j < m; j++) Si[sidx(i, j, m)] = W(k, Y::eSI + sidx(i, j, m));
            }
            Stage S;
            stage_of(k, lc, S);
            double gy[n], lu[m];
#pragma unroll
            for (int i = 0; i < n; i++) {
                double g = qrd[i];
#pragma unroll
                for (int l = 0; l < n; l++) g += S.M[l * n + i] * gx[l];
                gy[i] = (k >= 1) ? g : 0.0;
                W(k, Y::ePS + i) = pv[i];   // p_k, for the costates of pass D
            }
            if (k == 0) {
#pragma unroll
                for (int i = 0; i < n; i++) F.gxs[i] = gx[i];
            }
#pragma unroll
            for (int i = 0; i < m; i++) {
                double s = gu[i];
#pragma unroll
                for (int l = 0; l < n; l++) s += S.hb[l * m + i] * gy[l] + S.Gam[l * m + i] * (pv[l] + rv[l]);
                lu[i] = s;
            }
            double pn[n];
#pragma unroll
            for (int i = 0; i < n; i++) {
                double s = gy[i];
#pragma unroll
                for (int l = 0; l < n; l++) s += S.Phi[l * n + i] * (pv[l] + rv[l]);
#pragma unroll
                for (int l = 0; l < m; l++) s -= Kk[l * n + i] * lu[l];
                pn[i] = s;
                double t = pic[i];
#pragma unroll
                for (int l = 0; l < m; l++) t -= Dk[l * n + i] * lu[l];
                th[i] += t;
            }
#pragma unroll
            for (int i = 0; i < m; i++) {
                double s = 0;
#pragma unroll
                for (int l = 0; l < m; l++) s += Si[sidx(i, l, m)] * lu[l];
                W(k, Y::eD0 + i) = s;
            }
#pragma unroll
            for (int i = 0; i < n; i++) pv[i] = pn[i];
        }
        goal_multipliers(F, th);
    }

    // ---- passes B / D: forward recurrence of a right-hand side, primal step, costates (the final one), row steps ----
    struct StepRes { double amax, c0, c1, c2; };
    GD StepRes pass_fwd(const Fact& F, int pass, double mu_t, double tau) {
        const bool final_ = pass == 1 || ncomp == 0;
        double dy[n], nun1[n];
#pragma unroll
        for (int i = 0; i < n; i++) { dy[i] = 0; nun1[i] = 0; }
        StepFrac amax;
        double c0 = 0, c1 = 0, c2 = 0;
        for (int k = 0; k < N; k++) {
            double lc[NC], rd[n], Kk[m * n], Dk[m * n], d0[m], xs[n], us[m];
#pragma unroll
            for (int i = 0; i < NC; i++) lc[i] = W(k, Y::eLC + i);
#pragma unroll
            for (int i = 0; i < n; i++) { rd[i] = W(k, Y::eRD + i); xs[i] = W(k, Y::eXW + i); }
#pragma unroll
            for (int i = 0; i < m; i++) {
                d0[i] = W(k, Y::eD0 + i); us[i] = W(k, Y::eUW + i);
#pragma unroll
                for (int j = 0; j < n; j++) { Kk[i * n + j] = W(k, Y::eKK + i * n + j); Dk[i * n + j] = W(k, Y::eDD + i * n + j); }
            }
            const RS_t rs = rs_of(k);
            RowPre<NP> pre;
            pre.load(rs, T::NFIX, T::NFIX + 2 * n, [&](int var) {
                return var == RS_T || var == RS_LAM || var == RS_S || var == RS_LAMB || (pass && (var == RS_KA || var == RS_KB));
            });
            double Pk[NHX], Pik[n * n], ps[n];
            if (final_ && k + 1 < N) {
#pragma unroll
                for (int i = 0; i < NHX; i++) Pk[i] = W(k, Y::ePP + i);
#pragma unroll
                for (int i = 0; i < n * n; i++) Pik[i] = W(k, Y::ePI + i);
#pragma unroll
                for (int i = 0; i < n; i++) ps[i] = W(k, Y::ePS + i);
            }
            Stage S;
            stage_of(k, lc, S);
            double dus[m], dxs[n], dyn[n];
#pragma unroll
            for (int i = 0; i < m; i++) {
                double s = d0[i];
#pragma unroll
                for (int j = 0; j < n; j++) s += Dk[i * n + j] * F.mugn[j] + Kk[i * n + j] * dy[j];
                dus[i] = -s;
            }
            {
                double a[n];
#pragma unroll
                for (int i = 0; i < n; i++) {
                    double s = dy[i] + rd[i];
#pragma unroll
                    for (int l = 0; l < m; l++) s += S.hb[i * m + l] * dus[l];
                    a[i] = s;
                }
#pragma unroll
                for (int i = 0; i < n; i++) {
                    double s = 0, c = 0;
#pragma unroll
                    for (int l = 0; l < n; l++) { s += S.M[i * n + l] * a[l]; c += S.Phi[i * n + l] * (rd[l] + dy[l]); }
#pragma unroll
                    for (int l = 0; l < m; l++) c += S.Gam[i * m + l] * dus[l];
                    dxs[i] = (k >= 1) ? s : 0.0;
                    dyn[i] = c;   // Phi dy + Gam du + c_k, c_k = Phi rd_k (knot 0: Phi = 0, Gam = b_0)
                }
            }
            if (final_) {
#pragma unroll
                for (int i = 0; i < n; i++) W(k, Y::eDX + i) = dxs[i];
#pragma unroll
                for (int i = 0; i < m; i++) W(k, Y::eDU + i) = dus[i];
                if (k + 1 < N) {   // nu_{k+1} = P_k dy_k + p_k + Pi_k mu_g
#pragma unroll
                    for (int i = 0; i < n; i++) {
                        double s = ps[i];
#pragma unroll
                        for (int l = 0; l < n; l++) s += Pk[sidx(i, l, n)] * dyn[l] + Pik[i * n + l] * F.mugn[l];
                        W(k + 1, Y::eNUN + i) = s;
                        if (k == 0) nun1[i] = s;
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < n; i++) dy[i] = dyn[i];
            // row steps of this knot
            double gAx[n], gBx[n], gAu[m], gBu[m];
#pragma unroll
            for (int i = 0; i < n; i++) { gAx[i] = 0; gBx[i] = 0; }
#pragma unroll
            for (int i = 0; i < m; i++) { gAu[i] = 0; gBu[i] = 0; }
            RowCtx<MODEL> ctx;
            make_ctx(k, kappa, omega, Delta, ctx);
            OpStep<NP, RS_t> op{rs, dxs, dus, pass, mu_t, tau, gAx, gAu, gBx, gBu, &pre};
            op.amax = amax; op.c0 = c0; op.c1 = c1; op.c2 = c2;
            visit_rows<MODEL>(ctx, xs, us, op);
            amax = op.amax; c0 = op.c0; c1 = op.c1; c2 = op.c2;
            if (pass == 0) {
#pragma unroll
                for (int i = 0; i < n; i++) { W(k, Y::eGA + i) = gAx[i]; W(k, Y::eGB + i) = gBx[i]; }
#pragma unroll
                for (int i = 0; i < m; i++) { W(k, Y::eGA + n + i) = gAu[i]; W(k, Y::eGB + n + i) = gBu[i]; }
            }
        }
        if (final_) {   // multiplier of x_1 = x_init from the stationarity of x_1: gx_0 + nu_0 + F_0^T nu_1 = 0 (dx_0 = 0)
            double lc[NC], A[n * n];
#pragma unroll
            for (int i = 0; i < NC; i++) lc[i] = W(0, Y::eLC + i);
            LK::A(P.mp, lc, A);
#pragma unroll
            for (int i = 0; i < n; i++) {
                double s = F.gxs[i] + nun1[i];
#pragma unroll
                for (int j = 0; j < n; j++) s += hdt * A[j * n + i] * nun1[j];
                W(0, Y::eNUN + i) = -s;
            }
        }
        return StepRes{amax.value(), c0, c1, c2};
    }

    // ---- one interior point iteration; returns true when the subproblem is finished (status set) ----
    GD bool ipm_iteration() {
        const gusto_ipm_opts& io = P.io;
        Fact F;
        passA(F);
        if (resp <= io.tol && resd <= io.tol * (1 + numax_) && mu <= 0.1 * io.tol) { status = GUSTO_SOLVER_OPTIMAL; return true; }
        const bool acceptable = resp <= io.tol_acc && resd <= io.tol_acc * (1 + numax_) && mu <= io.tol_acc;
        n_acc = acceptable ? n_acc + 1 : 0;
        if (it >= io.max_iter || (io.acc_iter > 0 && n_acc >= io.acc_iter)) {
            if (acceptable) status = GUSTO_SOLVER_ALMOST;
            return true;
        }
        if (!isfinite(resp) || !isfinite(resd) || !isfinite(mu)) return true;
        if (it == 0) mu_start = mu;
        if (mu > IPM_DIVERGED * fmax(1.0, mu_start)) return true;   // (diverging: an infeasible subproblem, common.hpp)
        if (F.fail) return true;
        // predictor (mu_t = 0) and centred corrector share the factorisation
        double alpha, mu_t;
        {
            const StepRes r = pass_fwd(F, 0, 0.0, 1.0);
            alpha = r.amax;
            const double ca = r.c0 + r.amax * (r.c1 + r.amax * r.c2);   // complementarity after the affine step
            const double mu_aff = ncomp > 0 ? ca / ncomp : 0.0;
            const double rr = (mu > 0) ? mu_aff / mu : 0.0;
            double sigma = rr * rr * rr;
            if (io.sigma_max > 0) sigma = fmin(sigma, io.sigma_max);
            mu_t = fmax(sigma * mu, io.mu_floor);
        }
        if (ncomp != 0) {   // (an equality-constrained QP: the predictor already is the Newton step)
            passC(F, mu_t);
            const StepRes r = pass_fwd(F, 1, mu_t, fmax(0.995, 1.0 - mu));
            alpha = r.amax;
        }
#pragma unroll
        for (int i = 0; i < n; i++) mug[i] += alpha * (F.mugn[i] - mug[i]);
        alpha_prev = alpha;   // iterate, costates and row state are advanced by the next pass A
        it++;
        return false;
    }
};

// ---- the GuSTO outer loop (scp_gusto.jl:49-176) as a per-lane state machine: scp.hpp's scp_problem, lane by lane ----
template <int MODEL> __global__ void __launch_bounds__(64, 1) lane_kernel(const KParams P, int lanes_per_wave) {
    using T = MT<MODEL>;
    using Y = LaneLay<MODEL>;
    constexpr int n = T::n, m = T::m;
    const int lane = threadIdx.x, wave = blockIdx.x;
    const int b0 = wave * lanes_per_wave + lane;
    bool alive = false;
    LaneSolver<MODEL> K(P, P.ws, wave, lane, 0);
    const bool hook = P.mode == 1;
    const gusto_scp_params& sp = P.sp;
    int* sti = P.st_i;
    double* std_ = P.st_d;
    size_t hb = 0;
    int iterations = 0, converged = 0, successful = 0, stop = GUSTO_STOP_MAXITER, total_ipm = 0, n_hist = 1, nJ = 0, n_rho = 1, call_cap = 0;
    double Delta = 0, omega = 0, toggle = 0, conv_prev = 0, Jt = 0;
    bool warm = false;
    // 0: at the top of the GuSTO loop, 1: inside the interior point method, 2: stopped, state to be written back,
    // 3: wants the next problem of the batch, 5: retired (the batch has no more)
    int phase = 5;
    // PERSISTENT lanes: the grid is what the GPU keeps resident (one wave per SIMD), a lane that has finished its problem takes
    // the next one of the batch from a counter (KParams::queue[SQ_HEAD_A], preset to the number of problems handed out at
    // launch).  A wave therefore runs as long as the longest SEQUENCE of problems among its lanes -- with many problems per
    // lane that evens out, which is what makes this decomposition the faster one for batches of several hundred thousand.
    auto start_problem = [&](int bb) {
        K.b = bb;
        sti = P.st_i + (size_t)bb * ST_NI;
        std_ = P.st_d + (size_t)bb * SD_ND;
        hb = (size_t)bb * P.hist_cap;
        K.bind_problem();
        K.load_traj();
        iterations = sti[ST_ITER]; converged = sti[ST_CONV]; successful = sti[ST_SUCC]; stop = GUSTO_STOP_MAXITER;
        total_ipm = sti[ST_IPM]; n_hist = sti[ST_NHIST]; nJ = sti[ST_NJ]; n_rho = sti[ST_NRHO];
        call_cap = iterations + P.max_iter;   // scp_gusto.jl:67
        if (!hook) {   // scp_gusto.jl:73-76
            Jt = K.cost_true(Y::eUP);
            const double rho0 = K.trust_region_ratio(Y::eXP, Y::eUP);
            if (nJ < P.hist_cap) { P.J_true[hb + nJ] = Jt; P.J_full[hb + nJ] = Jt; }
            if (n_rho < P.hist_cap) P.rho[hb + n_rho] = rho0;
            nJ++; n_rho++;
        }
        Delta = hook ? P.sub_Delta[bb] : P.Delta[hb + n_hist - 1];
        omega = hook ? P.sub_omega[bb] : P.omega[hb + n_hist - 1];
        toggle = hook ? P.sub_toggle[bb] : Delta / 8 + P.mp.clearance;
        conv_prev = (n_hist >= 1) ? P.conv[hb + n_hist - 1] : 0.0;
        warm = sti[ST_WARM] != 0;   // the previous subproblem ended OPTIMAL: the next one starts centred at mu_warm
        alive = true; phase = 0;
    };
    if (lane < lanes_per_wave && b0 < P.B) start_problem(b0);
    for (;;) {
        if (phase == 3) {   // the next problem of the batch, if there is one
            const int nb = (lane < lanes_per_wave) ? atomicAdd(P.queue + SQ_HEAD_A, 1) : P.B;
            if (nb < P.B) start_problem(nb); else phase = 5;
        }
        if (alive && phase == 0) {
            if (hook || (iterations < call_cap && n_hist < P.hist_cap && nJ < P.hist_cap && n_rho < P.hist_cap)) {
                K.lin_init(Delta, omega, (warm && !hook) ? warm_mu(P.io, conv_prev) : 0.0);   // :95-102
                phase = 1;
            } else {
                // a history vector is full although iterations remain: say so instead of posing as MaxIter
                if (stop == GUSTO_STOP_MAXITER && iterations < call_cap) stop = GUSTO_STOP_HIST_FULL;
                alive = false;
                phase = 2;
            }
        }
        if (phase == 2) {   // the problem stopped: its state goes back to the per-problem arrays
            K.store_traj(P.X + (size_t)K.b * K.N * n, P.U + (size_t)K.b * K.N * m, Y::eXP, Y::eUP);
            sti[ST_ITER] = iterations; sti[ST_CONV] = converged; sti[ST_SUCC] = successful; sti[ST_STOP] = stop;
            sti[ST_IPM] = total_ipm; sti[ST_NHIST] = n_hist; sti[ST_NJ] = nJ; sti[ST_NRHO] = n_rho; sti[ST_WARM] = warm;
            sti[ST_CAP] = call_cap; sti[ST_VISITS] = 1;
            std_[SD_TOGGLE] = toggle;
            phase = 3;
        }
        if (__ballot(alive || phase == 3) == 0) break;
        bool fin = false;
        if (alive) fin = K.ipm_iteration();   // :96-104, one Newton step of it
        if (alive && fin) {
            phase = 0;
            if (hook) {
                K.store_traj(P.sub_X + (size_t)K.b * K.N * n, P.sub_U + (size_t)K.b * K.N * m, Y::eXW, Y::eUW);
                P.sub_obj[K.b] = K.obj; P.sub_status[K.b] = K.status; P.sub_iters[K.b] = K.it;
                for (int i = 0; i < n; i++) std_[SD_DUAL + i] = K.nu0[i] * fmax(1.0, omega);
                alive = false; phase = 3;
            } else {
                warm = K.status == GUSTO_SOLVER_OPTIMAL;
                total_ipm += K.it;
                const int h = n_hist;
                P.solver_status[hb + h] = K.status; P.ipm_it[hb + h] = K.it;
                if (K.status != GUSTO_SOLVER_OPTIMAL && K.status != GUSTO_SOLVER_ALMOST) {   // :106-111
                    stop = GUSTO_STOP_SUBPROBLEM_FAILED;
                    alive = false; phase = 2;
                } else {
                    // convergence_metric (traj_opt.jl:74-85), trust_region_satisfied_gusto (scp_gusto.jl:34-44),
                    // convex_ineq_satisfied_gusto_jump (:316-343: the same rows, raw values of the new trajectory)
                    double max_d2 = -INFINITY, max_x2 = -INFINITY;
                    bool cvx = true;
                    for (int k = 0; k < K.N; k++) {
                        double xs[n], us[m], dn = 0, xn = 0;
#pragma unroll
                        for (int i = 0; i < n; i++) {
                            xs[i] = K.W(k, Y::eXW + i);
                            const double e = xs[i] - K.W(k, Y::eXP + i);
                            dn += e * e; xn += xs[i] * xs[i];
                        }
#pragma unroll
                        for (int i = 0; i < m; i++) us[i] = K.W(k, Y::eUW + i);
                        max_d2 = fmax(max_d2, dn); max_x2 = fmax(max_x2, xn);
                        RowCtx<MODEL> ctx;
                        K.make_ctx(k, 1.0, 1.0, 1.0, ctx);
                        OpCheck op{sp.eps};
                        visit_rows<MODEL>(ctx, xs, us, op);
                        cvx = cvx && op.ok;
                    }
                    const double conv = sqrt(max_d2) / sqrt(max_x2);
                    const int cvx_sat = cvx ? 1 : 0;
                    // the literal `max_val - Delta <= 0` evaluated with the solver's accuracy as slack
                    const int tr_sat = (max_d2 - Delta <= P.io.tr_tol * fmax(1.0, Delta));
                    int accept, status;
                    double Delta_n, omega_n;
                    if (tr_sat) {                                       // :123-141
                        const double rho = K.trust_region_ratio(Y::eXW, Y::eUW);
                        P.rho[hb + n_rho] = rho;
                        n_rho++;
                        if (rho > sp.rho1) {
                            status = GUSTO_SCP_INACCURATE_MODEL; accept = 0; Delta_n = sp.beta_fail * Delta; omega_n = omega;
                        } else {
                            accept = 1;
                            Delta_n = (rho < sp.rho0) ? fmin(sp.beta_succ * Delta, sp.Delta0) : Delta;
                            if (!cvx_sat) { status = GUSTO_SCP_VIOLATES_CONSTRAINTS; omega_n = sp.gamma_fail * omega; }
                            else { status = GUSTO_SCP_OK; omega_n = omega; }
                        }
                    } else {                                            // :142-147
                        status = GUSTO_SCP_TRUST_REGION_VIOLATED; accept = 0; Delta_n = Delta; omega_n = sp.gamma_fail * omega;
                    }
                    if (accept) {                                       // :149-154
                        Jt = K.cost_true(Y::eUW);
                        for (int k = 0; k < K.N; k++) {
#pragma unroll
                            for (int i = 0; i < n; i++) K.W(k, Y::eXP + i) = K.W(k, Y::eXW + i);
#pragma unroll
                            for (int i = 0; i < m; i++) K.W(k, Y::eUP + i) = K.W(k, Y::eUW + i);
                        }
                    }
                    for (int i = 0; i < n; i++) std_[SD_DUAL + i] = K.nu0[i] * fmax(1.0, omega);  // :117 get_dual_jump
                    P.conv[hb + h] = conv; P.J_full[hb + nJ] = K.obj; P.J_true[hb + nJ] = Jt;
                    P.tr_sat[hb + h] = tr_sat; P.cvx_sat[hb + h] = cvx_sat; P.scp_status[hb + h] = status;
                    P.accept[hb + h] = accept; P.Delta[hb + h] = Delta_n; P.omega[hb + h] = omega_n;
                    nJ++;
                    Delta = Delta_n; omega = omega_n;
                    toggle = Delta / 8 + P.mp.clearance;               // :156
                    n_hist = h + 1;
                    iterations++;
                    const double conv_sum = conv_prev + conv;
                    conv_prev = conv;
                    if (omega > sp.omega_max) { stop = GUSTO_STOP_OMEGA_MAX; alive = false; phase = 2; }   // :163-166
                    else if (accept && iterations > 2 && conv_sum <= sp.convergence_threshold) {       // :169-175
                        converged = 1;
                        if (cvx_sat) successful = 1;
                        if (!P.force) { stop = GUSTO_STOP_CONVERGED; alive = false; phase = 2; }
                    }
                }
            }
        }
    }
}

}  // namespace gusto
