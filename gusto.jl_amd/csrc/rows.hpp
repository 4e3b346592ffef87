// rows.hpp -- the stage-local inequality rows of the convex subproblem and the per-row interior point algebra.
//
// The reference registers constraint functions per model in SCPConstraints(SCPP) (freeflyer_se2.jl:338-390,
// dubins_car.jl:184-226, astrobee_se3.jl:322-379, astrobee_se3_manifold.jl:533-608) and JuMP expands them
// into rows (scp_gusto.jl:192-314).  Here a model is a compile-time "row program": visit_rows<MODEL> walks
// the rows of one knot and hands each to an Op functor; every row is either a diagonal quadratic or a linear
// function of x_k or of u_k, so an Op only ever touches a fixed index window [I0, I0+CNT).
#pragma once
#include <type_traits>

#include "models.hpp"

namespace gusto {

#ifndef GUSTO_OBS_BATCH
#define GUSTO_OBS_BATCH 4
#endif
constexpr int OBS_BATCH = GUSTO_OBS_BATCH, FX_OBS = 64;   // obstacle rows are processed in batches (visit_rows, ObsPre)

template <int CNT> struct RowEv {
    double g;        // scaled row value  ghat = mul * raw - off
    double raw;      // the reference's constraint function value
    double gr[CNT];  // gradient of ghat on its window
    double hd[CNT];  // Hessian diagonal of ghat on its window
};

template <int MODEL> struct RowCtx {
    const KParams* P;
    int N, k, nslot;
    double kappa, omega, Delta;
    GPtr<const double> xp;      // linearisation state of this knot
    uint64_t mask;              // active obstacle rows (dist < obstacle_toggle_distance)
    GPtr<const double> obs_nh;  // [n_obs][WS][N]
    GPtr<const double> obs_c0;  // [n_obs][N]
    GPtr<const double> goal_lo;
    GPtr<const double> goal_hi;
    unsigned boxmask;           // coordinates of x_N with BoxGoal rows (Blk::boxmask)
    bool skip_ctl = false;      // the control rows of this knot are another wave's (segw.hpp: four waves per problem)
#ifdef GUSTO_PROFILE
    Prof* pf = nullptr;   // sub-phase stamps of the row passes (PF_R*)
    int pfb = 0;
    GD void tick(int id) const { if (pf) pf->tick(pfb + id); }
#else
    GD void tick(int) const {}
#endif
};

template <int I, int E, class F> GD void static_for(F&& f) {
    if constexpr (I < E) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, E>(f);
    }
}

template <bool ISU, int I0, int CNT, int FX = -1, class Op>
GD void quad_row(Op& op, int slot, int kind, const double* v, const double* a, const double* v0, double c0,
                 double mul, double off) {
    RowEv<CNT> ev;
    double g = c0;
#pragma unroll
    for (int j = 0; j < CNT; j++) {
        const double e = v[I0 + j] - (v0 ? v0[j] : 0.0);
        g += a[j] * e * e;
        ev.gr[j] = mul * (2 * a[j] * e);
        ev.hd[j] = mul * 2 * a[j];
    }
    ev.raw = g;
    ev.g = mul * g - off;
    op.template row<ISU, I0, CNT, FX>(slot, kind, ev);
}
template <bool ISU, int I0, int CNT, int FX = -1, class Op>
GD void lin_row(Op& op, int slot, int kind, const double* v, const double* b, double c0, double mul, double off) {
    RowEv<CNT> ev;
    double g = c0;
#pragma unroll
    for (int j = 0; j < CNT; j++) {
        g += b[j] * v[I0 + j];
        ev.gr[j] = mul * b[j];
        ev.hd[j] = 0.0;
    }
    ev.raw = g;
    ev.g = mul * g - off;
    op.template row<ISU, I0, CNT, FX>(slot, kind, ev);
}

// ncsi_*_obstacle_avoidance_*_convexified (freeflyer_se2.jl:265-288): clearance - (d + nhat.(r - r0)), the rows of the obstacles in
// `mk` (the knot's active set, or one wave's share of it: segw.hpp).
// The rows of one knot are walked in batches of OBS_BATCH: every load of a batch (normal, offset AND the row state
// the Op needs, Op::obs_load) is issued before the first row of the batch is processed.  One row at a time, each
// row's loads wait behind the stores of the row before it and a pass pays one memory round trip per active
// obstacle (3-8 per knot for the freeflyer table, 15-25 in the ISS corner) -- four passes per interior point
// iteration.  Lanes with fewer rows left aim the spare positions at their last row and skip them.
template <int MODEL, class Op> GD void visit_obs_rows(const RowCtx<MODEL>& c, const double* xs, Op& op, uint64_t mk) {
    using T = MT<MODEL>;
    const double kw = c.kappa * c.omega;
    const int slot_obs = T::NFIX;
    while (mk) {
        int oi[OBS_BATCH], oslot[OBS_BATCH];
        bool ov[OBS_BATCH];
        int last = 0;
#pragma unroll
        for (int q = 0; q < OBS_BATCH; q++) {
            ov[q] = mk != 0;
            if (ov[q]) { last = __ffsll((unsigned long long)mk) - 1; mk &= mk - 1; }
            oi[q] = last; oslot[q] = slot_obs + last;
        }
        double ob[OBS_BATCH][T::WS], oc[OBS_BATCH];
#pragma unroll
        for (int q = 0; q < OBS_BATCH; q++) {
#pragma unroll
            for (int j = 0; j < T::WS; j++) ob[q][j] = -(c.obs_nh + (size_t)(oi[q] * T::WS + j) * (size_t)c.N)[c.k];
            oc[q] = (c.obs_c0 + (size_t)oi[q] * (size_t)c.N)[c.k];
        }
        op.obs_load(oslot);
        static_for<0, OBS_BATCH>([&](auto Q) {
            constexpr int q = decltype(Q)::value;
            if (ov[q]) lin_row<false, 0, T::WS, FX_OBS + q>(op, oslot[q], ROW_PEN, xs, ob[q], oc[q], kw, 0.0);
        });
    }
}

// cci_*_accel_bound on k = 1..N-1 only (freeflyer_se2.jl:236-245,380-381): the two control rows of a knot of the freeflyer / astrobee
// models (a function of its own: with four waves per problem a helper wave runs them, segw.hpp)
template <int MODEL, class Op> GD void visit_ctl_rows(const RowCtx<MODEL>& c, const double* us, Op& op) {
    using T = MT<MODEL>;
    constexpr int n = T::n;
    constexpr bool is2 = MODEL == GUSTO_FREEFLYER_SE2, FB = Op::FIX_BATCH;
    const gusto_model_params& mp = c.P->mp;
    const int slot_u = T::NFIX + c.P->n_obs + 2 * n;
    if (c.k < c.N - 1) {
        constexpr int nf = is2 ? 2 : 3, im = is2 ? 2 : 3, nm = is2 ? 1 : 3;
        double af[nf], am[nm];
#pragma unroll
        for (int j = 0; j < nf; j++) af[j] = 1.0 / (mp.mass * mp.mass);
#pragma unroll
        for (int j = 0; j < nm; j++) { const double ji = 1.0 / mp.Jdiag[is2 ? 2 : j]; am[j] = ji * ji; }
        if constexpr (FB) {
            int fs[OBS_BATCH];
#pragma unroll
            for (int q = 0; q < OBS_BATCH; q++) fs[q] = slot_u + (q < 1 ? 0 : 1);
            op.obs_load(fs);
        }
        quad_row<true, 0, nf, (FB ? FX_OBS + 0 : T::NFIX)>(op, slot_u, ROW_HARD, us, af, nullptr, -mp.hard_limit_accel * mp.hard_limit_accel,
                              1.0 / (mp.hard_limit_accel * mp.hard_limit_accel), 0.0);
        quad_row<true, im, nm, (FB ? FX_OBS + 1 : T::NFIX + 1)>(op, slot_u + 1, ROW_HARD, us, am, nullptr, -mp.hard_limit_alpha * mp.hard_limit_alpha,
                               1.0 / (mp.hard_limit_alpha * mp.hard_limit_alpha), 0.0);
    }
}

// ---- the row programs ------------------------------------------------------------------------------
template <int MODEL, class Op> GD void visit_rows(const RowCtx<MODEL>& c, const double* xs, const double* us, Op& op) {
    using T = MT<MODEL>;
    constexpr int n = T::n;
    const gusto_model_params& mp = c.P->mp;
    const double kw = c.kappa * c.omega;
    const int slot_obs = T::NFIX, slot_goal = T::NFIX + c.P->n_obs, slot_u = slot_goal + 2 * n;
    double one[n];
#pragma unroll
    for (int j = 0; j < n; j++) one[j] = 1.0;

    if constexpr (T::NDEF > 0) {
        // TrajOpt subproblem (scp_trajopt.jl:159-279): the same registry treated differently -- state trust region HARD
        // (||x - xp||^2 - s <= 0, :165-173; c.Delta = s, normalised by s), state, obstacle AND control rows L1-penalised with
        // weight mu (:222-235; c.omega = mu), the dynamics as mu |d_kj| on the defect controls (:257-275, the pair +-mu d <= v)
        constexpr bool is2 = MODEL == GUSTO_TO_FREEFLYER_SE2, man = MODEL == GUSTO_TO_ASTROBEE_SE3_MANIFOLD;
        constexpr int nv = is2 ? 2 : 3, iw = is2 ? 5 : (man ? 10 : 9), nw = is2 ? 1 : 3, m0 = T::m - T::NDEF;
        // The row state of up to OBS_BATCH rows is fetched in one batch before they are processed (Op::obs_load, the buffer of the
        // obstacle batches; round 5): one row at a time, each row's loads wait behind the stores of the row before it and a pass
        // pays a memory round trip per row -- 17 + the active obstacles per knot here, 105 k cycles per pass for the freeflyer.
        // Same rows in the same order: bit-identical.
        constexpr bool FB = Op::FIX_BATCH;
#define GUSTO_FXQ(q) (FB ? FX_OBS + (q) : -1)
        auto batch = [&](int s0, int s1, int s2, int s3) {
            if constexpr (FB) { const int fs[OBS_BATCH] = {s0, s1, s2, s3}; op.obs_load(fs); }
        };
        static_assert(OBS_BATCH == 4 || !FB, "batches of four rows");
        if constexpr (!man) {
            batch(0, 1, 2, 2);
            quad_row<false, 0, n, GUSTO_FXQ(0)>(op, 0, ROW_HARD, xs, one, c.xp, -c.Delta, 1.0 / c.Delta, 0.0);
        } else {
            // the manifold model registers no trust region row (astrobee_se3_manifold.jl:601); its convex_state_eq row, the
            // linearised quaternion norm (:308-313), is HARD in TrajOpt (scp_trajopt.jl:200-208): an equality row (ROW_EQ, common.hpp);
            // csi_orientation_sign (:316-319) is penalised like every convex_state_ineq row
            const double* qp = c.xp + 6;
            const double qn = sqrt(qp[0] * qp[0] + qp[1] * qp[1] + qp[2] * qp[2] + qp[3] * qp[3]);
            double bp[4], c0 = qn - 1.0;
#pragma unroll
            for (int j = 0; j < 4; j++) { bp[j] = qp[j] / qn; c0 -= qp[j] * qp[j] / qn; }
            batch(0, 4, 4, 4);
            lin_row<false, 6, 4, GUSTO_FXQ(0)>(op, 0, ROW_EQ, xs, bp, c0, 1.0, 0.0);
            const double m1 = -1.0;
            lin_row<false, 6, 1, GUSTO_FXQ(1)>(op, 4, ROW_PEN, xs, &m1, 0.0, kw, 0.0);
            batch(1, 2, 2, 2);
        }
        quad_row<false, 3, nv, GUSTO_FXQ(man ? 0 : 1)>(op, 1, ROW_PEN, xs, one, nullptr, -mp.hard_limit_vel * mp.hard_limit_vel, kw, 0.0);
        quad_row<false, iw, nw, GUSTO_FXQ(man ? 1 : 2)>(op, 2, ROW_PEN, xs, one, nullptr, -mp.hard_limit_omega * mp.hard_limit_omega, kw, 0.0);
        uint64_t mk = c.mask;
        if constexpr (FB) {
            while (mk) {   // (the obstacle rows in batches, as in the GuSTO branch below)
                int oi[OBS_BATCH], oslot[OBS_BATCH];
                bool ov[OBS_BATCH];
                int last = 0;
#pragma unroll
                for (int q = 0; q < OBS_BATCH; q++) {
                    ov[q] = mk != 0;
                    if (ov[q]) { last = __ffsll((unsigned long long)mk) - 1; mk &= mk - 1; }
                    oi[q] = last; oslot[q] = slot_obs + last;
                }
                double ob[OBS_BATCH][T::WS], oc[OBS_BATCH];
#pragma unroll
                for (int q = 0; q < OBS_BATCH; q++) {
#pragma unroll
                    for (int j = 0; j < T::WS; j++) ob[q][j] = -(c.obs_nh + (size_t)(oi[q] * T::WS + j) * (size_t)c.N)[c.k];
                    oc[q] = (c.obs_c0 + (size_t)oi[q] * (size_t)c.N)[c.k];
                }
                op.obs_load(oslot);
                static_for<0, OBS_BATCH>([&](auto Q) {
                    constexpr int q = decltype(Q)::value;
                    if (ov[q]) lin_row<false, 0, T::WS, FX_OBS + q>(op, oslot[q], ROW_PEN, xs, ob[q], oc[q], kw, 0.0);
                });
            }
        } else
        while (mk) {   // (one at a time)
            const int oi = __ffsll((unsigned long long)mk) - 1;
            mk &= mk - 1;
            double ob[T::WS];
#pragma unroll
            for (int j = 0; j < T::WS; j++) ob[j] = -(c.obs_nh + (size_t)(oi * T::WS + j) * (size_t)c.N)[c.k];
            const double oc = (c.obs_c0 + (size_t)oi * (size_t)c.N)[c.k];
            lin_row<false, 0, T::WS>(op, slot_obs + oi, ROW_PEN, xs, ob, oc, kw, 0.0);
        }
        {
            // the NHU control rows -- the two acceleration bounds (not at the last knot), then mu |d_kj| as the pair +-mu d <= v per
            // defect (the defect of the last knot moves nothing and is driven to zero) -- in batches of four
            constexpr int nf = is2 ? 2 : 3, im = is2 ? 2 : 3, nm = is2 ? 1 : 3;
            double af[nf], am[nm];
#pragma unroll
            for (int j = 0; j < nf; j++) af[j] = 1.0 / (mp.mass * mp.mass);
#pragma unroll
            for (int j = 0; j < nm; j++) { const double ji = 1.0 / mp.Jdiag[is2 ? 2 : j]; am[j] = ji * ji; }
            const double p1 = 1.0, m1 = -1.0;
            static_for<0, (T::NHU + 3) / 4>([&](auto BB) {
                constexpr int b0 = 4 * decltype(BB)::value;
                auto sl = [&](int q) { return slot_u + (b0 + q < T::NHU ? b0 + q : T::NHU - 1); };
                batch(sl(0), sl(1), sl(2), sl(3));
                static_for<0, 4>([&](auto Q) {
                    constexpr int q = decltype(Q)::value, r = b0 + q;
                    if constexpr (r == 0) {
                        if (c.k < c.N - 1) quad_row<true, 0, nf, GUSTO_FXQ(q)>(op, slot_u, ROW_PEN, us, af, nullptr, -mp.hard_limit_accel * mp.hard_limit_accel, kw, 0.0);
                    } else if constexpr (r == 1) {
                        if (c.k < c.N - 1) quad_row<true, im, nm, GUSTO_FXQ(q)>(op, slot_u + 1, ROW_PEN, us, am, nullptr, -mp.hard_limit_alpha * mp.hard_limit_alpha, kw, 0.0);
                    } else if constexpr (r < T::NHU) {
                        constexpr int j = (r - 2) / 2;
                        lin_row<true, m0 + j, 1, GUSTO_FXQ(q)>(op, slot_u + r, ROW_PEN, us, ((r - 2) % 2 == 0) ? &p1 : &m1, 0.0, kw, 0.0);
                    }
                });
            });
        }
#undef GUSTO_FXQ
    } else
    if constexpr (MODEL == GUSTO_FREEFLYER_SE2 || MODEL == GUSTO_ASTROBEE_SE3 || MODEL == GUSTO_ASTROBEE_SE3_MANIFOLD) {
        constexpr bool is2 = MODEL == GUSTO_FREEFLYER_SE2, man = MODEL == GUSTO_ASTROBEE_SE3_MANIFOLD;
        constexpr int nv = is2 ? 2 : 3, iw = is2 ? 5 : (man ? 10 : 9), nw = is2 ? 1 : 3;
        int slot = 0;
        // Ops without a RowPre (the 12/13-state models: it does not fit their registers) borrow the obstacle batch buffer, idle
        // until the obstacle loop, for the first four state rows and again for the two control rows: a pass pays two or three
        // memory round trips for these rows instead of one per row (5 / 7).  Same rows, same order: bit-identical.
        constexpr bool FB = Op::FIX_BATCH;
        constexpr int NB1 = T::NFIX < OBS_BATCH ? T::NFIX : OBS_BATCH;
#define GUSTO_FXB(i, fx) ((FB && (i) < NB1) ? FX_OBS + (i) : (fx))
        if constexpr (FB) {
            int fs[OBS_BATCH];
#pragma unroll
            for (int q = 0; q < OBS_BATCH; q++) fs[q] = q < NB1 ? q : NB1 - 1;
            op.obs_load(fs);
        }
        if constexpr (!man) {  // stri_state_trust_region (freeflyer_se2.jl:323-326): w*||x-xp||^2 - Delta <= s
            quad_row<false, 0, n, GUSTO_FXB(0, 0)>(op, slot++, ROW_PEN_TR, xs, one, c.xp, 0.0, kw, c.kappa * c.Delta);
        } else {
            // cse_quaternion_norm (manifold.jl:308-313), penalised as a +-eps pair (scp_gusto.jl:297-311)
            const double* qp = c.xp + 6;
            const double qn = sqrt(qp[0] * qp[0] + qp[1] * qp[1] + qp[2] * qp[2] + qp[3] * qp[3]);
            double bp[4], bm[4], c0 = qn - 1.0;
#pragma unroll
            for (int j = 0; j < 4; j++) { bp[j] = qp[j] / qn; bm[j] = -bp[j]; c0 -= qp[j] * qp[j] / qn; }
            lin_row<false, 6, 4, GUSTO_FXB(0, 0)>(op, slot++, ROW_HARD_EQ, xs, bm, -c0, kw, c.kappa * c.P->sp.eps);
            lin_row<false, 6, 4, GUSTO_FXB(1, 1)>(op, slot++, ROW_PEN_EQ, xs, bp, c0, kw, c.kappa * c.P->sp.eps);
            const double m1 = -1.0;  // csi_orientation_sign (manifold.jl:316-319)
            lin_row<false, 6, 1, GUSTO_FXB(2, 2)>(op, slot++, ROW_PEN, xs, &m1, 0.0, kw, 0.0);
        }
        // csi_translational_velocity_bound / csi_angular_velocity_bound (freeflyer_se2.jl:225-233)
        quad_row<false, 3, nv, GUSTO_FXB(T::NFIX - 2, T::NFIX - 2)>(op, slot++, ROW_PEN, xs, one, nullptr, -mp.hard_limit_vel * mp.hard_limit_vel, kw, 0.0);
        quad_row<false, iw, nw, GUSTO_FXB(T::NFIX - 1, T::NFIX - 1)>(op, slot++, ROW_PEN, xs, one, nullptr, -mp.hard_limit_omega * mp.hard_limit_omega, kw, 0.0);
        c.tick(1);   // fixed state rows; then the obstacle rows (visit_obs_rows)
        visit_obs_rows<MODEL>(c, xs, op, c.mask);
        c.tick(2);   // obstacle rows
        if (!c.skip_ctl) visit_ctl_rows<MODEL>(c, us, op);
#undef GUSTO_FXB
    } else {  // DubinsCar: csi_max/min_bound_constraints, cci_max/min_bound_constraints (dynamics.jl:56-81)
        static_for<0, n>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const double p1 = 1.0, m1 = -1.0;
            lin_row<false, i, 1, i>(op, i, ROW_PEN, xs, &p1, -mp.x_max[i], kw, 0.0);
            lin_row<false, i, 1, n + i>(op, n + i, ROW_PEN, xs, &m1, mp.x_min[i], kw, 0.0);
        });
        if (c.k < c.N - 1) {
            const double p1 = 1.0, m1 = -1.0;
            lin_row<true, 0, 1, T::NFIX>(op, slot_u, ROW_HARD, us, &p1, -mp.u_max, 1.0 / fabs(mp.u_max), 0.0);
            lin_row<true, 0, 1, T::NFIX + 1>(op, slot_u + 1, ROW_HARD, us, &m1, mp.u_min, 1.0 / fabs(mp.u_min), 0.0);
        }
    }
    // csbci_goal_constraints: BoxGoal rows are hard (dynamics.jl:37-42, scp_gusto.jl:236-245).  The wave-uniform mask comes
    // first: without it the last knot's lane walked 2n dependent loads of the goal bounds in every row pass -- to find, for a
    // point goal, that there is no such row -- while the other 49 lanes waited (12 k of 200 k cycles per KKT solve)
    if (c.boxmask != 0 && c.k == c.N - 1) {
        // Only the last knot's lane has these rows and the other lanes wait for it, so what counts is its number of memory
        // round trips: two coordinates (four rows) at a time, their bounds and the row state the Op needs (Op::obs_load, as
        // for the obstacle rows) fetched in one batch; coordinates without a BoxGoal are skipped on the mask, without a load.
        // Same rows in the same order as one at a time: bit-identical.  (The manifold model's default goal is a BoxGoal on
        // the four quaternion coordinates: 8 round trips per pass -> 2, 13 k -> 4 k cycles of each of the three row passes.)
        static_assert(OBS_BATCH >= 4, "a batch holds the four rows of two coordinates");
        static_for<0, (n + 1) / 2>([&](auto G) {
            constexpr int i0 = 2 * decltype(G)::value, i1 = (i0 + 1 < n) ? i0 + 1 : i0;
            if ((c.boxmask >> i0) & 3u) {
                const double lo0 = c.goal_lo[i0], hi0 = c.goal_hi[i0], lo1 = c.goal_lo[i1], hi1 = c.goal_hi[i1];
                int oslot[OBS_BATCH];
#pragma unroll
                for (int q = 0; q < OBS_BATCH; q++) oslot[q] = slot_goal + 2 * ((q < 2) ? i0 : i1) + (q & 1);
                op.obs_load(oslot);
                const double p1 = 1.0, m1 = -1.0;
                if (lo0 != hi0) {
                    const double hw = (isfinite(hi0) && isfinite(lo0)) ? 0.5 * (hi0 - lo0) : 1.0;
                    const double sc = 1.0 / fmax(1e-3, fmin(1.0, hw));
                    if (isfinite(hi0)) lin_row<false, i0, 1, FX_OBS + 0>(op, oslot[0], ROW_HARD, xs, &p1, -hi0, sc, 0.0);
                    if (isfinite(lo0)) lin_row<false, i0, 1, FX_OBS + 1>(op, oslot[1], ROW_HARD, xs, &m1, lo0, sc, 0.0);
                }
                if (i1 != i0 && lo1 != hi1) {
                    const double hw = (isfinite(hi1) && isfinite(lo1)) ? 0.5 * (hi1 - lo1) : 1.0;
                    const double sc = 1.0 / fmax(1e-3, fmin(1.0, hw));
                    if (isfinite(hi1)) lin_row<false, i1, 1, FX_OBS + 2>(op, oslot[2], ROW_HARD, xs, &p1, -hi1, sc, 0.0);
                    if (isfinite(lo1)) lin_row<false, i1, 1, FX_OBS + 3>(op, oslot[3], ROW_HARD, xs, &m1, lo1, sc, 0.0);
                }
            }
        });
    }
}

// ---- per-row state access --------------------------------------------------------------------------
struct RowState {
    GPtr<double> base;
    int nslot, N, k;
    // uniform (scalar) row base + per-lane knot index: lets the compiler use the SGPR-base + VGPR-offset form of
    // global_load/store instead of materialising (and hoisting, and spilling) one 64-bit VGPR address per row
    GD auto& at(int var, int slot) const { return (base + (size_t)(var * nslot + slot) * (size_t)N)[k]; }
};

// State of the rows every knot has at compile-time positions (the NFIX state rows, then the NHU control rows; template
// id FX of visit_rows), fetched in ONE batch of loads before a row pass: otherwise each row's loads are issued when
// the pass reaches it, behind the stores of the row before, and the pass pays one memory round trip per row.
template <int NP> struct RowPre {
    double v[RS_NVAR][NP > 0 ? NP : 1];
    template <class RST, class F> GD void load(const RST& rs, int nfix, int slot_u, F&& want) {
#pragma unroll
        for (int var = 0; var < RS_NVAR; var++)
#pragma unroll
            for (int i = 0; i < NP; i++)
                if (want(var)) v[var][i] = rs.at(var, i < nfix ? i : slot_u + (i - nfix));
    }
};

// Row state of one batch of obstacle rows (visit_rows), fetched by Op::obs_load before the batch is processed.  Template
// id FX_OBS + q of a row = position q of the current batch.
struct ObsPre {
    double v[RS_NVAR][OBS_BATCH];
    template <class RST, class F> GD void load(const RST& rs, const int* slot, F&& want) {
#pragma unroll
        for (int var = 0; var < RS_NVAR; var++)
            if (want(var)) {
#pragma unroll
                for (int q = 0; q < OBS_BATCH; q++) v[var][q] = rs.at(var, slot[q]);
            }
    }
};

// ---- the Ops ---------------------------------------------------------------------------------------
// Start point.  Cold (muw == 0: the first subproblem of an SCP run, or after a solver failure): slacks just inside (offset 0.01),
// penalised multipliers lam_a = lam_b = 1/2, hard-row multipliers 0.01/t -- tuned on the freeflyer batch.
// Warm (the iterate starts at the optimum of the previous subproblem): every penalised row is put ON the central
// path at muw for its value g at the start point:  s - t = g, t lam_a = s lam_b = muw, lam_a + lam_b = 1
//   <=>  {s, t} = muw + (sqrt(g^2 + 4 muw^2) +- g) / 2;   hard rows get lam = muw / t.
template <class RST = RowState> struct OpInitT {
    static constexpr bool FIX_BATCH = false;
    RST rs;
    double muw;
    int ncomp = 0;
    GD void obs_load(const int*) {}
    template <bool ISU, int I0, int CNT, int FX> GD void row(int slot, int kind, const RowEv<CNT>& ev) {
        if (kind == ROW_EQ) {   // (eta = 0; t, s, lamb are not used by an equality row)
            rs.at(RS_T, slot) = 1.0; rs.at(RS_LAM, slot) = 0.0; rs.at(RS_LAMB, slot) = 0.0; rs.at(RS_S, slot) = 0.0;
        } else if (row_is_hard(kind)) {
            const double t = fmax(-ev.g, 1e-2), mu0 = (muw > 0) ? muw : 0.01;
            rs.at(RS_T, slot) = t; rs.at(RS_LAM, slot) = mu0 / t; rs.at(RS_LAMB, slot) = 0.0; rs.at(RS_S, slot) = 0.0;
            ncomp += 1;
        } else if (muw > 0) {
            const double g = ev.g, ag = fabs(g), rr = sqrt(g * g + 4 * muw * muw);
            const double big = muw + 0.5 * (rr + ag), small = muw + 2 * muw * muw / (rr + ag);
            const double s = (g >= 0) ? big : small, t = (g >= 0) ? small : big;
            rs.at(RS_S, slot) = s; rs.at(RS_T, slot) = t; rs.at(RS_LAM, slot) = muw / t; rs.at(RS_LAMB, slot) = muw / s;
            ncomp += 2;
        } else {
            const double s = fmax(ev.g, 0.0) + 0.01;
            rs.at(RS_S, slot) = s; rs.at(RS_T, slot) = s - ev.g; rs.at(RS_LAM, slot) = 0.5; rs.at(RS_LAMB, slot) = 0.5;
            ncomp += 2;
        }
    }
};

using OpInit = OpInitT<>;

// residuals + condensed Hessian:  H += sigma * grad grad^T + lam * hess,  dual residual += lam * grad.
// The row update of the previous interior point step (t += alpha dt, ...) is folded into this pass, and so is the
// row part of the PREDICTOR right-hand side (the coefficient of a row, the predicted new multiplier at dz = 0, reduces to
// coef = lam + sigma * g for mu_t = ka = kb = 0; the corrector's is accumulated by OpStep), which
// saves the predictor its own pass over the rows.
// LRTR: a row over ALL states (the trust region) is not added to H_x; its dyad sigma * grad grad^T and diagonal come back
// as (trs, trg, trh) and resid_phase applies them to the stage cost in factored form.
// (RST: the accessor of the per-row interior point state -- RowState for the wave-per-problem kernels, whose lane k walks
// [var][slot][k] arrays, LaneRS for the lane-per-problem kernel, lane.hpp; ssum: the slacks of the penalised rows after the
// update, for the kernels that take the objective from this pass)
template <int n, int m, int NP, bool LRTR = false, class RST = RowState> struct OpResidHess {
    static constexpr bool FIX_BATCH = NP == 0;   // (visit_rows: the fixed rows through the obstacle batch buffer)
    RST rs;
    double *Hx, *Hu, *rdx, *rdu, *gx0, *gu0;
    double alpha_prev;  // 0 on the first trip
    const RowPre<NP>* pre;
    double comp = 0, maxrp = 0;
    ObsPre ob;
    double trs = 0, trg[LRTR ? n : 1], trh[LRTR ? n : 1];
    double ssum = 0;
    GD void obs_load(const int* slot) {
        const bool upd = alpha_prev != 0.0;
        ob.load(rs, slot, [&](int var) {
            return var == RS_T || var == RS_LAM || var == RS_S || var == RS_LAMB ||
                   (upd && (var == RS_DT || var == RS_DL || var == RS_DS));
        });
    }
    template <int FX> GD double get(int var, int slot) const {
        if constexpr (FX >= FX_OBS) return ob.v[var][FX - FX_OBS];
        else if constexpr (FX >= 0 && NP > 0) return pre->v[var][FX];
        else return rs.at(var, slot);
    }
    template <bool ISU, int I0, int CNT, int FX> GD void row(int slot, int kind, const RowEv<CNT>& ev) {
        double t = get<FX>(RS_T, slot), lam = get<FX>(RS_LAM, slot);
        double sig, rp;
        if (alpha_prev != 0.0) {
            const double dl = get<FX>(RS_DL, slot);
            t += alpha_prev * get<FX>(RS_DT, slot);
            lam += alpha_prev * dl;
            rs.at(RS_T, slot) = t; rs.at(RS_LAM, slot) = lam;
        }
        if (kind == ROW_EQ) {   // (common.hpp: TRAJOPT_EQ_DELTA; lam = the multiplier eta, t is a constant 1)
            rp = ev.g;
            sig = 1.0 / TRAJOPT_EQ_DELTA;
        } else if (row_is_hard(kind)) {
            rp = ev.g + t;
            comp += t * lam;
            sig = lam * rcp_nr(t);
        } else {
            double s = get<FX>(RS_S, slot), lamb = get<FX>(RS_LAMB, slot);
            if (alpha_prev != 0.0) {
                s += alpha_prev * get<FX>(RS_DS, slot);
                lamb -= alpha_prev * get<FX>(RS_DL, slot);
                rs.at(RS_S, slot) = s; rs.at(RS_LAMB, slot) = lamb;
            }
            rp = ev.g - s + t;
            ssum += s;
            comp += t * lam + s * lamb;
            sig = lam * lamb * rcp_nr(t * lamb + lam * s);   // = lam / (t + lam s / lamb)
        }
        maxrp = nanmax(maxrp, fabs(rp));
        double* H = ISU ? Hu : Hx;
        double* r = ISU ? rdu : rdx;
        double* g0 = ISU ? gu0 : gx0;
        const double coef0 = lam + sig * ev.g;
        constexpr int dim = ISU ? m : n;
        if constexpr (LRTR && !ISU && CNT == n) {
            trs = sig;
#pragma unroll
            for (int a = 0; a < CNT; a++) {
                r[I0 + a] += lam * ev.gr[a];
                g0[I0 + a] += coef0 * ev.gr[a];
                trg[a] = ev.gr[a];
                trh[a] = lam * ev.hd[a];
            }
            return;
        }
#pragma unroll
        for (int a = 0; a < CNT; a++) {
            r[I0 + a] += lam * ev.gr[a];
            g0[I0 + a] += coef0 * ev.gr[a];
#pragma unroll
            for (int b = a; b < CNT; b++) H[sidx(I0 + a, I0 + b, dim)] += sig * ev.gr[a] * ev.gr[b];
            H[sidx(I0 + a, I0 + a, dim)] += lam * ev.hd[a];
        }
    }
};

// fraction-to-boundary ratio test without a division per candidate: the running minimum is kept as a fraction
// an/ad (ad > 0); tau v / (-dv) < an / ad  <=>  tau v ad < an (-dv)
struct StepFrac {
    double an = 1.0, ad = 1.0;
    GD void test(double v, double dv, double tau) {
        const double cn = tau * v, cd = -dv;
        const bool take = (dv < 0) && (cn * ad < an * cd);
        an = take ? cn : an; ad = take ? cd : ad;
    }
    GD double value() const { return an * rcp_nr(ad); }
};

// row steps (dt, dlam, ds) from the primal step and the fraction-to-boundary step length.  In the predictor pass
// the complementarity after a step alpha is accumulated as c0 + c1 alpha + c2 alpha^2 (so no second row pass is
// needed once alpha_aff is known) and the Mehrotra second-order terms are stored.
// The predictor pass also accumulates the row part of the CORRECTOR's right-hand side.  Its coefficient per row (what
// a row pass of its own used to evaluate: coef = predicted multiplier at dz = 0 with the second-order terms ka, kb) is
// affine in the centring parameter, coef = A + mu_t B, and everything A and B need is in registers right here, so the
// pass leaves gA = sum A grad and gB = sum B grad per knot and the corrector forms gA + mu_t gB once mu_t is known: one
// row pass (6 row-state reads per row) less per interior point iteration.
// HDX: the pass also leaves hdx = H_x dx of this knot, H_x = sum over the state rows of sigma grad grad^T + lam hess -- the
// condensed Hessian the residual pass put into the stage cost, applied to the step instead of stored (the adjoint costate
// recursion of the 12/13-state kernels, ipm.hpp: adjoint_sweep_1w, needs H_x dx_k and nothing else of H_x).
template <int NP, class RST = RowState, bool HDX = false> struct OpStep {
    static constexpr bool FIX_BATCH = NP == 0;
    RST rs;
    const double *dxs, *dus;
    int pass;
    double mu_t, tau;
    double *gAx, *gAu, *gBx, *gBu;   // (pass 0) corrector row sums of this knot
    const RowPre<NP>* pre;           // state of the fixed-position rows, fetched in one batch (small models)
    double* hdx = nullptr;           // (HDX) H_x dx of this knot
    StepFrac amax;
    double c0 = 0, c1 = 0, c2 = 0;
    ObsPre ob;
    GD void obs_load(const int* slot) {
        ob.load(rs, slot, [&](int var) {
            return var == RS_T || var == RS_LAM || var == RS_S || var == RS_LAMB || (pass && (var == RS_KA || var == RS_KB));
        });
    }
    template <int FX> GD double get(int var, int slot) const {
        if constexpr (FX >= FX_OBS) return ob.v[var][FX - FX_OBS];
        else if constexpr (FX >= 0 && NP > 0) return pre->v[var][FX];
        else return rs.at(var, slot);
    }
    template <bool ISU, int I0, int CNT, int FX> GD void row(int slot, int kind, const RowEv<CNT>& ev) {
        const double t = get<FX>(RS_T, slot), lam = get<FX>(RS_LAM, slot);
        const double ka = pass ? get<FX>(RS_KA, slot) : 0.0;
        const double* dv = ISU ? dus : dxs;
        double w = 0;
#pragma unroll
        for (int a = 0; a < CNT; a++) w += ev.gr[a] * dv[I0 + a];
        double dt, dl, ds;
        double cA = 0, cB = 0;   // (pass 0) corrector coefficient = cA + mu_t cB
        double sig = 0;          // (HDX) the row's weight in the condensed Hessian, as OpResidHess forms it
        if (kind == ROW_EQ) {    // d_eta = (h + grad' dx) / delta: nothing here has a sign to keep, no step-length test
            sig = 1.0 / TRAJOPT_EQ_DELTA;
            dl = sig * (ev.g + w);
            if constexpr (HDX && !ISU) {
                const double sw = sig * w;
#pragma unroll
                for (int a = 0; a < CNT; a++) hdx[I0 + a] += sw * ev.gr[a];
            }
            if (pass == 0) {     // the corrector's coefficient of an equality row has no second-order term and no mu_t
                rs.at(RS_KA, slot) = 0.0;
                const double cE = lam + sig * ev.g;
                double* gA = ISU ? gAu : gAx;
#pragma unroll
                for (int a = 0; a < CNT; a++) gA[I0 + a] += cE * ev.gr[a];
            }
            if (pass) { rs.at(RS_DT, slot) = 0.0; rs.at(RS_DL, slot) = dl; rs.at(RS_DS, slot) = 0.0; }
            return;
        }
        if (row_is_hard(kind)) {
            const double rp = ev.g + t;
            const double rt = rcp_nr(t);
            dt = -rp - w;
            dl = (mu_t - t * lam - ka - lam * dt) * rt;
            ds = 0.0;
            if (pass == 0) { cA = (lam * rp - dt * dl) * rt; cB = rt; }
            if constexpr (HDX) sig = lam * rt;
        } else {
            const double s = get<FX>(RS_S, slot), lamb = get<FX>(RS_LAMB, slot);
            const double kb = pass ? get<FX>(RS_KB, slot) : 0.0;
            const double rp = ev.g - s + t;
            const double il = rcp_nr(lamb), lol = lam * il;
            const double D = t + lol * s;
            const double rho0 = mu_t - t * lam - ka + lam * rp - lol * (mu_t - s * lamb - kb);
            const double rD = rcp_nr(D);
            dl = (rho0 + lam * w) * rD;
            ds = (mu_t - s * lamb - kb + s * dl) * il;
            dt = -rp - w + ds;
            if constexpr (HDX) sig = lam * rD;   // = lam lamb / (t lamb + lam s)
            amax.test(s, ds, tau);
            amax.test(lamb, -dl, tau);
            if (pass == 0) {
                c0 += s * lamb; c1 += ds * lamb - s * dl; c2 -= ds * dl; rs.at(RS_KB, slot) = -ds * dl;
                // with ka = dt dl, kb = -ds dl:  rho0 = mu_t (1 - lol) - t lam - ka + lam rp + lol (s lamb + kb)
                cA = lam + (lam * rp - t * lam - dt * dl + lol * (s * lamb - ds * dl)) * rD;
                cB = (1.0 - lol) * rD;
            }
        }
        if constexpr (HDX && !ISU) {
            const double sw = sig * w;
#pragma unroll
            for (int a = 0; a < CNT; a++) hdx[I0 + a] += sw * ev.gr[a] + (lam * ev.hd[a]) * dv[I0 + a];
        }
        amax.test(t, dt, tau);
        amax.test(lam, dl, tau);
        if (pass == 0) {
            c0 += t * lam; c1 += dt * lam + t * dl; c2 += dt * dl; rs.at(RS_KA, slot) = dt * dl;
            double* gA = ISU ? gAu : gAx;
            double* gB = ISU ? gBu : gBx;
#pragma unroll
            for (int a = 0; a < CNT; a++) { gA[I0 + a] += cA * ev.gr[a]; gB[I0 + a] += cB * ev.gr[a]; }
        }
        // (the predictor's steps only feed alpha_aff and the second-order terms: the corrector overwrites them, and a
        //  problem without a corrector pass -- ncomp == 0 -- has no rows)
        if (pass) { rs.at(RS_DT, slot) = dt; rs.at(RS_DL, slot) = dl; rs.at(RS_DS, slot) = ds; }
    }
};

template <class RST = RowState> struct OpSlackSumT {
    static constexpr bool FIX_BATCH = false;
    RST rs;
    double sum = 0;
    GD void obs_load(const int*) {}
    template <bool ISU, int I0, int CNT, int FX> GD void row(int slot, int kind, const RowEv<CNT>&) {
        if (!row_is_hard(kind) && kind != ROW_EQ) sum += rs.at(RS_S, slot);   // (the last residual pass already applied every update)
    }
};

using OpSlackSum = OpSlackSumT<>;

// convex_ineq_satisfied_gusto_jump (scp_gusto.jl:316-343): raw row values against eps
struct OpCheck {
    static constexpr bool FIX_BATCH = false;
    double eps;
    bool ok = true;
    GD void obs_load(const int*) {}
    template <bool ISU, int I0, int CNT, int FX> GD void row(int, int kind, const RowEv<CNT>& ev) {
        if (ISU) return;
        if (kind == ROW_PEN && ev.raw >= eps) ok = false;
        if (kind == ROW_PEN_EQ && (ev.raw <= -eps || ev.raw >= eps)) ok = false;
    }
};

}  // namespace gusto
