// scp.hpp -- the GuSTO outer loop (scp_gusto.jl:49-176) as a per-problem state machine on device.
// One workgroup = one problem for the whole solve; problems are independent, so there is no grid-wide
// synchronisation and the hardware dispatcher load-balances problems with different iteration counts.
#pragma once
#include "ipm.hpp"

#ifndef GUSTO_WAVES_PER_EU
#define GUSTO_WAVES_PER_EU 1
#endif

namespace gusto {

// cost_true: trapezoid control effort (freeflyer_se2.jl:66-76)
template <class BLK> GD double cost_true(BLK& K, const double* U) {
    constexpr int m = BLK::m, mc = m - BLK::T::NDEF;   // (TrajOpt: U rows hold (u, d); only u is costed)
    const int k = K.tid;
    double l = 0;
    if (k >= 1 && k < K.N) {
#pragma unroll
        for (int j = 0; j < mc; j++) l += 0.5 * K.dt * (U[(k - 1) * m + j] * U[(k - 1) * m + j] + U[k * m + j] * U[k * m + j]);
    }
    return block_reduce<BLK::ONE>(l, OpSum(), K.misc);
}

// trust_region_ratio_gusto (freeflyer_se2.jl:392-427 etc.); the "linearised" dynamics deliberately lack B*du
template <int MODEL, class BLK> GD double trust_region_ratio(BLK& K, const double* X, const double* U, const double* Xp, const double* Up) {
    using T = MT<MODEL>;
    constexpr int n = T::n, m = T::m;
    const int k = K.tid, N = K.N;
    double num = 0, den = 0;
    if (k < N) {
        double x[n], u[m], xp[n], up[m];
#pragma unroll
        for (int i = 0; i < n; i++) { x[i] = X[k * n + i]; xp[i] = Xp[k * n + i]; }
#pragma unroll
        for (int i = 0; i < m; i++) { u[i] = U[k * m + i]; up[i] = Up[k * m + i]; }
        if (k < N - 1) {
            double f[n], fp[n], A[n * n];
            Dyn<MODEL>::f(K.P.mp, xp, up, fp);
            Dyn<MODEL>::A(K.P.mp, xp, up, A);
            Dyn<MODEL>::f(K.P.mp, x, u, f);
            double a = 0, b = 0;
#pragma unroll
            for (int i = 0; i < n; i++) {
                double lin = fp[i];
#pragma unroll
                for (int j = 0; j < n; j++) lin += A[i * n + j] * (x[j] - xp[j]);
                a += (f[i] - lin) * (f[i] - lin);
                b += lin * lin;
            }
            num += sqrt(a); den += sqrt(b);
        }
        if constexpr (T::HAS_OBS) {
            const Env E = K.env();
            for (int c = 0; c < K.P.mp.n_robot_comp; c++)
                for (int i = 0; i < E.n_obs; i++) {
                    double nh[T::WS], nh1[T::WS];
                    const double d0 = signed_distance<T::WS>(K.P, E, c, xp, i, nh);
                    double lin = K.P.mp.clearance - d0;
#pragma unroll
                    for (int j = 0; j < T::WS; j++) lin -= nh[j] * (x[j] - xp[j]);
                    const double d1 = signed_distance<T::WS>(K.P, E, c, x, i, nh1);
                    num += fabs((K.P.mp.clearance - d1) - lin);
                    den += fabs(lin);
                }
        }
    }
    num = block_reduce<BLK::ONE>(num, OpSum(), K.misc);
    den = block_reduce<BLK::ONE>(den, OpSum(), K.misc);
    return num / den;
}

template <class BLK> GD void store_traj(BLK& K, const double* Xs, const double* Us, double* Xg, double* Ug) {
    constexpr int n = BLK::n, m = BLK::m;
    for (int e = K.tid; e < K.N * n; e += K.nt()) Xg[e] = Xs[e];
    for (int e = K.tid; e < K.N * m; e += K.nt()) Ug[e] = Us[e];
}

// One time slice of one problem: the body of the persistent kernel below.  `cont` = the problem has run before in this
// gusto_solve call (no new leading history entries), `trips` = how many GuSTO trips this slice may take.  Returns the
// penalty level of the problem (number of omega raises so far) if it has to come back for another slice, -1 if it stopped.
template <int MODEL, bool ONEWAVE, int NCH = 0> GD int scp_problem(const KParams& P, double* lds, int b_, int slot, bool cont, int trips) {
    using T = MT<MODEL>;
    constexpr int n = T::n, m = T::m;
    Blk<MODEL, ONEWAVE> K(P, lds, b_, slot);
    Prof pf;
    const int b = K.b, tid = K.tid, N = K.N, k = tid;
#ifdef GUSTO_PROFILE
    if (NCH > 0 && !cont && tid < 3 && P.prof) P.prof[(size_t)b * PROF_N + 29 + tid] = 0;
#endif
    double* Xg = P.X + (size_t)b * N * n;
    double* Ug = P.U + (size_t)b * N * m;

    // mode 1 = one convex subproblem around the stored (Xp,Up) (parity hook): it takes the first half of one trip of
    // the loop below, so the interior point code exists once in the kernel
    const bool hook = P.mode == 1;

    const gusto_scp_params& sp = P.sp;
    int* sti = P.st_i + (size_t)b * ST_NI;
    double* std_ = P.st_d + (size_t)b * SD_ND;
    const size_t hb = (size_t)b * P.hist_cap;
    int iterations = sti[ST_ITER], converged = sti[ST_CONV], successful = sti[ST_SUCC], stop = GUSTO_STOP_MAXITER;
    int total_ipm = sti[ST_IPM], n_hist = sti[ST_NHIST], nJ = sti[ST_NJ], n_rho = sti[ST_NRHO];
    const int call_cap = cont ? sti[ST_CAP] : iterations + P.max_iter;  // scp_gusto.jl:67 (of the whole gusto_solve call)
    const int iter_cap = (trips < call_cap - iterations) ? iterations + trips : call_cap;   // ... of this slice

    // K.Xp / K.Up are the stored trajectory (SCPS.traj) itself
    // scp_gusto.jl:73-76
    double Jt = 0, rho0v = 0;
    if (!hook) {
        Jt = cost_true(K, K.Up);
        rho0v = trust_region_ratio<MODEL>(K, K.Xp, K.Up, K.Xp, K.Up);
    }
    double Delta = hook ? P.sub_Delta[b] : P.Delta[hb + n_hist - 1], omega = hook ? P.sub_omega[b] : P.omega[hb + n_hist - 1];
    if (!cont && !hook) {
        if (tid == 0 && nJ < P.hist_cap) { P.J_true[hb + nJ] = Jt; P.J_full[hb + nJ] = Jt; }
        if (tid == 0 && n_rho < P.hist_cap) P.rho[hb + n_rho] = rho0v;
        nJ++; n_rho++;
    }
    double toggle = hook ? P.sub_toggle[b] : Delta / 8 + P.mp.clearance;
    double conv_prev = (n_hist >= 1) ? P.conv[hb + n_hist - 1] : 0.0;

    bool warm = sti[ST_WARM] != 0;  // the previous subproblem ended OPTIMAL: the next one starts centred at mu_warm
    while (hook || (iterations < iter_cap && n_hist < P.hist_cap && nJ < P.hist_cap && n_rho < P.hist_cap)) {
        pf.tick(PF_SCP);
        linearize<MODEL>(K, toggle);                       // :95  update_model_params!
        pf.tick(PF_LIN);
        IpmOut io;
        ipm_solve<MODEL, Blk<MODEL, ONEWAVE>, NCH>(K, Delta, omega, (warm && !hook) ? warm_mu(P.io, conv_prev) : 0.0, io, pf);  // :96-104
        if (hook) {
            pf.flush(P.prof, b, cont);
            store_traj(K, K.Xw, K.Uw, P.sub_X + (size_t)b * N * n, P.sub_U + (size_t)b * N * m);
            if (tid == 0) {
                P.sub_obj[b] = io.obj; P.sub_status[b] = io.status; P.sub_iters[b] = io.iters;
                for (int i = 0; i < n; i++) P.st_d[(size_t)b * SD_ND + SD_DUAL + i] = K.nu[i] * fmax(1.0, omega);
            }
            return -1;
        }
        // (profile builds, slots 40..44: objective after the loop | row check | reductions | rho | accept copy; the matrix-core
        // models use those slots for their factor stage and lump the trip into slot 47)
        auto tk = [&](int i_) { pf.tick(MT<MODEL>::MFMA ? 47 : 40 + i_); };
        tk(0);
        warm = io.status == GUSTO_SOLVER_OPTIMAL;
        total_ipm += io.iters;
        const int h = n_hist;
        if (tid == 0) { P.solver_status[hb + h] = io.status; P.ipm_it[hb + h] = io.iters; }
        if (io.status != GUSTO_SOLVER_OPTIMAL && io.status != GUSTO_SOLVER_ALMOST) {  // :106-111
            stop = GUSTO_STOP_SUBPROBLEM_FAILED;
            break;
        }
        // convergence_metric (traj_opt.jl:74-85) and trust_region_satisfied_gusto (scp_gusto.jl:34-44)
        double dn = 0, xn = 0;
        bool cvx_l = true;
        if (k < N) {
            double xs[n], us[m];
#pragma unroll
            for (int i = 0; i < n; i++) {
                xs[i] = K.Xw[k * n + i];
                const double e = xs[i] - K.Xp[k * n + i];
                dn += e * e; xn += xs[i] * xs[i];
            }
#pragma unroll
            for (int i = 0; i < m; i++) us[i] = K.Uw[k * m + i];
            // convex_ineq_satisfied_gusto_jump (:316-343): same rows, raw values of the new trajectory
            RowCtx<MODEL> ctx;
            ctx.P = &P; ctx.N = N; ctx.k = k; ctx.nslot = P.wl.nslot; ctx.kappa = 1.0; ctx.omega = 1.0; ctx.Delta = 1.0;
            ctx.xp = K.Xp + k * n; ctx.mask = K.obs_mask[k]; ctx.obs_nh = K.obs_nh; ctx.obs_c0 = K.obs_c0;
            ctx.goal_lo = K.goal_lo; ctx.goal_hi = K.goal_hi; ctx.boxmask = K.boxmask;
            OpCheck op{sp.eps};
            visit_rows<MODEL>(ctx, xs, us, op);
            cvx_l = op.ok;
        }
        tk(1);
        const double max_d2 = block_reduce<ONEWAVE>(dn, OpMax(), K.misc);
        const double max_x2 = block_reduce<ONEWAVE>(xn, OpMax(), K.misc);
        const double conv = sqrt(max_d2) / sqrt(max_x2);
        const int cvx_sat = block_reduce<ONEWAVE>(cvx_l ? 0.0 : 1.0, OpMax(), K.misc) == 0.0;
        tk(2);
        // the literal `max_val - Delta <= 0` evaluated with the solver's accuracy as slack
        const int tr_sat = (max_d2 - Delta <= P.io.tr_tol * fmax(1.0, Delta));
        int accept, status;
        double Delta_n, omega_n;
        if (tr_sat) {                                       // :123-141
            const double rho = trust_region_ratio<MODEL>(K, K.Xw, K.Uw, K.Xp, K.Up);
            tk(3);
            if (tid == 0) P.rho[hb + n_rho] = rho;
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
            Jt = cost_true(K, K.Uw);
            K.sync();
            for (int e = tid; e < N * n; e += K.nt()) K.Xp[e] = K.Xw[e];
            for (int e = tid; e < N * m; e += K.nt()) K.Up[e] = K.Uw[e];
            K.sync();
            tk(4);
        }
        if (tid == 0)
            for (int i = 0; i < n; i++) std_[SD_DUAL + i] = K.nu[i] * fmax(1.0, omega);  // :117 get_dual_jump
        if (tid == 0) {
            P.conv[hb + h] = conv; P.J_full[hb + nJ] = io.obj; P.J_true[hb + nJ] = Jt;
            P.tr_sat[hb + h] = tr_sat; P.cvx_sat[hb + h] = cvx_sat; P.scp_status[hb + h] = status;
            P.accept[hb + h] = accept; P.Delta[hb + h] = Delta_n; P.omega[hb + h] = omega_n;
        }
        nJ++;
        Delta = Delta_n; omega = omega_n;
        toggle = Delta / 8 + P.mp.clearance;               // :156
        n_hist = h + 1;
        iterations++;
        const double conv_sum = conv_prev + conv;
        conv_prev = conv;
        if (omega > sp.omega_max) { stop = GUSTO_STOP_OMEGA_MAX; break; }   // :163-166
        if (!accept) continue;
        if (iterations > 2 && conv_sum <= sp.convergence_threshold) {       // :169-175
            converged = 1;
            if (cvx_sat) successful = 1;
            if (!P.force) { stop = GUSTO_STOP_CONVERGED; break; }
        }
    }
    // more trips of this call remain: the problem goes back to the scheduler with its penalty level
    const bool again = stop == GUSTO_STOP_MAXITER && iterations == iter_cap && iterations < call_cap &&
                       n_hist < P.hist_cap && nJ < P.hist_cap && n_rho < P.hist_cap;
    // a history vector is full although iterations remain: say so instead of posing as MaxIter
    if (stop == GUSTO_STOP_MAXITER && !again && iterations < call_cap) stop = GUSTO_STOP_HIST_FULL;
    pf.tick(PF_SCP);
    pf.flush(P.prof, b, cont);
    if (tid == 0) {
        sti[ST_ITER] = iterations; sti[ST_CONV] = converged; sti[ST_SUCC] = successful; sti[ST_STOP] = stop;
        sti[ST_IPM] = total_ipm; sti[ST_NHIST] = n_hist; sti[ST_NJ] = nJ; sti[ST_NRHO] = n_rho; sti[ST_WARM] = warm;
        sti[ST_CAP] = call_cap; sti[ST_VISITS] = (cont ? sti[ST_VISITS] : 0) + 1;
        std_[SD_TOGGLE] = toggle;
    }
    if (!again) return -1;
    int lvl = 0;   // the long problems are almost exactly those whose penalty weight was raised early (:137-147)
    for (double x = 1.5 * sp.omega0; x < omega && lvl < SCHED_LEVELS - 1; x *= sp.gamma_fail) lvl++;
    return lvl;
}

// ---- scheduler of the persistent kernel -------------------------------------------------------------------------------
// PERSISTENT workgroups (the grid is the number of resident slots, launch.hpp) pull work until every problem of the
// batch has stopped.  Problem lengths vary 5...30 trips and are not known in advance; with first-come-first-served a long
// problem that starts late leaves the GPU empty at the end (makespan 674 "KKT units" against 374 balanced and 415 for the
// longest problem, freeflyer B = 4096).  So a problem's first `probe_visits` slices are ONE trip each, after which it goes
// into the list of its penalty level (omega raises so far: the long problems are those raised early); a workgroup takes
// the highest level >= 1 waiting, else a fresh problem (handed out hardest first, sched_key_kernel), else level 0; from slice
// probe_visits on a problem of level >= 1 runs to its end and one of level 0 goes on in slices of `slice_q` trips (0: to its
// end).  Slicing is free: between trips a problem's whole state lives in HBM (trajectory, histories, st_i), so results
// are bit-identical to an unsliced run.  SQ_PROBING counts the problems that may still be pushed: it drops when a problem
// starts its last slice or stops inside a finite one, and a workgroup that finds nothing to take retires once it is zero.
//   hand-off of a problem between workgroups (possibly on different XCDs): the producer writes the state, releases at
//   agent scope, then publishes the list entry; the consumer claims an index, spins until the entry is there, acquires.
constexpr int SCHED_SPIN_LIMIT = 1 << 20;   // (seconds of sleeping polls: a scheduler bug must not hang the GPU)
// wave-uniform primitives: every lane of the wave executes them, the result is the same scalar in every lane
GD int uload(const int* p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
GD int uload_acq(const int* p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT)); }
GD int uadd(int* p, int v) {
    int r = 0;
    if ((threadIdx.x & 63) == 0) r = atomicAdd(p, v);
    return __builtin_amdgcn_readfirstlane(r);
}
GD int ucas(int* p, int expected, int desired) {   // returns the value found (== expected: the swap happened)
    int r = expected;
    if ((threadIdx.x & 63) == 0)
        __hip_atomic_compare_exchange_strong(p, &r, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return __builtin_amdgcn_readfirstlane(r);
}
// Next piece of work for this workgroup: a fresh problem (cont = false), else the entry (slices so far << 24 | problem)
// at the head of the highest non-empty level list, else -1 when nothing can come any more.  Executed by a WHOLE wave
// with uniform control flow: a polling loop with exits under `if (lane == 0)` does not survive the compiler's
// structurizer when the other lanes stay in the outer loop (the wave never reconverges; found the hard way).
GD int sched_pop(const KParams& P, bool& cont, int& from) {   // from: the level of the list the entry came from
    int* Q = P.queue;
    // claims the head of the highest non-empty list of levels hi .. lo: its entry, -2 if they are empty, -1 if an entry is lost
    auto take = [&](int hi, int lo) -> int {
        for (int L = hi; L >= lo; L--) {
            int h = uload(Q + SQ_HEAD + L * SQ_STRIDE);
            while (h < uload(Q + SQ_TAIL + L * SQ_STRIDE)) {
                const int found = ucas(Q + SQ_HEAD + L * SQ_STRIDE, h, h + 1);
                if (found == h) {     // index h is ours; its entry follows the tail increment that made it visible
                    int e = -1;
                    for (int w = 0; w < SCHED_SPIN_LIMIT && (e = uload(P.lists + (size_t)L * P.list_cap + h)) < 0; w++)
                        __builtin_amdgcn_s_sleep(2);
                    if (e < 0) {   // a claimed index whose entry never came: the problem is lost -- say so (gusto_finish)
                        if ((threadIdx.x & 63) == 0) atomicExch(Q + SQ_ERR, 1);
                        return -1;
                    }
                    if (L >= 1) uadd(Q + SQ_HI, -1);
                    from = L;
                    return e;
                }
                h = found;
            }
        }
        return -2;
    };
    for (int spin = 0;; spin++) {
        // BEFORE the scan, with acquire: a push is published (tail increment, release) before SQ_PROBING drops
        const int probing_seen = uload_acq(Q + SQ_PROBING);
        // A problem whose penalty weight has been raised (level >= 1: the long ones) goes ahead of the fresh problems.
        // SQ_HI is only a hint that such an entry may be waiting (one load instead of a scan of every list per fresh
        // problem); whatever it misses is found by the full scan below -- at the latest by the workgroup that pushed it.
        if (uload(Q + SQ_HI) > 0) {
            const int e = take(SCHED_LEVELS - 1, 1);
            if (e != -2) { cont = true; return e; }
        }
        if (uload(Q + SQ_HEAD_A) < P.n_fresh) {
            const int q = uadd(Q + SQ_HEAD_A, 1);
            if (q < P.n_fresh) { cont = false; return P.order ? uload(P.order + q) : q; }
        }
        {
            const int e = take(SCHED_LEVELS - 1, 0);
            if (e != -2) { cont = true; return e; }
        }
        // Nothing to take.  More can only come from problems still in their probing slices; once there are none, this
        // workgroup retires and frees its slot -- the tail of a batch then overlaps the head of the next one enqueued on
        // another stream.
        if (probing_seen == 0) return -1;
        if (spin > SCHED_SPIN_LIMIT) {   // problems still probing but nothing arrives: a scheduler bug must not hang the GPU, nor pass
            if ((threadIdx.x & 63) == 0) atomicExch(Q + SQ_ERR, 2);
            return -1;
        }
        // idle polling backs off (8 -> 127 x 64 cycles): pollers share the L2 lines the working groups' pushes need
        if (spin < 4) __builtin_amdgcn_s_sleep(8); else if (spin < 16) __builtin_amdgcn_s_sleep(32); else __builtin_amdgcn_s_sleep(127);
    }
}

template <int MODEL, bool ONEWAVE> __global__ void __launch_bounds__(ONEWAVE ? 64 : 256, ONEWAVE ? MT<MODEL>::WAVES_PER_EU : 1)
scp_kernel(const KParams P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
#ifdef GUSTO_DEBUG_LDS
    if (threadIdx.x == 0) gusto_dbg_lds_limit() = __builtin_amdgcn_groupstaticsize() + (unsigned)P.ll.total * 8u;
    __syncthreads();
#endif
#define GUSTO_BODY_NCH 0
#define GUSTO_BODY_EXIT return
#include "scp_body.inc"
#undef GUSTO_BODY_NCH
#undef GUSTO_BODY_EXIT
}

#if GUSTO_SEG_W2
// A WAVE PER CHAIN of the segmented KKT solve (segw.hpp), for batches that leave SIMDs idle: NCH = 2 or 4 waves per problem.  Wave 0
// is the one-wave kernel above, unchanged but for the sequential phases, where it takes the last chain of the horizon; the other waves
// sleep at a barrier between those phases and take a chain each.
template <int MODEL, int NCH> __global__ void __launch_bounds__(64 * NCH, 1) scp_kernel_w2(const KParams P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
#ifdef GUSTO_DEBUG_LDS
    if (threadIdx.x == 0) gusto_dbg_lds_limit() = __builtin_amdgcn_groupstaticsize() + (unsigned)P.ll.total * 8u;
    __syncthreads();
#endif
    if constexpr (seg2_big<MODEL>()) {
        if (threadIdx.x >= 64) { segw_helper<MODEL, NCH>(P, lds); return; }
        constexpr bool ONEWAVE = true;
#define GUSTO_BODY_NCH NCH
#define GUSTO_BODY_EXIT segw_exit(lds, P.ll.seg + SegB<MODEL, NCH>::MBX); return
#include "scp_body.inc"
#undef GUSTO_BODY_NCH
#undef GUSTO_BODY_EXIT
    }
}
#endif


// ---- TrajOpt (src/scp/scp_trajopt.jl) -----------------------------------------------------------------------------------
// convergence_metric (traj_opt.jl:74-85) = evaluate_xtol (scp_trajopt.jl:281-283)
template <class BLK> GD double convergence_metric_blk(BLK& K, const double* X, const double* Xq) {
    constexpr int n = BLK::n;
    const int k = K.tid;
    double dn = 0, xn = 0;
    if (k < K.N) {
#pragma unroll
        for (int i = 0; i < n; i++) { const double e = X[k * n + i] - Xq[k * n + i]; dn += e * e; xn += X[k * n + i] * X[k * n + i]; }
    }
    const double a = block_reduce<BLK::ONE>(dn, OpMax(), K.misc), b = block_reduce<BLK::ONE>(xn, OpMax(), K.misc);
    return sqrt(a) / sqrt(b);
}
// trust_region_ratio_trajopt (freeflyer_se2.jl:429-467, astrobee_se3.jl:419-460) with the index typos of its dynamics terms
// read as meant -- the forward difference (X[:,k+1]-X[:,k])/dt and the linearised trapezoid defect of interval k -- and the
// obstacle terms linearised at traj_prev as in trust_region_ratio_gusto (the oracle's go_trajopt_ratio)
template <int MODEL, class BLK> GD double trajopt_ratio(BLK& K, const double* X, const double* U, const double* Xq, const double* Uq) {
    using T = MT<MODEL>;
    constexpr int n = T::n, m = T::m, m0 = m - T::NDEF;
    const int k = K.tid, N = K.N;
    double num = 0, den = 0;
    if (k < N - 1) {
        double x0[n], x1[n], q0[n], q1[n], u0[m], u1[m], v0[m], v1[m], f[n], fp[n], fp1[n], A[n * n], A1[n * n], B[n * m];
#pragma unroll
        for (int i = 0; i < n; i++) { x0[i] = X[k * n + i]; x1[i] = X[(k + 1) * n + i]; q0[i] = Xq[k * n + i]; q1[i] = Xq[(k + 1) * n + i]; }
#pragma unroll
        for (int i = 0; i < m; i++) { u0[i] = U[k * m + i]; u1[i] = U[(k + 1) * m + i]; v0[i] = Uq[k * m + i]; v1[i] = Uq[(k + 1) * m + i]; }
        Dyn<MODEL>::f(K.P.mp, q0, v0, fp); Dyn<MODEL>::A(K.P.mp, q0, v0, A);
        Dyn<MODEL>::f(K.P.mp, q1, v1, fp1); Dyn<MODEL>::A(K.P.mp, q1, v1, A1);
        Dyn<MODEL>::f(K.P.mp, x0, u0, f); Dyn<MODEL>::B(K.P.mp, B);
        double po = 0, pn = 0, ph = 0;
#pragma unroll
        for (int i = 0; i < n; i++) {
            double a0 = fp[i], a1 = fp1[i];
#pragma unroll
            for (int j = 0; j < n; j++) { a0 += A[i * n + j] * (x0[j] - q0[j]); a1 += A1[i * n + j] * (x1[j] - q1[j]); }
#pragma unroll
            for (int j = 0; j < m0; j++) { a0 += B[i * m + j] * (u0[j] - v0[j]); a1 += B[i * m + j] * (u1[j] - v1[j]); }
            po += fabs(fp[i] - (q1[i] - q0[i]) / K.dt);
            pn += fabs(f[i] - (x1[i] - x0[i]) / K.dt);
            ph += fabs(x1[i] - x0[i] - 0.5 * K.dt * (a0 + a1));
        }
        num += po - pn; den += po - ph;
    }
    if (k < N) {
        double xw[T::WS], qw[T::WS];
#pragma unroll
        for (int j = 0; j < T::WS; j++) { xw[j] = X[k * n + j]; qw[j] = Xq[k * n + j]; }
        const Env E = K.env();
        for (int c = 0; c < K.P.mp.n_robot_comp; c++)
            for (int i = 0; i < E.n_obs; i++) {
                double nh[T::WS], nh1[T::WS];
                const double d0 = signed_distance<T::WS>(K.P, E, c, qw, i, nh), d1 = signed_distance<T::WS>(K.P, E, c, xw, i, nh1);
                double lin = d0;
#pragma unroll
                for (int j = 0; j < T::WS; j++) lin += nh[j] * (xw[j] - qw[j]);
                const double cl = K.P.mp.clearance;
                num += (cl - d0) - (cl - d1); den += (cl - d0) - (cl - lin);
            }
    }
    num = block_reduce<BLK::ONE>(num, OpSum(), K.misc);
    den = block_reduce<BLK::ONE>(den, OpSum(), K.misc);
    return num / den;
}
// evaluate_ctol (scp_trajopt.jl:289-312): per class of constraints the largest change and the largest value over its members,
// summed over the classes (every class counts: as written the class entered last is dropped)
template <int MODEL, class BLK> GD double trajopt_ctol(BLK& K, const double* X, const double* U, const double* Xq, const double* Uq) {
    using T = MT<MODEL>;
    constexpr int n = T::n, m = T::m;
    constexpr bool is2 = MODEL == GUSTO_TO_FREEFLYER_SE2, man = MODEL == GUSTO_TO_ASTROBEE_SE3_MANIFOLD;
    constexpr int nv = is2 ? 2 : 3, iw = is2 ? 5 : (man ? 10 : 9), nw = is2 ? 1 : 3;
    const int k = K.tid, N = K.N;
    const gusto_model_params& mp = K.P.mp;
    double JN = 0, JD = 0;
    auto cls = [&](double a, double b) {
        JN += block_reduce<BLK::ONE>(a, OpMax(), K.misc); JD += block_reduce<BLK::ONE>(b, OpMax(), K.misc);
    };
    double x[n], q[n];
#pragma unroll
    for (int i = 0; i < n; i++) { x[i] = (k < N) ? X[k * n + i] : 0.0; q[i] = (k < N) ? Xq[k * n + i] : 0.0; }
    if constexpr (man) {   // csi_orientation_sign (-qw) and cse_quaternion_norm(traj, traj) = |q_k| - 1 (manifold.jl:308-319)
        cls(k < N ? fabs(-x[6] + q[6]) : 0.0, k < N ? fabs(-x[6]) : 0.0);
        double a = 0, b = 0;
#pragma unroll
        for (int j = 0; j < 4; j++) { a += x[6 + j] * x[6 + j]; b += q[6 + j] * q[6 + j]; }
        const double g = sqrt(a) - 1.0, gq = sqrt(b) - 1.0;
        cls(k < N ? fabs(g - gq) : 0.0, k < N ? fabs(g) : 0.0);
    }
    {   // csi_translational_velocity_bound, csi_angular_velocity_bound
        double g = -mp.hard_limit_vel * mp.hard_limit_vel, gq = g;
#pragma unroll
        for (int j = 0; j < nv; j++) { g += x[3 + j] * x[3 + j]; gq += q[3 + j] * q[3 + j]; }
        cls(k < N ? fabs(g - gq) : 0.0, k < N ? fabs(g) : 0.0);
        g = -mp.hard_limit_omega * mp.hard_limit_omega; gq = g;
#pragma unroll
        for (int j = 0; j < nw; j++) { g += x[iw + j] * x[iw + j]; gq += q[iw + j] * q[iw + j]; }
        cls(k < N ? fabs(g - gq) : 0.0, k < N ? fabs(g) : 0.0);
    }
    const Env E = K.env();
    if (E.n_obs > 0) {   // ncsi_body_obstacle_avoidance_constraints: clearance - dist
        double a = 0, b = 0;
        if (k < N) {
            double xw[T::WS], qw[T::WS], nh[T::WS];
#pragma unroll
            for (int j = 0; j < T::WS; j++) { xw[j] = x[j]; qw[j] = q[j]; }
            for (int i = 0; i < E.n_obs; i++) {
                const double g = mp.clearance - signed_distance<T::WS>(K.P, E, 0, xw, i, nh);
                const double gq = mp.clearance - signed_distance<T::WS>(K.P, E, 0, qw, i, nh);
                a = fmax(a, fabs(g - gq)); b = fmax(b, fabs(g));
            }
        }
        cls(a, b);
    }
    {   // csbci_goal_constraints (BoxGoal rows, :array): wave-uniform, every lane forms it from the last knot
        double a = 0, b = 0;
        bool any = false;
#pragma unroll
        for (int i = 0; i < n; i++) {
            const double lo = K.goal_lo[i], hi = K.goal_hi[i];
            if (lo == hi) continue;
            const double xv = X[(N - 1) * n + i], xq = Xq[(N - 1) * n + i];
            if (isfinite(hi)) { a += (xv - xq) * (xv - xq); b += (xv - hi) * (xv - hi); any = true; }
            if (isfinite(lo)) { a += (xv - xq) * (xv - xq); b += (lo - xv) * (lo - xv); any = true; }
        }
        if (any) { JN += sqrt(a); JD += sqrt(b); }
    }
    {   // dynamics_constraints(traj, traj, k): the trapezoid defect of the trajectory itself
        double a = 0, b = 0;
        if (k < N - 1) {
            double x1[n], q1[n], u0[m], u1[m], v0[m], v1[m], f0[n], f1[n], g0[n], g1[n];
#pragma unroll
            for (int i = 0; i < n; i++) { x1[i] = X[(k + 1) * n + i]; q1[i] = Xq[(k + 1) * n + i]; }
#pragma unroll
            for (int i = 0; i < m; i++) { u0[i] = U[k * m + i]; u1[i] = U[(k + 1) * m + i]; v0[i] = Uq[k * m + i]; v1[i] = Uq[(k + 1) * m + i]; }
            Dyn<MODEL>::f(mp, x, u0, f0); Dyn<MODEL>::f(mp, x1, u1, f1);
            Dyn<MODEL>::f(mp, q, v0, g0); Dyn<MODEL>::f(mp, q1, v1, g1);
#pragma unroll
            for (int i = 0; i < n; i++) {
                const double F = x1[i] - x[i] - 0.5 * K.dt * (f0[i] + f1[i]), Fq = q1[i] - q[i] - 0.5 * K.dt * (g0[i] + g1[i]);
                a += (F - Fq) * (F - Fq); b += F * F;
            }
        }
        cls(sqrt(a), sqrt(b));
    }
    return JN / JD;
}

// solve_trajopt_jump! (scp_trajopt.jl:33-157) of one problem: the oracle's go_solve_trajopt, statement for statement
template <int MODEL> GD void trajopt_problem(const KParams& P, double* lds, int b_, int slot) {
    using T = MT<MODEL>;
    constexpr int n = T::n, m = T::m;
    Blk<MODEL, false> K(P, lds, b_, slot);
    Prof pf;
    const int b = K.b, tid = K.tid, N = K.N;
    const gusto_trajopt_params& tp = P.tp;
    int* sti = P.st_i + (size_t)b * ST_NI;
    const size_t hb = (size_t)b * P.hist_cap;
    const double toggle = P.mp.clearance + 1.0;                                   // :64
    if (P.mode == 1) {   // one subproblem around the stored (Xp, Up) (parity hook)
        linearize<MODEL>(K, toggle);
        IpmOut io;
        ipm_solve<MODEL>(K, P.sub_Delta[b] /* s */, P.sub_omega[b] /* mu */, 0.0, io, pf);
        store_traj(K, K.Xw, K.Uw, P.sub_X + (size_t)b * N * n, P.sub_U + (size_t)b * N * m);
        if (tid == 0) {
            P.sub_obj[b] = io.obj; P.sub_status[b] = io.status; P.sub_iters[b] = io.iters;
            for (int i = 0; i < n; i++) P.st_d[(size_t)b * SD_ND + SD_DUAL + i] = K.nu[i] * fmax(1.0, P.sub_omega[b]);
        }
        return;
    }
    double* w = P.ws + (size_t)slot * P.wl.total + P.wl.to_traj;
    double *Xpen = w, *Upen = Xpen + N * n, *Xcvx = Upen + N * m, *Ucvx = Xcvx + N * n;
    // SCPParam_TrajOpt ctor (:25-28): rho_vec = [0.], mu_vec = [mu0], s_vec = [s0], xtol_vec = ftol_vec = ctol_vec = [0.]
    int n_solves = 0, n_mu = 1, n_xtol = 1, n_ftol = 1, n_ctol = 1, nJ = 0, n_hist = 1, total_ipm = 0;
    int stop = GUSTO_STOP_MAXITER, converged = 0;
    if (tid == 0) {
        P.to_mu[hb] = tp.mu0; P.Delta[hb] = tp.s0; P.to_xtol[hb] = 0.0; P.to_ftol[hb] = 0.0; P.to_ctol[hb] = 0.0; P.rho[hb] = 0.0;
        P.solver_status[hb] = GUSTO_SOLVER_NA; P.ipm_it[hb] = 0; P.conv[hb] = 0.0;
    }
    double mu = tp.mu0, s = tp.s0;
    double Jt = cost_true(K, K.Up);                                               // :63
    if (tid == 0) P.J_true[hb + nJ] = Jt;
    nJ++;
    auto copy_traj = [&](double* Xd, double* Ud, const double* Xs, const double* Us) {
        K.sync();
        for (int e = tid; e < N * n; e += K.nt()) Xd[e] = Xs[e];
        for (int e = tid; e < N * m; e += K.nt()) Ud[e] = Us[e];
        K.sync();
    };
    const int room = P.hist_cap - 2;   // every vector gets at most one entry per solve plus the leading one
    bool constraints_satisfied = false, xtol_satisfied = false, halt = false;
    for (int pi = 0; pi < tp.max_penalty_iteration && !halt; pi++) {
        if (constraints_satisfied) break;
        copy_traj(Xpen, Upen, K.Xp, K.Up);                                        // :73 (a copy, not the alias of :67)
        for (int ci = 0; ci < tp.max_convex_iteration && !halt; ci++) {
            copy_traj(Xcvx, Ucvx, K.Xp, K.Up);                                    // :76
            if (constraints_satisfied) break;
            if (xtol_satisfied) { xtol_satisfied = false; break; }
            for (int ti = 0; ti < tp.max_trust_iteration; ti++) {
                if (n_solves >= P.max_iter) { halt = true; break; }
                if (n_solves >= room || n_xtol >= room) { halt = true; stop = GUSTO_STOP_HIST_FULL; break; }
                linearize<MODEL>(K, toggle);                                      // :94 update_model_params!
                IpmOut io;
                ipm_solve<MODEL>(K, s, mu, 0.0, io, pf);                          // :95-110
                total_ipm += io.iters;
                const int h = n_hist;
                if (tid == 0) { P.solver_status[hb + h] = io.status; P.ipm_it[hb + h] = io.iters; }
                if (io.status != GUSTO_SOLVER_OPTIMAL && io.status != GUSTO_SOLVER_ALMOST) {   // (:107-110 warns and goes on)
                    stop = GUSTO_STOP_SUBPROBLEM_FAILED; halt = true; break;
                }
                const double xt = convergence_metric_blk(K, K.Xw, Xcvx);           // evaluate_xtol :120-121
                const double rho = trajopt_ratio<MODEL>(K, K.Xw, K.Uw, Xcvx, Ucvx);  // :127
                const double s_n = (rho > tp.c) ? tp.tau_plus * s : tp.tau_minus * s;   // :128-132
                copy_traj(K.Xp, K.Up, K.Xw, K.Uw);                                // :134: every step is taken
                Jt = cost_true(K, K.Up);
                if (tid == 0) {
                    P.to_xtol[hb + n_xtol] = xt; P.conv[hb + h] = xt; P.J_full[hb + n_solves] = io.obj;
                    P.rho[hb + n_solves + 1] = rho; P.Delta[hb + n_solves + 1] = s_n; P.J_true[hb + nJ] = Jt;
                    for (int i = 0; i < n; i++) P.st_d[(size_t)b * SD_ND + SD_DUAL + i] = K.nu[i] * fmax(1.0, mu);
                }
                n_xtol++; nJ++; n_hist = h + 1; n_solves++;
                s = s_n;
                if (s < tp.xtol) { xtol_satisfied = true; break; }                // :140-143
            }
            if (halt) break;
            const double Jn = cost_true(K, K.Up), Jo = cost_true(K, Ucvx);
            const double ft = fabs(Jn - Jo) / fabs(Jn);                           // evaluate_ftol :146
            const double xt = convergence_metric_blk(K, K.Xp, Xcvx);              // :147
            if (tid == 0) { P.to_ftol[hb + n_ftol] = ft; P.to_xtol[hb + n_xtol] = xt; }
            n_ftol++; n_xtol++;
            if (ft < tp.ftol || xt < tp.xtol) { constraints_satisfied = true; break; }   // :148 (`xtol[end]` means xtol_vec[end])
        }
        if (halt) break;
        const double ct = trajopt_ctol<MODEL>(K, K.Xp, K.Up, Xpen, Upen);          // :155
        if (tid == 0) P.to_ctol[hb + n_ctol] = ct;
        n_ctol++;
        if (ct < tp.ctol) { constraints_satisfied = true; converged = 1; stop = GUSTO_STOP_CONVERGED; break; }
        mu *= tp.k;                                                               // :161
        if (tid == 0) P.to_mu[hb + n_mu] = mu;
        n_mu++;
    }
    pf.flush(P.prof, b, false);
    if (tid == 0) {
        sti[ST_ITER] = n_solves; sti[ST_CONV] = converged; sti[ST_SUCC] = 0; sti[ST_STOP] = stop; sti[ST_IPM] = total_ipm;
        sti[ST_NHIST] = n_hist; sti[ST_NJ] = nJ; sti[ST_NRHO] = n_solves + 1; sti[ST_NMU] = n_mu; sti[ST_NXTOL] = n_xtol;
        sti[ST_NFTOL] = n_ftol; sti[ST_NCTOL] = n_ctol;
    }
}
// problems in a static round robin over the resident workgroups (no device-side scheduler: TrajOpt problems take 5-25 solves)
template <int MODEL> __global__ void __launch_bounds__(256, 1) trajopt_kernel(const KParams P) {
    extern __shared__ __attribute__((aligned(16))) double lds[];
#ifdef GUSTO_DEBUG_LDS
    if (threadIdx.x == 0) gusto_dbg_lds_limit() = __builtin_amdgcn_groupstaticsize() + (unsigned)P.ll.total * 8u;
    __syncthreads();
#endif
    for (int b = blockIdx.x; b < P.B; b += gridDim.x) {
        __syncthreads();
        trajopt_problem<MODEL>(P, lds, b, blockIdx.x);
        __syncthreads();
    }
}

// ---- hardest first: the order in which fresh problems are handed out ------------------------------------------------
// The long problems of a batch are those that start deep inside an obstacle (their penalty weight is raised in the first
// trips and they run to max_iter): of the 64 longest problems of the freeflyer batch, 62 are among the 1024 with the
// largest penetration of the initial trajectory.  Handing those out first lets them start at t = 0 instead of wherever
// their index falls (simulated makespan of the B = 4096 batch: 533 -> 510 KKT units).  Key of a problem = the largest
// violation clearance - dist over the knots and obstacles of its stored trajectory, quantised to SCHED_BUCKETS levels of
// the robot size; the order is a counting sort by bucket, largest first.  Affects time only.
constexpr int SCHED_BUCKETS = 64;
template <int MODEL> __global__ void sched_key_kernel(const KParams P, int* bucket) {
    using T = MT<MODEL>;
    constexpr int n = T::n;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= P.B * P.N) return;
    if constexpr (T::HAS_OBS) {
        const int b = gid / P.N;
        const double* x = P.X + (size_t)gid * n;
        double xw[T::WS], nh[T::WS], worst = 0.0;
#pragma unroll
        for (int j = 0; j < T::WS; j++) xw[j] = x[j];
        const Env E = problem_env(P, b);
        for (int i = 0; i < E.n_obs; i++) worst = fmax(worst, P.mp.clearance - signed_distance<T::WS>(P, E, 0, xw, i, nh));
        const double full = 2.0 * (P.mp.radius + P.mp.clearance);      // a robot diameter inside an obstacle: the last bucket
        const int q = (int)fmin((double)(SCHED_BUCKETS - 1), worst / full * (SCHED_BUCKETS - 1));
        if (q > 0) atomicMax(bucket + b, q);
    }
}
// one workgroup: counting sort of the problems by bucket, largest first.  The place of a problem inside its bucket is
// whatever its atomic draws (the order only moves time, never results); three passes over B with all threads -- the first
// version, one thread per bucket walking all of B, took 0.27 ms at B = 2048.
static __global__ void sched_order_kernel(int B, const int* bucket, int* order) {
    __shared__ int cnt[SCHED_BUCKETS], start[SCHED_BUCKETS];
    const int tid = threadIdx.x, nt = blockDim.x;
    if (tid < SCHED_BUCKETS) cnt[tid] = 0;
    __syncthreads();
    for (int b = tid; b < B; b += nt) atomicAdd(&cnt[bucket[b]], 1);
    __syncthreads();
    if (tid == 0) {
        int acc = 0;
        for (int q = SCHED_BUCKETS - 1; q >= 0; q--) { start[q] = acc; acc += cnt[q]; }
    }
    __syncthreads();
    for (int b = tid; b < B; b += nt) order[atomicAdd(&start[bucket[b]], 1)] = b;
}

// straight-line initial trajectory (freeflyer_se2.jl:97-111): one thread per (problem, knot)
template <int MODEL> __global__ void init_straightline_kernel(const KParams P) {
    using T = MT<MODEL>;
    constexpr int n = T::n, m = T::m;
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    if (gid >= P.B * P.N) return;
    const int b = gid / P.N, k = gid % P.N;
    const double t = (double)k / (double)(P.N - 1);  // LinRange element: (1-t)*a + t*b
#pragma unroll
    for (int i = 0; i < n; i++) {
        const double lo = P.goal_lo[(size_t)b * n + i], hi = P.goal_hi[(size_t)b * n + i];
        const double xg = (isfinite(lo) && isfinite(hi)) ? 0.5 * (lo + hi) : 0.0;  // center(goal), zeros elsewhere
        P.X[((size_t)b * P.N + k) * n + i] = (1 - t) * P.x_init[(size_t)b * n + i] + t * xg;
    }
#pragma unroll
    for (int i = 0; i < m; i++) P.U[((size_t)b * P.N + k) * m + i] = 0.0;
}

// SCPSolution(SCPP, traj_init) + SCPParam_GuSTO ctor (types.jl:233, scp_gusto.jl:21-23)
template <int MODEL> __global__ void reset_state_kernel(const KParams P) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= P.B) return;
    int* sti = P.st_i + (size_t)b * ST_NI;
    for (int i = 0; i < ST_NI; i++) sti[i] = 0;
    sti[ST_NHIST] = 1; sti[ST_NRHO] = 1;
    const size_t hb = (size_t)b * P.hist_cap;
    P.solver_status[hb] = GUSTO_SOLVER_NA; P.scp_status[hb] = GUSTO_SCP_NA; P.accept[hb] = 1; P.conv[hb] = 0.0;
    P.ipm_it[hb] = 0; P.Delta[hb] = P.sp.Delta0; P.omega[hb] = P.sp.omega0; P.tr_sat[hb] = 0; P.cvx_sat[hb] = 0;
    P.rho[hb] = 0.0;
    double* sd = P.st_d + (size_t)b * SD_ND;
    for (int i = 0; i < SD_ND; i++) sd[i] = 0.0;
}

}  // namespace gusto
