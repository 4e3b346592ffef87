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
    constexpr int m = BLK::m;
    const int k = K.tid;
    double l = 0;
    if (k >= 1 && k < K.N) {
#pragma unroll
        for (int j = 0; j < m; j++) l += 0.5 * K.dt * (U[(k - 1) * m + j] * U[(k - 1) * m + j] + U[k * m + j] * U[k * m + j]);
    }
    return block_reduce(l, OpSum(), K.misc);
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
            for (int c = 0; c < K.P.mp.n_robot_comp; c++)
                for (int i = 0; i < K.P.n_obs; i++) {
                    double nh[T::WS], nh1[T::WS];
                    const double d0 = signed_distance<T::WS>(K.P, c, xp, i, nh);
                    double lin = K.P.mp.clearance - d0;
#pragma unroll
                    for (int j = 0; j < T::WS; j++) lin -= nh[j] * (x[j] - xp[j]);
                    const double d1 = signed_distance<T::WS>(K.P, c, x, i, nh1);
                    num += fabs((K.P.mp.clearance - d1) - lin);
                    den += fabs(lin);
                }
        }
    }
    num = block_reduce(num, OpSum(), K.misc);
    den = block_reduce(den, OpSum(), K.misc);
    return num / den;
}

template <class BLK> GD void store_traj(BLK& K, const double* Xs, const double* Us, double* Xg, double* Ug) {
    constexpr int n = BLK::n, m = BLK::m;
    for (int e = K.tid; e < K.N * n; e += K.nt()) Xg[e] = Xs[e];
    for (int e = K.tid; e < K.N * m; e += K.nt()) Ug[e] = Us[e];
}

// One problem, start to stop: the body of the persistent kernel below.
template <int MODEL, bool ONEWAVE> GD void scp_problem(const KParams& P, double* lds, int b_, int slot) {
    using T = MT<MODEL>;
    constexpr int n = T::n, m = T::m;
    Blk<MODEL, ONEWAVE> K(P, lds, b_, slot);
    Prof pf;
    const int b = K.b, tid = K.tid, N = K.N, k = tid;
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
    const int iter_cap = iterations + P.max_iter;  // scp_gusto.jl:67
    // second launch of one gusto_solve call: problems that already stopped are done, the others carry on exactly
    // where the first launch left them (no new leading history entries)
    if (P.cont && sti[ST_STOP] != GUSTO_STOP_MAXITER) return;

    // K.Xp / K.Up are the stored trajectory (SCPS.traj) itself
    // scp_gusto.jl:73-76
    double Jt = 0, rho0v = 0;
    if (!hook) {
        Jt = cost_true(K, K.Up);
        rho0v = trust_region_ratio<MODEL>(K, K.Xp, K.Up, K.Xp, K.Up);
    }
    double Delta = hook ? P.sub_Delta[b] : P.Delta[hb + n_hist - 1], omega = hook ? P.sub_omega[b] : P.omega[hb + n_hist - 1];
    if (!P.cont && !hook) {
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
        ipm_solve<MODEL>(K, Delta, omega, (warm && !hook) ? P.io.mu_warm : 0.0, io, pf);  // :96-104
        if (hook) {
            pf.flush(P.prof, b);
            store_traj(K, K.Xw, K.Uw, P.sub_X + (size_t)b * N * n, P.sub_U + (size_t)b * N * m);
            if (tid == 0) {
                P.sub_obj[b] = io.obj; P.sub_status[b] = io.status; P.sub_iters[b] = io.iters;
                for (int i = 0; i < n; i++) P.st_d[(size_t)b * SD_ND + SD_DUAL + i] = K.nu[i] * fmax(1.0, omega);
            }
            return;
        }
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
            ctx.goal_lo = K.goal_lo; ctx.goal_hi = K.goal_hi;
            OpCheck op{sp.eps};
            visit_rows<MODEL>(ctx, xs, us, op);
            cvx_l = op.ok;
        }
        const double max_d2 = block_reduce(dn, OpMax(), K.misc);
        const double max_x2 = block_reduce(xn, OpMax(), K.misc);
        const double conv = sqrt(max_d2) / sqrt(max_x2);
        const int cvx_sat = block_reduce(cvx_l ? 0.0 : 1.0, OpMax(), K.misc) == 0.0;
        // the literal `max_val - Delta <= 0` evaluated with the solver's accuracy as slack
        const int tr_sat = (max_d2 - Delta <= P.io.tr_tol * fmax(1.0, Delta));
        int accept, status;
        double Delta_n, omega_n;
        if (tr_sat) {                                       // :123-141
            const double rho = trust_region_ratio<MODEL>(K, K.Xw, K.Uw, K.Xp, K.Up);
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
    // a history vector is full although iterations remain: say so instead of posing as MaxIter
    if (stop == GUSTO_STOP_MAXITER && iterations < iter_cap) stop = GUSTO_STOP_HIST_FULL;
    pf.tick(PF_SCP);
    pf.flush(P.prof, b);
    if (tid == 0) {
        sti[ST_ITER] = iterations; sti[ST_CONV] = converged; sti[ST_SUCC] = successful; sti[ST_STOP] = stop;
        sti[ST_IPM] = total_ipm; sti[ST_NHIST] = n_hist; sti[ST_NJ] = nJ; sti[ST_NRHO] = n_rho; sti[ST_WARM] = warm;
        std_[SD_TOGGLE] = toggle;
    }
}

// The kernel: PERSISTENT workgroups (the grid is the number of resident slots, launch.hpp) pull problems from a work
// queue -- one atomicAdd per problem on P.queue -- until it is empty.  A problem that needs 30 trips and one that needs
// 5 occupy their slot for different times and the queue backfills; P.order (longest-first schedule) maps queue
// positions to problems.  Problems are independent: no grid-wide synchronisation, no inter-workgroup data.
template <int MODEL, bool ONEWAVE> __global__ void __launch_bounds__(ONEWAVE ? 64 : 256, ONEWAVE ? MT<MODEL>::WAVES_PER_EU : 1)
scp_kernel(const KParams P) {
    extern __shared__ double lds[];
    const int slot = blockIdx.x;
    for (;;) {
        int q = 0;
        if (threadIdx.x == 0) q = atomicAdd(P.queue, 1);
        if constexpr (ONEWAVE) {
            q = __builtin_amdgcn_readfirstlane(q);
        } else {
            __shared__ int q_sh;
            __syncthreads();               // (also: everyone is done with the previous problem's LDS)
            if (threadIdx.x == 0) q_sh = q;
            __syncthreads();
            q = q_sh;
        }
        if (q >= P.B) return;
        const int b = P.order ? P.order[q] : q;
        scp_problem<MODEL, ONEWAVE>(P, lds, b, slot);
        blk_sync<ONEWAVE>();               // the next problem reuses this workgroup's LDS and workspace slot
    }
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

// Longest-first order for the second launch of a gusto_solve call.  Problem lengths are not known in advance, but
// the long ones are almost exactly those whose penalty weight omega was raised during their first trips
// (ViolatesConstraints / TrustRegionViolated, scp_gusto.jl:137-147).  Key = number of omega raises so far, problems
// that already stopped last; counting sort by descending key, stable in the problem index.  One workgroup.
template <int MODEL> __global__ void __launch_bounds__(256) order_kernel(const KParams P, int* order) {
    constexpr int NB = 16, NT = 256, KC = 16;
    __shared__ int cnt[NB][NT + 1];
    __shared__ int base[NB];
    const int t = threadIdx.x, B = P.B;
    const int chunk = (B + NT - 1) / NT, b0 = t * chunk, b1 = min(B, b0 + chunk);
    auto key = [&](int b) {
        const int* sti = P.st_i + (size_t)b * ST_NI;
        if (sti[ST_STOP] != GUSTO_STOP_MAXITER) return 0;
        const double w = P.omega[(size_t)b * P.hist_cap + sti[ST_NHIST] - 1] / P.sp.omega0;
        int lvl = 0;
        for (double x = 1.5; x < w && lvl < NB - 2; x *= P.sp.gamma_fail) lvl++;
        return 1 + lvl;
    };
    int kc[KC];   // keys of this thread's problems (batches up to NT * KC problems: no second trip to memory)
    for (int q = 0; q < NB; q++) cnt[q][t] = 0;
#pragma unroll
    for (int i = 0; i < KC; i++) kc[i] = (b0 + i < b1) ? key(b0 + i) : -1;
#pragma unroll
    for (int i = 0; i < KC; i++) if (kc[i] >= 0) cnt[kc[i]][t]++;
    for (int b = b0 + KC; b < b1; b++) cnt[key(b)][t]++;
    __syncthreads();
    if (t < NB) {   // exclusive prefix of row t over the threads; row totals -> offsets in descending key order
        int acc = 0;
        for (int u = 0; u < NT; u++) { const int c = cnt[t][u]; cnt[t][u] = acc; acc += c; }
        cnt[t][NT] = acc;
    }
    __syncthreads();
    if (t == 0) {
        int acc = 0;
        for (int q = NB - 1; q >= 0; q--) { base[q] = acc; acc += cnt[q][NT]; }
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < KC; i++) if (kc[i] >= 0) order[base[kc[i]] + cnt[kc[i]][t]++] = b0 + i;
    for (int b = b0 + KC; b < b1; b++) { const int q = key(b); order[base[q] + cnt[q][t]++] = b; }
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
