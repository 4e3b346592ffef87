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

// One time slice of one problem: the body of the persistent kernel below.  `cont` = the problem has run before in this
// gusto_solve call (no new leading history entries), `trips` = how many GuSTO trips this slice may take.  Returns the
// penalty level of the problem (number of omega raises so far) if it has to come back for another slice, -1 if it stopped.
template <int MODEL, bool ONEWAVE> GD int scp_problem(const KParams& P, double* lds, int b_, int slot, bool cont, int trips) {
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
        ipm_solve<MODEL>(K, Delta, omega, (warm && !hook) ? P.io.mu_warm : 0.0, io, pf);  // :96-104
        if (hook) {
            pf.flush(P.prof, b, cont);
            store_traj(K, K.Xw, K.Uw, P.sub_X + (size_t)b * N * n, P.sub_U + (size_t)b * N * m);
            if (tid == 0) {
                P.sub_obj[b] = io.obj; P.sub_status[b] = io.status; P.sub_iters[b] = io.iters;
                for (int i = 0; i < n; i++) P.st_d[(size_t)b * SD_ND + SD_DUAL + i] = K.nu[i] * fmax(1.0, omega);
            }
            return -1;
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
// into the list of its penalty level (omega raises so far: the long problems are those raised early); workgroups take
// fresh problems while there are any, then always the highest level waiting, and from slice probe_visits on a problem
// runs to its end (simulated makespan 473).  Slicing is free: between trips a problem's whole state lives in HBM
// (trajectory, histories, st_i), so results are bit-identical to an unsliced run.
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
GD int sched_pop(const KParams& P, bool& cont) {
    int* Q = P.queue;
    for (int spin = 0;; spin++) {
        // BEFORE the scan, with acquire: a push is published (tail increment, release) before SQ_PROBING drops
        const int probing_seen = uload_acq(Q + SQ_PROBING);
        if (uload(Q + SQ_HEAD_A) < P.B) {
            const int q = uadd(Q + SQ_HEAD_A, 1);
            if (q < P.B) { cont = false; return P.order ? uload(P.order + q) : q; }
        }
        for (int L = SCHED_LEVELS - 1; L >= 0; L--) {
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
                    cont = true;
                    return e;
                }
                h = found;
            }
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
    const int slot = blockIdx.x;
    for (;;) {
        int b = 0, ci = 0;
        if constexpr (ONEWAVE) {
            bool c = false;
            b = sched_pop(P, c); ci = c;
        } else {
            __shared__ int sh_b, sh_cont;
            __syncthreads();               // (everyone is done with the previous problem's LDS)
            if (threadIdx.x < 64) {        // wave 0 asks the scheduler, the others get the answer through LDS
                bool c = false;
                b = sched_pop(P, c);
                if (threadIdx.x == 0) { sh_b = b; sh_cont = c; }
            }
            __syncthreads();
            b = sh_b; ci = sh_cont;
        }
        if (b < 0) return;
        const bool cont = ci != 0;
        const int visits = cont ? (b >> 24) : 0;     // time slices this problem has had in this gusto_solve call
        b &= (1 << 24) - 1;
        // hipcc (ROCm 7.2) otherwise forms some of the problem's base addresses from the UNMASKED register (seen in the
        // ISA: s_and_b32 for tf[b], but v_mad_u64_u32 with the raw entry for goal_lo + b * n): pin the masked value
        asm volatile("" : "+v"(b));
        b = __builtin_amdgcn_readfirstlane(b);
        if (cont) {   // the state another workgroup left in HBM: drop whatever this CU's L1 still holds of it.  ONE lane
            // issues the invalidate (the L1 is the CU's, not the lane's), then the workgroup synchronises.
            if (threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            blk_sync<ONEWAVE>();
        }
#ifdef GUSTO_SCHED_DEBUG
        if (b >= P.B) { if (threadIdx.x == 0) printf("sched: bad b %d (ci %d visits %d) slot %d\n", b, ci, visits, slot); return; }
#endif
        const int trips = (P.mode == 0 && visits < P.probe_visits) ? 1 : (1 << 30);
        const int lvl = scp_problem<MODEL, ONEWAVE>(P, lds, b, slot, cont, trips);
        blk_sync<ONEWAVE>();               // the next problem reuses this workgroup's LDS and workspace slot
        if constexpr (!ONEWAVE) __syncthreads();
        if (threadIdx.x == 0) {
#ifdef GUSTO_SCHED_DEBUG
            printf("sched: slot %d problem %d cont %d visits %d trips %d -> lvl %d\n", slot, b, (int)cont, visits, trips, lvl);
#endif
            if (lvl >= 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");       // state first ...
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int idx = atomicAdd(P.queue + SQ_TAIL + lvl * SQ_STRIDE, 1);
                __hip_atomic_store(P.lists + (size_t)lvl * P.list_cap + idx, ((visits + 1) << 24) | b, __ATOMIC_RELAXED,
                                   __HIP_MEMORY_SCOPE_AGENT);                                                        // ... then the entry
            }
            // this was the problem's last probing slice (or it stopped inside one): it will not be pushed again
            // (release: the tail increment and the entry above are visible to whoever sees the counter drop)
            if (trips == 1 && (lvl < 0 || visits + 1 >= P.probe_visits))
                __hip_atomic_fetch_sub(P.queue + SQ_PROBING, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
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
        for (int i = 0; i < P.n_obs; i++) worst = fmax(worst, P.mp.clearance - signed_distance<T::WS>(P, 0, xw, i, nh));
        const double full = 2.0 * (P.mp.radius + P.mp.clearance);      // a robot diameter inside an obstacle: the last bucket
        const int q = (int)fmin((double)(SCHED_BUCKETS - 1), worst / full * (SCHED_BUCKETS - 1));
        if (q > 0) atomicMax(bucket + b, q);
    }
}
// one workgroup: counting sort of the problems by bucket, largest first (order inside a bucket: by index)
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
    // stable and deterministic: thread q places the problems of bucket q in index order (B <= a few 10^4: microseconds)
    if (tid < SCHED_BUCKETS && cnt[tid] > 0) {
        int at = start[tid];
        for (int b = 0; b < B; b++)
            if (bucket[b] == tid) order[at++] = b;
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
