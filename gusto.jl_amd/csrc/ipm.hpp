// ipm.hpp -- the convex subproblem of one GuSTO iteration (scp_gusto.jl:178-314) solved on device.
//
// One workgroup owns one problem; thread k owns knot k for everything that is stage-local (linearisation,
// rows, residuals, Hessian blocks, step lengths) and the whole workgroup cooperates, entry-per-lane, on the
// three sequential sweeps of the block-banded Newton/KKT system:
//
//   trapezoid rows (freeflyer_se2.jl:160-172) in Newton form, y_k := F_k x_k + b_k u_k, M_k = (I - dt/2 A_k)^-1:
//       dx_k = M_k (dy_{k-1} + b_k du_k + rd_k),   dy_k = Phi_k dy_{k-1} + Gam_k du_k + c_k,
//       Phi_k = 2 M_k - I,  Gam_k = 2 M_k b_k,  c_k = Phi_k rd_k
//   -> an LQR in the n-dim state dy.  factor_sweep() is the Riccati recursion (matrix work, 4 LDS phases per
//   knot); each right-hand side then costs two affine vector recurrences with the closed-loop matrix
//   Phicl_k = Phi_k - Gam_k K_k (backward for the value gradient p_k, forward for dy_k); everything else is
//   stage-parallel.  Goal point rows x_N[i] = g_i carry multipliers mu_g (kept in state-index space: entry i is
//   the multiplier of the goal row on x_N[i], zero when coordinate i has no point goal) whose sensitivities
//   Pi_k = dp_k/dmu_g ride along the factor sweep, so mu_g is known after the backward sweep.
//
// The reference hands this problem to JuMP -> Ipopt/Gurobi (scp_gusto.jl:82-104); the interior point method
// here is a Mehrotra predictor-corrector on the same problem (same optimum, it is strictly convex in U).
#pragma once
#include "rows.hpp"

#ifndef GUSTO_COOP_CHOL_MIN
#define GUSTO_COOP_CHOL_MIN 7   // control blocks from this size on are factored once per workgroup in LDS (factor_sweep_mw)
#endif

namespace gusto {

// the dynamic LDS of the workgroup (the same memory as the `extern __shared__` array of scp_kernel)
extern __shared__ __attribute__((aligned(16))) double gusto_dyn_lds[];

template <int MODEL> constexpr bool costate_adjoint();

struct IpmOut {
    int status, iters;
    double obj, res_p, res_d, mu;
};

// Workgroup-wide ordering point.  A problem with N <= 64 knots runs as ONE wave: its LDS and global accesses
// are already ordered in hardware, so the only thing needed is to stop the compiler from moving accesses
// across the point -- no s_barrier and, crucially, no `s_waitcnt vmcnt(0)`, which is what __syncthreads()
// costs and which would drain the operand prefetches of the sequential sweeps at every knot.
template <bool ONEWAVE> GD void blk_sync() {
    if constexpr (ONEWAVE) {
#ifdef GUSTO_STRICT_SYNC
        // (check build, tools/strict_sync.sh: every ordering point of a one-wave problem drains the wave's LDS and memory
        // operations and passes a real barrier -- results must not move by a bit, profiles/r06_dubins_scan_repro.txt)
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

template <int MODEL, bool ONEWAVE> struct Blk {
    using T = MT<MODEL>;
    using C = LdsC<MODEL, ONEWAVE>;
    using R = Rec<MODEL>;
    static constexpr int n = T::n, m = T::m, NZ = n + m;
    static constexpr bool ONE = ONEWAVE;
    static constexpr int MODEL_ID = MODEL;
    static constexpr int SPH = R::SNN;   // stride of the Phicl records where the one-wave vector sweeps run on this view (the global, padded records)
    const KParams& P;
    int b, tid, N;
    int NTr;  // runtime block size (multi-wave problems)
    double dt;
    double* lds;
    // LDS working set (compile-time offsets)
    LPtr<double> sP, sPi, sPG, sT, sHh, sZ, sK, sD, sW, sV, sGd, misc;
    LPtr<int> lut;  // packed upper-triangle index -> (i << 8 | j)
    // LDS per-knot vectors
    LPtr<double> Xw, dY, pv, cv, rv, nu, nun, Uw;
    // knot-private vectors and the linearisation point (global)
    GPtr<double> Xp, rd, qrd, dXs, Up, qu, dv, dUs, gAx, gBx, gAu, gBu;
    // global workspace of this problem
    GPtr<double> rowstate, obs_nh, obs_c0, PG, PGS, QQ, Paft, Piaft, KD, Phicl;
    LPtr<double> kdl;   // K | D | S^-1 (upper triangle) per knot in LDS, stride C::KDS (LdsC::KD_LDS); before factor stage k the slot of knot k holds QQ_k
    LPtr<double> pgl;   // [Phi Gam] per knot in LDS (LdsC::PG_LDS)
    LPtr<double> lcl;   // linearisation cache per knot in LDS (LdsC::LC_LDS)
    GPtr<uint64_t> obs_mask;
    GPtr<const double> x_init, goal_lo, goal_hi;
    unsigned goalmask;  // bit i: coordinate i of x_N has a point goal (goal_lo == goal_hi)
    unsigned boxmask;   // bit i: coordinate i of x_N has BoxGoal rows (goal_lo != goal_hi, at least one of them finite)
    // the keep-out set of this problem (models.hpp), re-derived where a trip needs it (linearize, rho, the hand-out order) and
    // not kept: the interior point loop never looks at it and has no scalar registers to spare
    GD Env env() const { return problem_env(P, b); }

    // Knot-private vectors (only lane k ever touches those of knot k): rd qrd dXs | dUs qu dv | gAx gBx gAu gBu.  The one-wave
    // kernels whose phases are inlined keep them in REGISTERS of lane k over the whole interior point iteration (the
    // allocator parks them in AGPRs across the sweeps) instead of the per-slot global workspace: every phase started with
    // a round trip to L2 for them (~45 loads and as many stores per lane and iteration).  The called phases of the 12/13-state
    // models and the multi-wave kernels keep the workspace arrays.
    static constexpr bool PVT_REG = ONEWAVE && !T::SWEEP_CALL && T::NDEF == 0 && T::WAVES_PER_EU == 1;   // (dubins, 2 waves per SIMD, has no registers to spare: 181 -> 201 ms)
    struct PvtRegs { double rd[n], qrd[n], dXs[n], dUs[m], qu[m], dv[m], gAx[n], gBx[n], gAu[m], gBu[m]; };
    std::conditional_t<PVT_REG, PvtRegs, char> pr;
#define GUSTO_PVT(name, dim)                                                                      \
    GD decltype(auto) name##_(int k, int i) {                                                     \
        if constexpr (PVT_REG) return (pr.name[i]); else return (name[k * dim + i]);              \
    }
    GUSTO_PVT(rd, n) GUSTO_PVT(qrd, n) GUSTO_PVT(dXs, n) GUSTO_PVT(dUs, m) GUSTO_PVT(qu, m) GUSTO_PVT(dv, m)
    GUSTO_PVT(gAx, n) GUSTO_PVT(gBx, n) GUSTO_PVT(gAu, m) GUSTO_PVT(gBu, m)
#undef GUSTO_PVT
    GD int nt() const { return ONEWAVE ? 64 : NTr; }

    // every LDS pointer from the base of the dynamic LDS.  Also called at the top of the phases that run as real calls
    // (MT::SWEEP_CALL) with the __shared__ symbol itself: a pointer that crosses a call is generic to the compiler and
    // its accesses become flat_load/flat_store; re-derived from the symbol inside the callee they are ds_ operations.
    GD void rebind_lds(double* l) {
        lds = l;
        sP = lds + C::sP; sPi = lds + C::sPi; sPG = lds + C::sPG; sT = lds + C::sT; sHh = lds + C::sHh; sZ = lds + C::sZ;
        sK = lds + C::sK; sD = lds + C::sD; sW = lds + C::sW; sV = lds + C::sV; sGd = lds + C::sGd; misc = lds + C::misc;
        lut = reinterpret_cast<int*>(lds + C::lut);
        double* v = lds + C::vecs;
        Xw = v; dY = v + N * n; pv = v + 2 * N * n; cv = v + 3 * N * n; rv = v + 4 * N * n; nu = v + 5 * N * n;
        nun = v + 6 * N * n;
        Uw = v + C::NVN * N * n;
        if constexpr (C::KD_LDS) kdl = lds + P.ll.kd;
        if constexpr (C::PG_LDS) pgl = lds + P.ll.pg;
        if constexpr (C::LC_LDS) lcl = lds + P.ll.lc;
    }
    // entry e of the K | D block of knot k (e = i n + j of K, m n + i n + j of D) and S_k^-1[i][l]
    GD double kd(int k, int e) const {
        if constexpr (C::KD_LDS) return kdl[k * C::KDS + e];
        else return KD[(size_t)k * R::SKD + e];
    }
    GD void qq_put(int k, int e, double v) const {   // entry e of the packed stage cost of knot k
        if constexpr (C::KD_LDS) kdl[k * C::KDS + e] = v;
        else QQ[(size_t)k * R::SQQ + e] = v;
    }
    GD double kdS(int k, int i, int l) const {
        if constexpr (C::KD_LDS) return kdl[k * C::KDS + 2 * m * n + sidx(i, l, m)];
        else return KD[(size_t)k * R::SKD + R::sS(i, l)];
    }

    GD void rebind_global() {   // (see rebind_lds)
        // (the members carry their address space, GPtr / LPtr; only the alignment promise is renewed)
        PG = al16((double*)PG); QQ = al16((double*)QQ); Paft = al16((double*)Paft); Piaft = al16((double*)Piaft); KD = al16((double*)KD);
    }

    // What a phase that runs as a REAL CALL (MT::SWEEP_CALL) receives instead of the Blk itself.  Passed by value the Blk is
    // ~650 bytes of per-lane arguments: the AMDGPU calling convention has no scalar arguments for such a struct, so every
    // wave-uniform pointer travels as a VGPR (x 64 lanes) and, beyond the argument registers, through scratch -- ~80 scratch
    // stores in front of each call and as many loads in the callee, eleven calls per KKT solve (the stack stays in the L2:
    // the fabric counters FETCH_SIZE / WRITE_SIZE do not see it, the wave waits for it all the same).  The callee now
    // gets these eight dwords, makes them scalar again (readfirstlane) and rebuilds the view from the kernel arguments
    // with scalar loads, so the ~45 base pointers live in SGPRs there as they do in the kernel body: scratch 3.1 -> 1.2 KB
    // per lane, astrobeeSE3 B = 8192 122.7 -> 114.5 ms.
    struct Args {
        const KParams* Pk;
        int b, slot;
        unsigned goalmask, boxmask;
        double dt;
    };
    int slot_;            // (the workspace slot, kept for args())
    const KParams* Pk;    // the kernel arguments where they really live: the kernarg segment (KParams is the FIRST argument of
                          // every kernel that reaches a called phase).  The address of the kernel's own `P` must not travel --
                          // the compiler may keep that copy in private memory, whose flat address is lane-swizzled scratch: read
                          // through the constant address space it is an aperture violation -- and a callee cannot ask for the
                          // segment itself (__builtin_amdgcn_kernarg_segment_ptr() folds to null outside a kernel).
    GD Args args() const { return Args{Pk, b, slot_, goalmask, boxmask, dt}; }
    GD static const KParams* uniform_ptr(const KParams* p) {
        const unsigned long long u = (unsigned long long)(uintptr_t)p;
        const unsigned lo = __builtin_amdgcn_readfirstlane((int)(u & 0xffffffffu)), hi = __builtin_amdgcn_readfirstlane((int)(u >> 32));
        return (const KParams*)(uintptr_t)(((unsigned long long)hi << 32) | lo);
    }
    // callee side of a real call: the same view, every base pointer from scalar loads of the kernel arguments (through the
    // constant address space); nothing is initialised (the goal values in LDS, the index table: the kernel body did that)
    GD Blk(const Args& a, double* lds_) : P(*uniform_ptr(a.Pk)), lds(lds_) {
        typedef const __attribute__((address_space(4))) KParams CP;
        Pk = &P;
        CP* Pc = (CP*)(uintptr_t)Pk;
        b = __builtin_amdgcn_readfirstlane(a.b); slot_ = __builtin_amdgcn_readfirstlane(a.slot);
        tid = ONEWAVE ? (threadIdx.x & 63) : threadIdx.x;   // (the helper wave of scp_kernel_w2 runs called phases too: lanes 64..127)
        NTr = ONEWAVE ? 64 : blockDim.x; N = Pc->N;
        rebind_lds(lds_);
        double* w = Pc->ws + (size_t)slot_ * Pc->wl.total;
        rowstate = w + Pc->wl.rowstate; obs_nh = w + Pc->wl.obs_nh; obs_c0 = w + Pc->wl.obs_c0;
        obs_mask = reinterpret_cast<uint64_t*>(w + Pc->wl.obs_mask);
        PG = al16(w + Pc->wl.PG); PGS = al16(w + Pc->wl.PGS); QQ = al16(w + Pc->wl.QQ); Paft = al16(w + Pc->wl.Paft); Piaft = al16(w + Pc->wl.Piaft);
        KD = al16(w + Pc->wl.KD); Phicl = w + Pc->wl.Phicl;
        {
            double* q = w + Pc->wl.pvt;
            rd = q; qrd = q + N * n; dXs = q + 2 * N * n; dUs = q + 3 * N * n; qu = dUs + N * m; dv = qu + N * m;
            gAx = dv + N * m; gBx = gAx + N * n; gAu = gBx + N * n; gBu = gAu + N * m;
            Xp = Pc->X + (size_t)b * N * n; Up = Pc->U + (size_t)b * N * m;
        }
        x_init = Pc->x_init + (size_t)b * n; goal_lo = Pc->goal_lo + (size_t)b * n; goal_hi = Pc->goal_hi + (size_t)b * n;
        const unsigned long long du = __builtin_bit_cast(unsigned long long, a.dt);
        dt = __builtin_bit_cast(double, ((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(du >> 32)) << 32) |
                                            (unsigned)__builtin_amdgcn_readfirstlane((int)(du & 0xffffffffu)));
        goalmask = __builtin_amdgcn_readfirstlane((int)a.goalmask); boxmask = __builtin_amdgcn_readfirstlane((int)a.boxmask);
    }

    // b = the problem, slot = the resident workgroup: the interior point workspace belongs to the SLOT (a few hundred
    // KB that every problem this workgroup pulls from the queue reuses, so the working set of a launch is
    // #slots x wl.total, cache-resident, instead of B x wl.total streamed through HBM once per problem)
    GD Blk(const KParams& P_, double* lds_, int b_, int slot) : P(P_), lds(lds_) {
        b = b_; tid = threadIdx.x; NTr = blockDim.x; N = P.N; slot_ = slot;
        Pk = (const KParams*)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();   // (this constructor is inlined into the kernel)
#ifdef GUSTO_DEBUG_LDS
        {   // the called phases read the kernel arguments THROUGH this pointer: KParams must be the first argument, at offset 0
            typedef const __attribute__((address_space(4))) KParams CP;
            CP* Pc = (CP*)(uintptr_t)Pk;
            if (Pc->N != P.N || Pc->B != P.B || Pc->ws != P.ws) {
                if (threadIdx.x == 0) printf("gusto: the kernarg segment does not start with this launch's KParams (block %d)\n", (int)blockIdx.x);
                __builtin_trap();
            }
        }
#endif
        rebind_lds(lds_);
        double* w = P.ws + (size_t)slot * P.wl.total;
        const WsLayout& W = P.wl;
        rowstate = w + W.rowstate; obs_nh = w + W.obs_nh; obs_c0 = w + W.obs_c0;
        obs_mask = reinterpret_cast<uint64_t*>(w + W.obs_mask);
        PG = al16(w + W.PG); PGS = al16(w + W.PGS); QQ = al16(w + W.QQ); Paft = al16(w + W.Paft); Piaft = al16(w + W.Piaft); KD = al16(w + W.KD); Phicl = w + W.Phicl;
        {   // knot-private vectors in the global workspace, the linearisation point = the stored trajectory
            double* q = w + W.pvt;
            rd = q; qrd = q + N * n; dXs = q + 2 * N * n; dUs = q + 3 * N * n; qu = dUs + N * m; dv = qu + N * m;
            gAx = dv + N * m; gBx = gAx + N * n; gAu = gBx + N * n; gBu = gAu + N * m;
            Xp = P.X + (size_t)b * N * n; Up = P.U + (size_t)b * N * m;
        }
        x_init = P.x_init + (size_t)b * n; goal_lo = P.goal_lo + (size_t)b * n; goal_hi = P.goal_hi + (size_t)b * n;
        dt = P.tf[b] / (N - 1);  // Trajectory(X,U,Tf): dt = Tf/(N-1), types.jl:235
        goalmask = 0; boxmask = 0;
#pragma unroll
        for (int i = 0; i < n; i++) {
            const double lo = goal_lo[i], hi = goal_hi[i];
            if (lo == hi) goalmask |= 1u << i;
            else if (isfinite(lo) || isfinite(hi)) boxmask |= 1u << i;
        }
        // the point-goal values next to the reduction scratch: the phases read them in every pass, and as loads from the
        // problem's arrays in HBM/L2 each was a memory round trip of its own on the critical path of a one-wave problem
        if (tid < n) (lds + C::sgoal)[tid] = goal_lo[tid];
        for (int e = tid; e < NZ * (NZ + 1) / 2; e += nt()) {
            int i = 0, rem = e;
            while (rem >= NZ - i) { rem -= NZ - i; i++; }
            lut[e] = (i << 8) | (i + rem);
        }
    }
    // (the multi-wave kernels also run with ONE wave -- the TrajOpt variants for N <= 64 --: no s_barrier then, and none of the
    // s_waitcnt vmcnt(0) a __syncthreads() implies)
    GD void sync() const {
        if constexpr (ONEWAVE) blk_sync<true>();
        else if (NTr <= 64) blk_sync<true>();
        else blk_sync<false>();
    }
    GD auto PGk(int k) const {
        if constexpr (C::PG_LDS) return pgl + k * n * NZ;
        else return PG + (size_t)(T::LTI ? 0 : k) * n * NZ;
    }
    // the P | Pi record after knot k (k >= -1) and its dummy slot
    GD auto pprec(int k) const {
        return Paft + (size_t)k * R::SNN;
    }
    static constexpr int PP_DUMMY = R::SNN - 1;
    GD bool is_goal(int i) const { return (goalmask >> i) & 1u; }
};

// M = (I - dt/2 A(x, u))^-1 and B of one knot: the linearisation of linearize(), and of the stage-parallel phases of the
// 12/13-state models, which RECOMPUTE M and Gam (~350 instructions) instead of walking the knot's [Phi Gam] record
// (~80 loads of one cache line per lane each, five times per interior point iteration).
template <int MODEL, class XP, class UP>
GD void stage_M(const gusto_model_params& mp, XP xp, UP up, double h, double* M, double* B) {
    using T = MT<MODEL>;
    constexpr int n = T::n, m = T::m;
    double xl[n], ul[m];   // (read through the caller's typed pointer: a plain pointer parameter would be generic)
#pragma unroll
    for (int i = 0; i < n; i++) xl[i] = xp[i];
#pragma unroll
    for (int i = 0; i < m; i++) ul[i] = up[i];
    double A[n * n], G[n * n];
    Dyn<MODEL>::A(mp, xl, ul, A);
    Dyn<MODEL>::B(mp, B);
#pragma unroll
    for (int i = 0; i < n; i++)
#pragma unroll
        for (int j = 0; j < n; j++) G[i * n + j] = (i == j ? 1.0 : 0.0) - h * A[i * n + j];
    if constexpr (MODEL == GUSTO_ASTROBEE_SE3 || MODEL == GUSTO_ASTROBEE_SE3_MANIFOLD || MODEL == GUSTO_TO_ASTROBEE_SE3 ||
                  MODEL == GUSTO_TO_ASTROBEE_SE3_MANIFOLD) {
        // x = (r, v, attitude a, w): G = I - hA = [I -hI 0 0; 0 I 0 0; 0 0 P Q; 0 0 0 W] (Anz), so
        // M = G^-1 = [I hI 0 0; 0 I 0 0; 0 0 P^-1 -P^-1 Q W^-1; 0 0 0 W^-1]: two small inverses instead of a
        // pivoted 12 x 24 / 13 x 26 elimination in registers (which was most of this phase for these models)
        constexpr int q = n - 9;   // attitude block: 3 (MRP) or 4 (quaternion)
        double Pm[q * q], Pi[q * q], Wm[9], Wi[9], Qm[q * 3], T1[q * 3];
#pragma unroll
        for (int i = 0; i < q; i++) {
#pragma unroll
            for (int j = 0; j < q; j++) Pm[i * q + j] = G[(6 + i) * n + 6 + j];
#pragma unroll
            for (int j = 0; j < 3; j++) Qm[i * 3 + j] = G[(6 + i) * n + 6 + q + j];
        }
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) Wm[i * 3 + j] = G[(6 + q + i) * n + 6 + q + j];
        inv_gauss_jordan<q>(Pm, Pi);
        inv_gauss_jordan<3>(Wm, Wi);
#pragma unroll
        for (int i = 0; i < q; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) {
                double t = 0;
#pragma unroll
                for (int l = 0; l < 3; l++) t += Qm[i * 3 + l] * Wi[l * 3 + j];
                T1[i * 3 + j] = t;
            }
#pragma unroll
        for (int i = 0; i < n * n; i++) M[i] = 0.0;
#pragma unroll
        for (int i = 0; i < 6; i++) M[i * n + i] = 1.0;
#pragma unroll
        for (int i = 0; i < 3; i++) M[i * n + 3 + i] = h;
#pragma unroll
        for (int i = 0; i < q; i++) {
#pragma unroll
            for (int j = 0; j < q; j++) M[(6 + i) * n + 6 + j] = Pi[i * q + j];
#pragma unroll
            for (int j = 0; j < 3; j++) {
                double t = 0;
#pragma unroll
                for (int l = 0; l < q; l++) t -= Pi[i * q + l] * T1[l * 3 + j];
                M[(6 + i) * n + 6 + q + j] = t;
            }
        }
#pragma unroll
        for (int i = 0; i < 3; i++)
#pragma unroll
            for (int j = 0; j < 3; j++) M[(6 + q + i) * n + 6 + q + j] = Wi[i * 3 + j];
    } else {
        inv_gauss_jordan<n>(G, M);
    }
}

// M_k and Gam_k of knot k (k >= 1) from the stored [Phi | Gam] block
template <class BLK> GD void load_M_Gam(const BLK& K, int k, double* M, double* Gam) {
    using T = typename BLK::T;
    constexpr int n = BLK::n, m = BLK::m, NZ = n + m;
    if constexpr (T::PG2) {
        // double integrator (A = [0 I; 0 0], B = [0; diag(beta)]): M = I + dt/2 A and Gam = 2 M (dt/2 B) are known in closed
        // form -- formed from dt and the model constants (SGPRs) with exactly the operations linearize() performs on
        // them, so the values are bit-identical to the stored block and ~18 wave-uniform global loads per call
        // (eight calls per interior point iteration) disappear
        constexpr int h3 = n / 2;
        const double h = 0.5 * K.dt;
        double Bd[n * m];
        Dyn<BLK::MODEL_ID>::B(K.P.mp, Bd);
#pragma unroll
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int j = 0; j < n; j++) M[i * n + j] = (i == j) ? 1.0 : ((j == i + h3) ? h : 0.0);
#pragma unroll
            for (int j = 0; j < m; j++) {
                const double hb = h * Bd[(j + h3) * m + j];
                Gam[i * m + j] = (i == j) ? 2.0 * (h * hb) : ((i == j + h3) ? 2.0 * hb : 0.0);
            }
        }
        return;
    }
    if constexpr (!T::LTI && n > 8 && T::NDEF == 0) {
        // recomputed with the operations of linearize() (and the round trip M -> Phi = 2 M - I -> M of the stored block),
        // so the values are those of the record
        double Mx[n * n], Bx[n * m];
        const double h = 0.5 * K.dt;
        stage_M<BLK::MODEL_ID>(K.P.mp, K.Xp + k * n, K.Up + k * m, h, Mx, Bx);
#pragma unroll
        for (int i = 0; i < n; i++) {
#pragma unroll
            for (int j = 0; j < n; j++)
                M[i * n + j] = T::Mnz(i, j) ? 0.5 * ((2.0 * Mx[i * n + j] - (i == j ? 1.0 : 0.0)) + (i == j ? 1.0 : 0.0)) : 0.0;
#pragma unroll
            for (int j = 0; j < m; j++) {
                double s = 0;
#pragma unroll
                for (int l = 0; l < n; l++) if (T::Bnz(l, j)) s += Mx[i * n + l] * (h * Bx[l * m + j]);   // (as linearize())
                Gam[i * m + j] = T::Gnz(i, j) ? 2.0 * s : 0.0;
            }
        }
        return;
    }
    const auto pg = K.PGk(k);
#pragma unroll
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < n; j++) M[i * n + j] = T::Mnz(i, j) ? 0.5 * (pg[i * NZ + j] + (i == j ? 1.0 : 0.0)) : 0.0;
#pragma unroll
        for (int j = 0; j < m; j++) Gam[i * m + j] = T::Gnz(i, j) ? pg[i * NZ + n + j] : 0.0;
    }
}

// initialize_model_params!/update_model_params! (freeflyer_se2.jl:116-147): linearise at (Xp,Up); also evaluates
// the signed distances of the linearisation point and freezes which obstacle rows are active this trip.
template <int MODEL, class BLK> GD void linearize(BLK& K, double toggle) {
    using T = MT<MODEL>;
    using R = Rec<MODEL>;
    constexpr int n = T::n, m = T::m, NZ = n + m;
    const int k = K.tid;
    if (k < K.N) {
        const double* xp = K.Xp + k * n;
        const double* up = K.Up + k * m;
        if (!T::LTI || k == 0) {
            double M[n * n], B[n * m];
            const double h = 0.5 * K.dt;
            stage_M<MODEL>(K.P.mp, xp, up, h, M, B);
            auto pg = K.PGk(T::LTI ? 0 : k);
            const bool knot0 = !T::LTI && k == 0;   // x_1 is pinned: the sweep's operand of knot 0 is [0 | b_0], b_0 = dt/2 B
            // (matrix-core models, one wave: also the compact record of the structural nonzeros, common.hpp: SpPG)
            constexpr bool SPR = SpPG<MODEL>::USE && BLK::ONE;
            using SP = SpPG<MODEL>;
            auto ps = K.PGS + (size_t)(SPR ? k : 0) * SP::S;
#pragma unroll
            for (int i = 0; i < n; i++) {
#pragma unroll
                for (int j = 0; j < n; j++) {
                    const double v = knot0 ? 0.0 : 2.0 * M[i * n + j] - (i == j ? 1.0 : 0.0);
                    pg[i * NZ + j] = v;
                    if constexpr (SPR) if (T::Mnz(i, j)) ps[SP::pos_phi(i, j)] = v;
                }
#pragma unroll
                for (int j = 0; j < m; j++) {
                    double s = 0;
#pragma unroll
                    for (int l = 0; l < n; l++) if (T::Bnz(l, j)) s += M[i * n + l] * (h * B[l * m + j]);   // (B's structural zeros add nothing: n m (n - 1) fewer FMAs for the 12/13-state models, where a column of B has one entry)
                    const double v = knot0 ? h * B[i * m + j] : 2.0 * s;
                    pg[i * NZ + n + j] = v;
                    if constexpr (SPR) if (T::Gnz(i, j)) ps[SP::pos_gam(i, j)] = v;
                }
                if constexpr (T::NDEF > 0) {   // a defect moves y_k directly: Gam_d = I (common.hpp)
#pragma unroll
                    for (int j = 0; j < T::NDEF; j++) pg[i * NZ + n + (m - T::NDEF) + j] = (i == j) ? 1.0 : 0.0;
                }
            }
        }
        uint64_t mask = 0;
        if constexpr (T::HAS_OBS) {
            double xw[T::WS];
#pragma unroll
            for (int j = 0; j < T::WS; j++) xw[j] = xp[j];
            const Env E = K.env();
            for (int i = 0; i < E.n_obs; i++) {
                double nh[T::WS];
                const double dist = signed_distance<T::WS>(K.P, E, 0, xw, i, nh);
                if (dist < toggle) {
                    mask |= (uint64_t)1 << i;
                    double c0 = K.P.mp.clearance - dist;
#pragma unroll
                    for (int j = 0; j < T::WS; j++) {
                        K.obs_nh[((size_t)i * T::WS + j) * K.N + k] = nh[j];
                        c0 += nh[j] * xw[j];
                    }
                    K.obs_c0[(size_t)i * K.N + k] = c0;
                }
            }
        }
        K.obs_mask[k] = mask;
    }
    K.sync();
}

// ---- Riccati factorisation of the condensed KKT system (cooperative, sequential in k) ---------------
// All goal-multiplier blocks (Pi, Z, V, D, Gd) have n columns in state-index space; column i is identically
// zero when coordinate i has no point goal.
template <int MODEL, class BLK> GD void factor_sweep_mw(BLK& K, double* fail, Prof& pf) {
    using T = MT<MODEL>;
    using R = Rec<MODEL>;
    constexpr int n = T::n, m = T::m, NZ = n + m, NQ = NZ * (NZ + 1) / 2, NPG = n * NZ;
    constexpr int QPT = (NQ + 63) / 64, PPT = (NPG + 63) / 64;
    const int tid = K.tid, NT = K.nt(), N = K.N;
    for (int e = tid; e < n * n; e += NT) { K.sP[e] = 0; K.sPi[e] = 0; K.sGd[e] = 0; }
    // operands of knot N-1
    double qq[QPT], pgn[PPT];
#pragma unroll
    for (int r = 0; r < QPT; r++) { const int e = tid + r * NT; qq[r] = (e < NQ) ? K.QQ[(size_t)(N - 1) * R::SQQ + e] : 0.0; }
#pragma unroll
    for (int r = 0; r < PPT; r++) pgn[r] = 0.0;
    {
        const auto pg = K.PGk(N - 1);
        for (int e = tid; e < NPG; e += NT) K.sPG[((N - 1) & 1) * NPG + e] = pg[e];
    }
    K.sync();
    for (int k = N - 1; k >= 0; k--) {
        const double* PGs = K.sPG + (k & 1) * NPG;
        // prefetch the operands of knot k-1 while this knot is processed
        double qqn[QPT];
#pragma unroll
        for (int r = 0; r < QPT; r++) {
            const int e = tid + r * NT;
            qqn[r] = (k > 0 && e < NQ) ? K.QQ[(size_t)(k - 1) * R::SQQ + e] : 0.0;
        }
        if (!T::LTI && k > 1) {
            const auto pg = K.PGk(k - 1);
#pragma unroll
            for (int r = 0; r < PPT; r++) { const int e = tid + r * NT; pgn[r] = (e < NPG) ? pg[e] : 0.0; }
        }
        pf.tick(PF_FPRE);
        // value function after knot k
        for (int e = tid; e < n * n; e += NT) {
            K.Paft[(size_t)k * R::SNN + e] = K.sP[e];
            K.Piaft[(size_t)k * R::SNN + e] = K.sPi[e];
        }
        // phase 1: T = P [Phi Gam],  Z = [Phi Gam]^T Pi (+ E at the last knot)
        // (operands are fetched as a batch, then pinned with a scheduling barrier: left alone, the compiler
        //  interleaves LDS reads and FMAs pairwise and pays the LDS latency n/2 times per dot product)
        for (int e = tid; e < NPG + NZ * n; e += NT) {
            double a[n], bb[n];
            double add = 0.0;
            if (e < NPG) {
                const int i = e / NZ, j = e % NZ;
#pragma unroll
                for (int l = 0; l < n; l++) { a[l] = K.sP[i * n + l]; bb[l] = PGs[l * NZ + j]; }
            } else {
                const int e2 = e - NPG, j = e2 / n, g = e2 % n;
#pragma unroll
                for (int l = 0; l < n; l++) { a[l] = PGs[l * NZ + j]; bb[l] = K.sPi[l * n + g]; }
                // E = [M^T C^T; b^T M^T C^T], M = (Phi + I)/2, M b = Gam/2; column g only for goal coordinates
                // (the u part of E is b^T M^T C^T = Gam^T / 2 for a model control; a defect control of the TrajOpt variants does not
                //  move x_N at all -- b_d = 0 although Gam_d = I)
                if (k == N - 1 && K.is_goal(g) && !(T::NDEF > 0 && j >= NZ - T::NDEF))
                    add = 0.5 * (PGs[g * NZ + j] + ((j == g) ? 1.0 : 0.0));
            }
            __builtin_amdgcn_sched_barrier(0);
            double s = 0;
#pragma unroll
            for (int l = 0; l < n; l++) s += a[l] * bb[l];
            s += add;
            if (e < NPG) K.sT[e] = s; else K.sZ[e - NPG] = s;
        }
        for (int e = tid; e < 2 * n; e += NT) {  // r_k = P_k c_k and Pi_k^T c_k
            const bool isr = e < n;
            const int i = isr ? e : e - n;
            double s = 0;
#pragma unroll
            for (int l = 0; l < n; l++) s += (isr ? K.sP[i * n + l] : K.sPi[l * n + i]) * K.cv[k * n + l];
            (isr ? K.rv : K.nun)[k * n + i] = s;
        }
        K.sync();
        pf.tick(PF_FAB);
        // phase 2: Hh = QQ + [Phi Gam]^T T (one triangle, mirrored)
#pragma unroll
        for (int r = 0; r < QPT; r++) {
            const int e = tid + r * NT;
            if (e < NQ) {
                const int ij = K.lut[e], i = ij >> 8, j = ij & 255;
                double a[n], bb[n];
#pragma unroll
                for (int l = 0; l < n; l++) { a[l] = PGs[l * NZ + i]; bb[l] = K.sT[l * NZ + j]; }
                __builtin_amdgcn_sched_barrier(0);
                double s = qq[r];
#pragma unroll
                for (int l = 0; l < n; l++) s += a[l] * bb[l];
                K.sHh[i * NZ + j] = s;
                K.sHh[j * NZ + i] = s;
            }
        }
        K.sync();
        pf.tick(PF_F1);
        // phase 3: block Cholesky of [S Hyu^T; Hyu Hyy]: L = chol(S), W = L^-1 Hyu^T, V = L^-1 Zu,
        // K = L^-T W, D = L^-T V.  One thread per column; the m x m factor is recomputed by each of them.
        if constexpr (m >= GUSTO_COOP_CHOL_MIN) {
            // Large control blocks (the TrajOpt variants: m = 9, 18 with the defect variables): one Cholesky for the workgroup,
            // in place in the uu block of Hh (left-looking, a lane per row, one synchronisation per column), then a lane per
            // right-hand side -- the n columns of Hyu^T, the n of Zu and the m unit vectors for S^-1 -- runs the two
            // triangular solves with L read from LDS.  With every lane factoring the block in registers (below: fine for
            // m <= 6) the 9 x 9 case held 243 doubles per lane, spilled, and was 45 % of a KKT solve.
            static_assert(2 * n + m <= 64 && m * m <= n * NZ, "a lane per right-hand side; r in the T buffer");
            auto Lu = [&](int i, int j) -> decltype(auto) { return (K.sHh[(n + i) * NZ + n + j]); };
            auto rinv = K.sT;      // 1 / L(i, i)   (the T buffer is free after phase 2)
            // lane i < m keeps row i of L in registers; column by column it reads row j (final, written by lane j) as a
            // broadcast from LDS, forms its entry and the pivot (every lane the pivot itself: j more FMAs, one
            // synchronisation per column less), scales and publishes its entry
            {
                const int i = tid < m ? tid : m - 1;
                double row[m];
#pragma unroll
                for (int l = 0; l < m; l++) row[l] = Lu(i, l);
                static_for<0, m>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    double rj[j > 0 ? j : 1];
#pragma unroll
                    for (int l = 0; l < j; l++) rj[l] = Lu(j, l);
                    double d = Lu(j, j), sacc = row[j];   // (the diagonal keeps S(j, j): nothing reads L(j, j) itself)
#pragma unroll
                    for (int l = 0; l < j; l++) { d -= rj[l] * rj[l]; sacc -= row[l] * rj[l]; }
                    const double r = rsqrt_nr(d);
                    if (!(d > 0.0)) *fail = 1.0;
                    row[j] = sacc * r;
                    if (tid > j && tid < m) Lu(tid, j) = row[j];
                    if (tid == j) rinv[j] = r;
                    K.sync();
                });
            }
            if (tid < 2 * n + m) {
                const int c = tid;
                const bool isK = c < n, isD = c >= n && c < 2 * n;
                const int g = isK ? c : (isD ? c - n : c - 2 * n);
                double w[m], kk[m];
#pragma unroll
                for (int l = 0; l < m; l++) w[l] = isK ? K.sHh[g * NZ + n + l] : (isD ? K.sZ[(n + l) * n + g] : ((l == g) ? 1.0 : 0.0));
#pragma unroll
                for (int i = 0; i < m; i++) {      // L w = col
                    double sacc = w[i];
#pragma unroll
                    for (int l = 0; l < i; l++) sacc -= Lu(i, l) * w[l];
                    w[i] = sacc * rinv[i];
                }
#pragma unroll
                for (int i = m - 1; i >= 0; i--) {  // L^T kk = w
                    double sacc = w[i];
#pragma unroll
                    for (int l = i + 1; l < m; l++) sacc -= Lu(l, i) * kk[l];
                    kk[i] = sacc * rinv[i];
                }
                if (isK || isD) {
                    double* sw = isK ? K.sW : K.sV;
                    double* sk = isK ? K.sK : K.sD;
                    double* gk = K.KD + (size_t)k * R::SKD + (isK ? R::oK : R::oD);
#pragma unroll
                    for (int i = 0; i < m; i++) { sw[i * n + g] = w[i]; sk[i * n + g] = kk[i]; gk[i * n + g] = kk[i]; }
                } else {   // column g of S^-1 = L^-T L^-1 (feed-forward only)
#pragma unroll
                    for (int i = 0; i < m; i++) K.KD[(size_t)k * R::SKD + ((R::S_TRI && i > g) ? R::KD_DUMMY : R::sS(i, g))] = kk[i];
                }
            }
        } else
        if (tid < 2 * n || tid < m * m) {
            double S[m * m], Li[m * m];
#pragma unroll
            for (int i = 0; i < m; i++)
#pragma unroll
                for (int j = 0; j < m; j++) S[i * m + j] = K.sHh[(n + i) * NZ + n + j];
            if (!chol_inv<m>(S, Li)) *fail = 1.0;
            for (int c = tid; c < 2 * n; c += NT) {
                double col[m], w[m], kk[m];
                const bool isK = c < n;
                const int g = isK ? c : c - n;
#pragma unroll
                for (int l = 0; l < m; l++) col[l] = isK ? K.sHh[g * NZ + n + l] : K.sZ[(n + l) * n + g];
#pragma unroll
                for (int i = 0; i < m; i++) {
                    double s = 0;
#pragma unroll
                    for (int l = 0; l <= i; l++) s += Li[i * m + l] * col[l];
                    w[i] = s;
                }
#pragma unroll
                for (int i = 0; i < m; i++) {
                    double s = 0;
#pragma unroll
                    for (int l = i; l < m; l++) s += Li[l * m + i] * w[l];
                    kk[i] = s;
                }
                double* sw = isK ? K.sW : K.sV;
                double* sk = isK ? K.sK : K.sD;
                double* gk = K.KD + (size_t)k * R::SKD + (isK ? R::oK : R::oD);
#pragma unroll
                for (int i = 0; i < m; i++) { sw[i * n + g] = w[i]; sk[i * n + g] = kk[i]; gk[i * n + g] = kk[i]; }
            }
            for (int e = tid; e < m * m; e += NT) {  // S^-1 = L^-T L^-1 (feed-forward only)
                const int i = e / m, j = e % m;
                double s = 0;
#pragma unroll
                for (int l = 0; l < m; l++) if (l >= i && l >= j) s += Li[l * m + i] * Li[l * m + j];
                K.KD[(size_t)k * R::SKD + R::oS + e] = s;
            }
        }
        K.sync();
        pf.tick(PF_F4);
        // phase 4: P' = Hyy - W^T W, Pi' = Zy - W^T V, Phicl = Phi - Gam K, Gd += V^T V.  The Schur complements are
        // never formed through an explicit S^-1: with barrier weights ~1/mu in Hyy that loses every digit.
        for (int e = tid; e < 4 * n * n; e += NT) {
            const int q = e / (n * n), e2 = e % (n * n), i = e2 / n, j = e2 % n;
            double a[m], bb[m], base;
            if (q == 0) {
                base = K.sHh[i * NZ + j];
#pragma unroll
                for (int l = 0; l < m; l++) { a[l] = -K.sW[l * n + i]; bb[l] = K.sW[l * n + j]; }
            } else if (q == 1) {
                base = PGs[i * NZ + j];
#pragma unroll
                for (int l = 0; l < m; l++) { a[l] = -PGs[i * NZ + n + l]; bb[l] = K.sK[l * n + j]; }
            } else if (q == 2) {
                base = K.sZ[i * n + j];
#pragma unroll
                for (int l = 0; l < m; l++) { a[l] = -K.sW[l * n + i]; bb[l] = K.sV[l * n + j]; }
            } else {
                base = K.sGd[e2];
#pragma unroll
                for (int l = 0; l < m; l++) { a[l] = K.sV[l * n + i]; bb[l] = K.sV[l * n + j]; }
            }
            __builtin_amdgcn_sched_barrier(0);
            double s = base;
#pragma unroll
            for (int l = 0; l < m; l++) s += a[l] * bb[l];
            if (q == 0) K.sP[e2] = s;
            else if (q == 1) K.Phicl[(size_t)k * R::SNN + e2] = s;
            else if (q == 2) K.sPi[e2] = s;
            else K.sGd[e2] = s;
        }
        // operands of the next knot: knot 0 has Phi = 0, Gam = b_0 (x_1 is pinned)
        if (k > 1) {
            if constexpr (T::LTI) {
                for (int e = tid; e < NPG; e += NT) K.sPG[((k - 1) & 1) * NPG + e] = PGs[e];
            } else {
#pragma unroll
                for (int r = 0; r < PPT; r++) { const int e = tid + r * NT; if (e < NPG) K.sPG[((k - 1) & 1) * NPG + e] = pgn[r]; }
            }
        } else if (k == 1) {
            double B[n * m];
            Dyn<MODEL>::B(K.P.mp, B);
            for (int e = tid; e < NPG; e += NT) {
                const int i = e / NZ, j = e % NZ;
                double v = 0.0;
#pragma unroll
                for (int q = 0; q < n * m; q++) if (j >= n && q == i * m + (j - n)) v = 0.5 * K.dt * B[q];
                if constexpr (T::NDEF > 0) { if (j >= n + (m - T::NDEF) && j - n - (m - T::NDEF) == i) v = 1.0; }
                K.sPG[0 * NPG + e] = v;
            }
        }
#pragma unroll
        for (int r = 0; r < QPT; r++) qq[r] = qqn[r];
        pf.tick(PF_F7);
        K.sync();
        pf.tick(PF_FCD);
    }
}

// factor_sweep_mw for a problem of ONE wave whose control block is large (the TrajOpt variants with N <= 64: m = 9 / 18 / 19 with
// the defect controls, m > n).  The same stage -- the same sums in the same order, bit-identical results -- laid out for 64
// lanes at compile time:
//  * every round of an entry-per-lane product is unrolled, its LDS operands requested as a batch; the four Schur-type
//    updates of phase 4 (P', Phicl, Pi', Gd) are formed by the lane of entry (i, j) from shared operands instead of four
//    divergent passes over 4 n^2 entries;
//  * the m x m Cholesky runs without LDS: lane i keeps row i of L in registers, column j takes row j from lane j by v_readlane
//    (every lane forms the pivot itself: no synchronisation per column);
//  * no global store under a branch (lanes without an entry aim at the padding slot of the record): the wait of a stage
//    counts its stores instead of draining them.
// Measured (freeflyerSE2 TrajOpt, B = 1024): the generic stage 12.9 k cycles, 49 % of a KKT solve (profiles/r05_trajopt_stage.txt).
template <int MODEL, class BLK> GD void factor_sweep_w1(BLK& K, double* fail, Prof& pf) {
    using T = MT<MODEL>;
    using R = Rec<MODEL>;
    constexpr int n = T::n, m = T::m, NZ = n + m, NQ = NZ * (NZ + 1) / 2, NPG = n * NZ, NN = n * n, NZN = NZ * n;
    constexpr int RT = (NPG + 63) / 64, RQ = (NQ + 63) / 64, RN = (NN + 63) / 64, RZ = (NZN + 63) / 64;
    constexpr bool SMALL = n <= 8;   // (operands of all rounds of a phase in flight at once; the large models go round by round)
    static_assert(2 * n + m <= 64 && R::SQQ == 64 * RQ && R::SNN > NN && R::SKD > R::KD_DUMMY, "a lane per right-hand side; padded records");
    const int tid = K.tid, N = K.N;
    int hI[RQ], hJ[RQ];
#pragma unroll
    for (int r = 0; r < RQ; r++) { const int e = tid + 64 * r, ij = K.lut[e < NQ ? e : 0]; hI[r] = ij >> 8; hJ[r] = ij & 255; }
#pragma unroll
    for (int r = 0; r < RN; r++) { const int e = tid + 64 * r; if (e < NN) { K.sP[e] = 0; K.sPi[e] = 0; K.sGd[e] = 0; } }
    double qq[RQ], pgn[RT];
#pragma unroll
    for (int r = 0; r < RQ; r++) qq[r] = K.QQ[(size_t)(N - 1) * R::SQQ + tid + 64 * r];   // (padded record: every lane has an entry)
#pragma unroll
    for (int r = 0; r < RT; r++) pgn[r] = 0.0;
    {
        const auto pg = K.PGk(N - 1);
#pragma unroll
        for (int r = 0; r < RT; r++) { const int e = tid + 64 * r; if (e < NPG) K.sPG[((N - 1) & 1) * NPG + e] = pg[e]; }
    }
    bool okall = true;
    K.sync();
    for (int k = N - 1; k >= 0; k--) {
        const double* PGs = K.sPG + (k & 1) * NPG;
        // (large models: the entry indices of the rounds -- e / NZ, e % NZ, the table look-ups, dozens of LDS addresses -- are
        // formed again in every stage from a lane id the compiler cannot see through; hoisted out of the knot loop, as it
        // does unasked, they do not fit the register file and the stage makes ~300 scratch accesses: 652 against 465 ms per
        // astrobeeSE3 TrajOpt batch of 256)
        int tl = tid;
        if constexpr (!SMALL) asm volatile("" : "+v"(tl));
        // operands of knot k-1, in flight over the stage (clamped, unconditional)
        double qqn[RQ];
#pragma unroll
        for (int r = 0; r < RQ; r++) qqn[r] = K.QQ[(size_t)((k > 0) ? k - 1 : 0) * R::SQQ + tl + 64 * r];
        if constexpr (!T::LTI) {
            const auto pg = K.PGk((k > 1) ? k - 1 : 1);
#pragma unroll
            for (int r = 0; r < RT; r++) { const int e = tl + 64 * r; pgn[r] = pg[(e < NPG) ? e : NPG - 1]; }
        }
        // value function after knot k
#pragma unroll
        for (int r = 0; r < RN; r++) {
            const int e = tl + 64 * r, ei = (e < NN) ? e : 0, eo = (e < NN) ? e : NN;
            K.Paft[(size_t)k * R::SNN + eo] = K.sP[ei];
            K.Piaft[(size_t)k * R::SNN + eo] = K.sPi[ei];
        }
        pf.tick(PF_FPRE);
        // ---- phase 1: T = P [Phi Gam], Z = [Phi Gam]^T Pi (+ E at the last knot), r_k = P_k c_k, Pi_k^T c_k --------------
        auto round_T = [&](int r, double* a, double* bb) {
            const int e = tl + 64 * r, i = (e < NPG) ? e / NZ : 0, j = (e < NPG) ? e % NZ : 0;
#pragma unroll
            for (int l = 0; l < n; l++) { a[l] = K.sP[i * n + l]; bb[l] = PGs[l * NZ + j]; }
        };
        auto round_Z = [&](int r, double* a, double* bb, double& add) {
            const int e2 = tl + 64 * r, j = (e2 < NZN) ? e2 / n : 0, g = (e2 < NZN) ? e2 % n : 0;
#pragma unroll
            for (int l = 0; l < n; l++) { a[l] = PGs[l * NZ + j]; bb[l] = K.sPi[l * n + g]; }
            // E = [M^T C^T; b^T M^T C^T] (factor_sweep_mw): column g only for goal coordinates, no entry for a defect control
            const double ev = 0.5 * (PGs[g * NZ + j] + ((j == g) ? 1.0 : 0.0));
            add = (k == N - 1 && K.is_goal(g) && !(T::NDEF > 0 && j >= NZ - T::NDEF)) ? ev : 0.0;
        };
        auto dot = [&](const double* a, const double* bb, double s) {
#pragma unroll
            for (int l = 0; l < n; l++) s += a[l] * bb[l];
            return s;
        };
        if constexpr (SMALL) {
            double ta[RT][n], tb[RT][n], za[RZ][n], zb[RZ][n], zadd[RZ], ra[n], rb[n];
            const bool isr = tl < n;
            const int ri = isr ? tl : ((tl < 2 * n) ? tl - n : 0);
#pragma unroll
            for (int r = 0; r < RT; r++) round_T(r, ta[r], tb[r]);
#pragma unroll
            for (int r = 0; r < RZ; r++) round_Z(r, za[r], zb[r], zadd[r]);
#pragma unroll
            for (int l = 0; l < n; l++) { ra[l] = *(isr ? K.sP + ri * n + l : K.sPi + l * n + ri); rb[l] = K.cv[k * n + l]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < RT; r++) { const double t = dot(ta[r], tb[r], 0.0); if (tl + 64 * r < NPG) K.sT[tl + 64 * r] = t; }
#pragma unroll
            for (int r = 0; r < RZ; r++) { const double z = dot(za[r], zb[r], 0.0) + zadd[r]; if (tl + 64 * r < NZN) K.sZ[tl + 64 * r] = z; }
            { const double rr = dot(ra, rb, 0.0); if (tl < 2 * n) (isr ? K.rv : K.nun)[k * n + ri] = rr; }
        } else {
#pragma unroll
            for (int r = 0; r < RT; r++) {
                double a[n], bb[n];
                round_T(r, a, bb);
                __builtin_amdgcn_sched_barrier(0);
                const double t = dot(a, bb, 0.0);
                if (tl + 64 * r < NPG) K.sT[tl + 64 * r] = t;
            }
#pragma unroll
            for (int r = 0; r < RZ; r++) {
                double a[n], bb[n], add;
                round_Z(r, a, bb, add);
                __builtin_amdgcn_sched_barrier(0);
                const double z = dot(a, bb, 0.0) + add;
                if (tl + 64 * r < NZN) K.sZ[tl + 64 * r] = z;
            }
            {
                const bool isr = tl < n;
                const int ri = isr ? tl : ((tl < 2 * n) ? tl - n : 0);
                double ra[n], rb[n];
#pragma unroll
                for (int l = 0; l < n; l++) { ra[l] = *(isr ? K.sP + ri * n + l : K.sPi + l * n + ri); rb[l] = K.cv[k * n + l]; }
                const double rr = dot(ra, rb, 0.0);
                if (tl < 2 * n) (isr ? K.rv : K.nun)[k * n + ri] = rr;
            }
        }
        K.sync();
        pf.tick(PF_FAB);
        // ---- phase 2: Hh = QQ + [Phi Gam]^T T (one triangle, mirrored) ----------------------------------------------------
        auto round_H = [&](int r, double* a, double* bb) {
#pragma unroll
            for (int l = 0; l < n; l++) { a[l] = PGs[l * NZ + hI[r]]; bb[l] = K.sT[l * NZ + hJ[r]]; }
        };
        if constexpr (!SMALL) {
#pragma unroll
            for (int r = 0; r < RQ; r++) { const int e = tl + 64 * r, ij = K.lut[e < NQ ? e : 0]; hI[r] = ij >> 8; hJ[r] = ij & 255; }
        }
        if constexpr (SMALL) {
            double ha[RQ][n], hb[RQ][n];
#pragma unroll
            for (int r = 0; r < RQ; r++) round_H(r, ha[r], hb[r]);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < RQ; r++) {
                const double h = dot(ha[r], hb[r], qq[r]);
                if (tl + 64 * r < NQ) { K.sHh[hI[r] * NZ + hJ[r]] = h; K.sHh[hJ[r] * NZ + hI[r]] = h; }
            }
        } else {
#pragma unroll
            for (int r = 0; r < RQ; r++) {
                double a[n], bb[n];
                round_H(r, a, bb);
                __builtin_amdgcn_sched_barrier(0);
                const double h = dot(a, bb, qq[r]);
                if (tl + 64 * r < NQ) { K.sHh[hI[r] * NZ + hJ[r]] = h; K.sHh[hJ[r] * NZ + hI[r]] = h; }
            }
        }
        K.sync();
        pf.tick(PF_F1);
        // ---- phase 3: L = chol(S) (lane i = row i, row j by v_readlane), then a lane per right-hand side: W = L^-1 Hyu^T, V = L^-1 Zu,
        //      K = L^-T W, D = L^-T V, S^-1 = L^-T L^-1 ---------------------------------------------------------------------
        auto Lu = [&](int i, int j) -> decltype(auto) { return (K.sHh[(n + i) * NZ + n + j]); };
        double rinv[m];   // 1 / L(i, i): wave-uniform
        {
            const int i = tl < m ? tl : m - 1;
            double row[m];
#pragma unroll
            for (int l = 0; l < m; l++) row[l] = Lu(i, l);
            static_for<0, m>([&](auto J) {
                constexpr int j = decltype(J)::value;
                double rj[j > 0 ? j : 1];
#pragma unroll
                for (int l = 0; l < j; l++) rj[l] = readlane_f64(row[l], j);
                double d = readlane_f64(row[j], j), sacc = row[j];
#pragma unroll
                for (int l = 0; l < j; l++) { d -= rj[l] * rj[l]; sacc -= row[l] * rj[l]; }
                const double r = rsqrt_nr(d);
                okall = okall && (d > 0.0);
                row[j] = sacc * r;
                rinv[j] = r;
            });
#pragma unroll
            for (int l = 0; l < m - 1; l++) if (tl < m && tl > l) Lu(tl, l) = row[l];
        }
        K.sync();
        pf.tick(PF_F4);
        {
            const int c = tl < 2 * n + m ? tl : 0;
            const bool isK = c < n, isD = c >= n && c < 2 * n;
            const int g = isK ? c : (isD ? c - n : c - 2 * n);
            double w[m], kk[m];
#pragma unroll
            for (int l = 0; l < m; l++) w[l] = isK ? K.sHh[g * NZ + n + l] : (isD ? K.sZ[(n + l) * n + g] : ((l == g) ? 1.0 : 0.0));
#pragma unroll
            for (int i = 0; i < m; i++) {      // L w = col
                double sacc = w[i];
#pragma unroll
                for (int l = 0; l < i; l++) sacc -= Lu(i, l) * w[l];
                w[i] = sacc * rinv[i];
            }
#pragma unroll
            for (int i = m - 1; i >= 0; i--) {  // L^T kk = w
                double sacc = w[i];
#pragma unroll
                for (int l = i + 1; l < m; l++) sacc -= Lu(l, i) * kk[l];
                kk[i] = sacc * rinv[i];
            }
            const bool rhs = tl < 2 * n + m, kd = rhs && (isK || isD);
            double* sw = isK ? K.sW : K.sV;
            double* sk = isK ? K.sK : K.sD;
            // (one unconditional store per entry: K and D at oK / oD + i n + g, column g of S^-1 at oS + i m + g, idle lanes in the padding)
            const bool isS = rhs && !isK && !isD;
            const int gbase = !rhs ? R::KD_DUMMY : (isK ? R::oK + g : (isD ? R::oD + g : R::oS + g));
            const int gstr = !rhs ? 0 : ((isK || isD) ? n : m);
#pragma unroll
            for (int i = 0; i < m; i++) {
                if (kd) { sw[i * n + g] = w[i]; sk[i * n + g] = kk[i]; }
                // (S^-1 as its upper triangle for the TrajOpt variants, Rec::S_TRI: the lane of column g stores rows i <= g)
                const int gi = (R::S_TRI && isS) ? ((i <= g) ? R::oS + (i * m - i * (i - 1) / 2 - i) + g : R::KD_DUMMY) : gbase + i * gstr;
                K.KD[(size_t)k * R::SKD + gi] = kk[i];
            }
        }
        K.sync();
        pf.tick(PF_F6);
        // ---- phase 4: P' = Hyy - W^T W, Phicl = Phi - Gam K, Pi' = Zy - W^T V, Gd += V^T V: the lane of entry (i, j) all four ----
#pragma unroll
        for (int r = 0; r < RN; r++) {
            const int e2 = tl + 64 * r;
            const bool on = e2 < NN;
            const int i = on ? e2 / n : 0, j = on ? e2 % n : 0;
            double wi[m], wj[m], vi[m], vj[m], gi[m], kj[m];
#pragma unroll
            for (int l = 0; l < m; l++) {
                wi[l] = K.sW[l * n + i]; wj[l] = K.sW[l * n + j]; vi[l] = K.sV[l * n + i]; vj[l] = K.sV[l * n + j];
                gi[l] = PGs[i * NZ + n + l]; kj[l] = K.sK[l * n + j];
            }
            double pn = K.sHh[i * NZ + j], ph = PGs[i * NZ + j], pin = K.sZ[i * n + j], gd = K.sGd[on ? e2 : 0];
            if constexpr (SMALL) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int l = 0; l < m; l++) pn -= wi[l] * wj[l];
#pragma unroll
            for (int l = 0; l < m; l++) ph -= gi[l] * kj[l];
#pragma unroll
            for (int l = 0; l < m; l++) pin -= wi[l] * vj[l];
#pragma unroll
            for (int l = 0; l < m; l++) gd += vi[l] * vj[l];
            if (on) { K.sP[e2] = pn; K.sPi[e2] = pin; K.sGd[e2] = gd; }
            K.Phicl[(size_t)k * R::SNN + (on ? e2 : NN)] = ph;
        }
        // operands of the next knot: knot 0 has Phi = 0, Gam = b_0 (x_1 is pinned)
        if (k > 1) {
#pragma unroll
            for (int r = 0; r < RT; r++) { const int e = tl + 64 * r; if (e < NPG) K.sPG[((k - 1) & 1) * NPG + e] = T::LTI ? PGs[e] : pgn[r]; }
        } else if (k == 1) {
            double B[n * m];
            Dyn<MODEL>::B(K.P.mp, B);
#pragma unroll
            for (int r = 0; r < RT; r++) {
                const int e = tl + 64 * r;
                if (e < NPG) {
                    const int i = e / NZ, j = e % NZ;
                    double v = 0.0;
#pragma unroll
                    for (int q = 0; q < n * m; q++) if (j >= n && q == i * m + (j - n)) v = 0.5 * K.dt * B[q];
                    if constexpr (T::NDEF > 0) { if (j >= n + (m - T::NDEF) && j - n - (m - T::NDEF) == i) v = 1.0; }
                    K.sPG[0 * NPG + e] = v;
                }
            }
        }
#pragma unroll
        for (int r = 0; r < RQ; r++) qq[r] = qqn[r];
        pf.tick(PF_F7);
        K.sync();
        pf.tick(PF_FCD);
    }
    if (!okall) *fail = 1.0;
}

// p_{k-1} = Phicl_k^T (p_k + r_k) + qt_k, k = N-1..1; pv[k] holds qt_k on entry and p_k on exit
template <class BLK> GD void backward_sweep_mw(BLK& K) {
    // pt_{k-1} = Phicl_k^T pt_k + qq_k (pt = p + r, qq_k = qt_k + r_{k-1}); pv[k]: qq_k on entry, pt_k on exit
    constexpr int n = BLK::n;
    using R = typename BLK::R;
    const int tid = K.tid, N = K.N;
    double p = 0.0, col[n], coln[n];
#pragma unroll
    for (int l = 0; l < n; l++) { col[l] = 0; coln[l] = 0; }
    if (tid < n) {
#pragma unroll
        for (int l = 0; l < n; l++) col[l] = K.Phicl[(size_t)(N - 1) * R::SNN + l * n + tid];
        p = K.rv[(N - 1) * n + tid];
    }
    for (int k = N - 1; k >= 1; k--) {
        double* buf = K.sT + (k & 1) * n;
        if (tid < n) {
            if (k > 1) {
#pragma unroll
                for (int l = 0; l < n; l++) coln[l] = K.Phicl[(size_t)(k - 1) * R::SNN + l * n + tid];
            }
            buf[tid] = p;
        }
        K.sync();
        if (tid < n) {
            double s = K.pv[k * n + tid], v[n];
#pragma unroll
            for (int l = 0; l < n; l++) v[l] = buf[l];
            __builtin_amdgcn_sched_barrier(0);
            K.pv[k * n + tid] = p;
#pragma unroll
            for (int l = 0; l < n; l++) s += col[l] * v[l];
            p = s;
#pragma unroll
            for (int l = 0; l < n; l++) col[l] = coln[l];
        }
    }
    if (tid < n) K.pv[tid] = p;
    K.sync();
}

// dy_k = Phicl_k dy_{k-1} + ct_k, k = 0..N-1; dY[k] holds ct_k on entry and dy_k on exit
template <class BLK> GD void forward_sweep_mw(BLK& K) {
    constexpr int n = BLK::n;
    using R = typename BLK::R;
    const int tid = K.tid, N = K.N;
    double y = 0.0, row[n], rown[n];
#pragma unroll
    for (int l = 0; l < n; l++) { row[l] = 0; rown[l] = 0; }
    if (tid < n) {
#pragma unroll
        for (int l = 0; l < n; l++) row[l] = K.Phicl[(size_t)tid * n + l];
    }
    for (int k = 0; k < N; k++) {
        double* buf = K.sT + (k & 1) * n;
        if (tid < n) {
            if (k + 1 < N) {
#pragma unroll
                for (int l = 0; l < n; l++) rown[l] = K.Phicl[(size_t)(k + 1) * R::SNN + tid * n + l];
            }
            buf[tid] = y;
        }
        K.sync();
        if (tid < n) {
            double s = K.dY[k * n + tid], v[n];
#pragma unroll
            for (int l = 0; l < n; l++) v[l] = buf[l];
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int l = 0; l < n; l++) s += row[l] * v[l];
            y = s;
            K.dY[k * n + tid] = y;
#pragma unroll
            for (int l = 0; l < n; l++) row[l] = rown[l];
        }
    }
    K.sync();
}


// Compact by-value view of a problem for the one-wave sweeps (only what their knot loops touch).  Whether a sweep runs
// inlined or as a real call is a per-model choice (MT::SWEEP_CALL): for freeflyerSE2 the call costs 40 % (its frame
// goes to scratch), for the 12/13-state models it is what keeps the knot loop out of scratch.
template <int MODEL> struct SweepView {
    using T = MT<MODEL>;
    using C = LdsC<MODEL, true>;
    using R = Rec<MODEL>;
    static constexpr int n = T::n, m = T::m, NZ = n + m;
    static constexpr bool ONE = true;
    static constexpr int MODEL_ID = MODEL;
    static constexpr int SPH = C::PHICL_LDS ? n * n : R::SNN;   // stride of Phicl records (LDS copy is unpadded)
    double* lds;
    LPtr<double> sP, sPi, sPG, sT, sHh, sZ, sK, sD, sW, sV, sGd;
    LPtr<int> lut;
    LPtr<double> cv, rv, nun, pv, dY;
    GPtr<double> PG, PGS, QQ, Paft, Piaft, KD;
    std::conditional_t<C::PHICL_LDS, LPtr<double>, GPtr<double>> Phicl;
    LPtr<double> kdl;   // K | D | S^-1 per knot in LDS (LdsC::KD_LDS; the double integrator's sweeps rebuild Phicl from it)
    LPtr<double> pgl;   // [Phi Gam] per knot in LDS (LdsC::PG_LDS)
    int pg_off, seg_off;
    const gusto_model_params* mpp;
    struct PW { const gusto_model_params& mp; } ;
    int tid, N;
    double dt;
    unsigned goalmask;
    GD void sync() const { blk_sync<true>(); }
    GD auto PGk(int k) const {
        if constexpr (C::PG_LDS) return pgl + k * n * NZ;
        else return PG + (size_t)(T::LTI ? 0 : k) * n * NZ;
    }
    // the P | Pi record after knot k (k >= -1) and its dummy slot
    GD auto pprec(int k) const {
        return Paft + (size_t)k * R::SNN;
    }
    static constexpr int PP_DUMMY = R::SNN - 1;
    GD bool is_goal(int i) const { return (goalmask >> i) & 1u; }
    int phicl_off, kd_off;   // LdsLayout::phicl, ::kd
    GD void rebind_lds(double* l) {   // see Blk::rebind_lds
        lds = l;
        sP = lds + C::sP; sPi = lds + C::sPi; sPG = lds + C::sPG; sT = lds + C::sT; sHh = lds + C::sHh;
        sZ = lds + C::sZ; sK = lds + C::sK; sD = lds + C::sD; sW = lds + C::sW; sV = lds + C::sV;
        sGd = lds + C::sGd; lut = reinterpret_cast<int*>(lds + C::lut);
        double* v = lds + C::vecs;
        dY = v + N * n; pv = v + 2 * N * n; cv = v + 3 * N * n; rv = v + 4 * N * n; nun = v + 6 * N * n;
        // (one-wave problems of the small models: Phicl lives in LDS, stride n*n; otherwise padded global records)
        if constexpr (C::PHICL_LDS) Phicl = lds + phicl_off;
        if constexpr (C::KD_LDS) kdl = lds + kd_off;
        if constexpr (C::PG_LDS) pgl = lds + pg_off;
    }
    GD void rebind_global() {
        PG = al16((double*)PG); QQ = al16((double*)QQ); Paft = al16((double*)Paft); Piaft = al16((double*)Piaft); KD = al16((double*)KD);
    }
    template <class BLK> GD static SweepView make(const BLK& K) {
        SweepView v;
        v.N = K.N; v.phicl_off = K.P.ll.phicl; v.kd_off = K.P.ll.kd; v.pg_off = K.P.ll.pg; v.seg_off = K.P.ll.seg;
        if constexpr (!C::PHICL_LDS) v.Phicl = K.Phicl;   // (LDS copy of the small models: set by rebind_lds)
        v.rebind_lds(K.lds);
        v.PG = K.PG; v.PGS = K.PGS; v.QQ = K.QQ; v.Paft = K.Paft; v.Piaft = K.Piaft; v.KD = K.KD;
        v.mpp = &K.P.mp; v.tid = K.tid; v.dt = K.dt; v.goalmask = K.goalmask;
        return v;
    }
};

}  // namespace gusto
#include "factor1w.hpp"   // factor_sweep_1w / _pg2 / _mfma and the costate-form switches
namespace gusto {

template <class BLKA> GD void backward_sweep_1w(BLKA K) {
    using BLK = std::remove_cv_t<std::remove_reference_t<BLKA>>;   // (a SweepView by value, or a Blk by reference)
    constexpr int n = BLK::n, C = 64 / n, PS = n > 8 ? 4 : 1;
    // 12/13-state models: the n-vector goes from group to group through 64 doubles of LDS (one ds_write per lane, broadcast
    // ds_reads at compile-time addresses) instead of 2 n v_readlane per knot
    constexpr bool XL = n > 8;
    double* ex = K.sHh;
    using R = typename BLK::R;
    const int tid = K.tid, N = K.N;
    const int g = (tid < C * n) ? tid / n : C - 1, i = (tid < C * n) ? tid % n : 0;
    // (KD_LDS) the entries of Gam and Phi of the double integrator, formed as linearize() forms them
    double gl[n];
    auto phi_e = [&](int r_, int c_) { return (r_ == c_) ? 1.0 : ((c_ == r_ + n / 2) ? K.dt : 0.0); };
    if constexpr (BLK::C::PHI_FROM_K) {
        constexpr int m = BLK::m;
        double Bd[n * m];
        Dyn<BLK::MODEL_ID>::B(*K.mpp, Bd);
        const double h = 0.5 * K.dt;
#pragma unroll
        for (int l = 0; l < n; l++) {
            const int c_ = l % m;
            const double hb = h * Bd[(c_ + n / 2) * m + c_];
            gl[l] = (l < n / 2) ? 2.0 * (h * hb) : 2.0 * hb;
        }
    }
    double phc[n];   // column i of Phi
#pragma unroll
    for (int l = 0; l < n; l++) phc[l] = phi_e(l, i);
    double col[n], coln[n], qv, qvn = 0, pval;
    auto fetch = [&](int k0, double* c, double& q) {
        // (clamped, unconditional loads: a group past the end of the sweep never has its step executed -- the steps are
        // guarded wave-uniformly below -- so its operands only need to be loadable.  As `ok ? load : 0` every load sat
        // in a branch of its own.)
        const int kk = (k0 - g >= 1) ? k0 - g : 1;
        if constexpr (BLK::C::PHI_FROM_K) {
            // column i of Phicl = Phi - Gam K from the LDS copy of K: Gam has ONE entry per row (row l: column l mod m),
            // Phi = I + dt [0 I; 0 0]; the fma of the factor sweep (which formed Phi - Gam K with the zero terms too)
            constexpr int m = BLK::m;
            double kc[m];
#pragma unroll
            for (int a = 0; a < m; a++) kc[a] = K.kdl[kk * BLK::C::KDS + a * n + i];
#pragma unroll
            for (int l = 0; l < n; l++) c[l] = phc[l] - gl[l] * kc[l % m];
        } else if constexpr (n <= 4) {
            const bool ok = k0 - g >= 1;
#pragma unroll
            for (int l = 0; l < n; l++) c[l] = ok ? K.Phicl[(size_t)kk * BLK::SPH + l * n + i] : 0.0;
            q = ok ? K.pv[kk * n + i] : 0.0;
            return;
        } else {
#pragma unroll
            for (int l = 0; l < n; l++) c[l] = K.Phicl[(size_t)kk * BLK::SPH + l * n + i];
        }
        q = K.pv[kk * n + i];
    };
    // 12/13-state models: the operands come from the workspace in HBM (~3 k cycles away; 540 MB of slot workspaces do not
    // fit the caches), and one chunk of work (~1.2 k cycles) in front of a fetch does not cover that: a ring of RING chunk
    // buffers keeps RING - 1 fetches in flight.  (No buffer is ever copied: a copy waits for the newest fetch.)
    constexpr int RING = (XL && !BLK::C::KD_LDS) ? GUSTO_SWEEP_RING : 1;
    if constexpr (RING > 1) {
        double cb[RING][n], qb[RING];
#pragma unroll
        for (int d = 0; d < RING - 1; d++) fetch(N - 1 - d * C, cb[d], qb[d]);
        pval = K.rv[(N - 1) * n + i];
        ex[tid] = pval;
        K.sync();
        if (tid < n) K.pv[(N - 1) * n + tid] = pval;
        for (int kb = N - 1; kb >= 1; kb -= RING * C) {
            static_for<0, RING>([&](auto DD) {
                constexpr int d = decltype(DD)::value;
                const int k0 = kb - d * C;
                fetch(k0 - (RING - 1) * C, cb[(d + RING - 1) % RING], qb[(d + RING - 1) % RING]);   // (clamped: always loadable)
                if (k0 >= 1) {
#pragma unroll
                    for (int gs = 0; gs < C; gs++) {
                        if (k0 - gs >= 1) {
                            const int sg = (gs == 0) ? C - 1 : gs - 1;
                            double acc[PS];
#pragma unroll
                            for (int q = 0; q < PS; q++) acc[q] = (q == 0) ? qb[d] : 0.0;
                            double pb[n];
                            // (the n-vector by 2 n v_readlane into n scalar pairs of their own, all of them before the first
                            // FMA -- see the small-model path below -- instead of a ds_write / broadcast ds_read round trip)
#pragma unroll
                            for (int l = 0; l < n; l++) pb[l] = readlane_f64(pval, sg * n + l);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int l = 0; l < n; l++) acc[l % PS] += cb[d][l] * pb[l];
                            const double s = (PS == 4) ? (acc[0] + acc[1]) + (acc[2] + acc[PS - 1]) : acc[0];
                            pval = (g == gs) ? s : pval;
                        }
                    }
                    const int kk = k0 - g;
                    if (tid < C * n && kk >= 1) K.pv[(kk - 1) * n + i] = pval;
                }
            });
        }
        K.sync();
        return;
    }
    fetch(N - 1, col, qv);
    pval = K.rv[(N - 1) * n + i];       // every group starts from pt_{N-1}; only group C-1 is read at step 0
    if constexpr (XL) ex[tid] = pval;
    K.sync();
    if (tid < n) K.pv[(N - 1) * n + tid] = pval;
    for (int k0 = N - 1; k0 >= 1; k0 -= C) {
        if (k0 - C >= 1) fetch(k0 - C, coln, qvn);
#pragma unroll
        for (int gs = 0; gs < C; gs++) {
            if (k0 - gs >= 1) {
                const int sg = (gs == 0) ? C - 1 : gs - 1;
                // (independent partial sums: the n dependent FMAs of one dot product were most of the time per knot)
                double acc[PS];
#pragma unroll
                for (int q = 0; q < PS; q++) acc[q] = (q == 0) ? qv : 0.0;
                if constexpr (XL) {
                    double pb[n];
#pragma unroll
                    for (int l = 0; l < n; l++) pb[l] = ex[sg * n + l];
#pragma unroll
                    for (int l = 0; l < n; l++) acc[l % PS] += col[l] * pb[l];
                } else {
                    // the 2 n v_readlane of a step first, into n scalar pairs of their own, then the n dependent FMAs: left alone
                    // hipcc reuses ONE scalar pair -- readlane, readlane, s_nop, fmac, n times over -- and every fmac waits for its
                    // two readlanes, which wait for the fmac before them to have read the pair (24 cycles per element, 12 of them
                    // avoidable).  Same FMAs in the same order.
                    double pb[n];
#pragma unroll
                    for (int l = 0; l < n; l++) pb[l] = readlane_f64(pval, sg * n + l);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int l = 0; l < n; l++) acc[l % PS] += col[l] * pb[l];
                }
                const double s = (PS == 4) ? (acc[0] + acc[1]) + (acc[2] + acc[PS - 1]) : acc[0];
                pval = (g == gs) ? s : pval;
                if constexpr (XL) ex[tid] = pval;
            }
        }
        {   // group g produced pt_{kk-1}, kk = k0 - g
            const int kk = k0 - g;
            if (tid < C * n && kk >= 1) K.pv[(kk - 1) * n + i] = pval;
        }
#pragma unroll
        for (int l = 0; l < n; l++) col[l] = coln[l];
        qv = qvn;
    }
    K.sync();
}

// forward: dy_k = Phicl_k dy_{k-1} + ct_k, k = 0..N-1; dY[k] holds ct_k on entry and dy_k on exit
template <class BLKA> GD void forward_sweep_1w(BLKA K) {
    using BLK = std::remove_cv_t<std::remove_reference_t<BLKA>>;   // (a SweepView by value, or a Blk by reference)
    constexpr int n = BLK::n, C = 64 / n, PS = n > 8 ? 4 : 1;
    constexpr bool XL = n > 8;   // (see backward_sweep_1w)
    double* ex = K.sHh;
    using R = typename BLK::R;
    const int tid = K.tid, N = K.N;
    const int g = (tid < C * n) ? tid / n : C - 1, i = (tid < C * n) ? tid % n : 0;
    double gi = 0.0;
    auto phi_e = [&](int r_, int c_) { return (r_ == c_) ? 1.0 : ((c_ == r_ + n / 2) ? K.dt : 0.0); };
    if constexpr (BLK::C::PHI_FROM_K) {
        constexpr int m = BLK::m;
        double Bd[n * m];
        Dyn<BLK::MODEL_ID>::B(*K.mpp, Bd);
        const double h = 0.5 * K.dt;
#pragma unroll
        for (int l = 0; l < n; l++) {   // Gam[i][i mod m] of this lane's row
            const int c_ = l % m;
            const double hb = h * Bd[(c_ + n / 2) * m + c_];
            const double gv = (l < n / 2) ? 2.0 * (h * hb) : 2.0 * hb;
            gi = (i == l) ? gv : gi;
        }
    }
    double phr[n];   // row i of Phi
#pragma unroll
    for (int l = 0; l < n; l++) phr[l] = phi_e(i, l);
    double row[n], rown[n], cv, cvn = 0, yval = 0.0;
    auto fetch = [&](int k0, double* r, double& c) {
        const int kk = (k0 + g < N) ? k0 + g : N - 1;   // (clamped, see backward_sweep_1w)
        if constexpr (BLK::C::PHI_FROM_K) {   // row i of Phicl = Phi - Gam K (see backward_sweep_1w)
            constexpr int m = BLK::m;
            const int ic = (i < m) ? i : i - m;
#pragma unroll
            for (int l = 0; l < n; l++) r[l] = phr[l] - gi * K.kdl[kk * BLK::C::KDS + ic * n + l];
        } else if constexpr (n <= 4) {
            const bool ok = k0 + g < N;
#pragma unroll
            for (int l = 0; l < n; l++) r[l] = ok ? K.Phicl[(size_t)kk * BLK::SPH + i * n + l] : 0.0;
            c = ok ? K.dY[kk * n + i] : 0.0;
            return;
        } else {
#pragma unroll
            for (int l = 0; l < n; l++) r[l] = K.Phicl[(size_t)kk * BLK::SPH + i * n + l];
        }
        c = K.dY[kk * n + i];
    };
    constexpr int RING = (XL && !BLK::C::KD_LDS) ? GUSTO_SWEEP_RING : 1;   // (see backward_sweep_1w)
    if constexpr (RING > 1) {
        double rb[RING][n], qb[RING];
#pragma unroll
        for (int d = 0; d < RING - 1; d++) fetch(d * C, rb[d], qb[d]);
        ex[tid] = yval;
        K.sync();
        for (int kb = 0; kb < N; kb += RING * C) {
            static_for<0, RING>([&](auto DD) {
                constexpr int d = decltype(DD)::value;
                const int k0 = kb + d * C;
                fetch(k0 + (RING - 1) * C, rb[(d + RING - 1) % RING], qb[(d + RING - 1) % RING]);   // (clamped: always loadable)
                if (k0 < N) {
#pragma unroll
                    for (int gs = 0; gs < C; gs++) {
                        if (k0 + gs < N) {
                            const int sg = (gs == 0) ? C - 1 : gs - 1;
                            double acc[PS];
#pragma unroll
                            for (int q = 0; q < PS; q++) acc[q] = (q == 0) ? qb[d] : 0.0;
                            double pb[n];
#pragma unroll
                            for (int l = 0; l < n; l++) pb[l] = readlane_f64(yval, sg * n + l);
                            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                            for (int l = 0; l < n; l++) acc[l % PS] += rb[d][l] * pb[l];
                            const double s = (PS == 4) ? (acc[0] + acc[1]) + (acc[2] + acc[PS - 1]) : acc[0];
                            yval = (g == gs) ? s : yval;
                        }
                    }
                    const int kk = k0 + g;
                    if (tid < C * n && kk < N) K.dY[kk * n + i] = yval;
                }
            });
        }
        K.sync();
        return;
    }
    fetch(0, row, cv);
    if constexpr (XL) ex[tid] = yval;
    K.sync();
    for (int k0 = 0; k0 < N; k0 += C) {
        if (k0 + C < N) fetch(k0 + C, rown, cvn);
#pragma unroll
        for (int gs = 0; gs < C; gs++) {
            if (k0 + gs < N) {
                const int sg = (gs == 0) ? C - 1 : gs - 1;
                double acc[PS];
#pragma unroll
                for (int q = 0; q < PS; q++) acc[q] = (q == 0) ? cv : 0.0;
                if constexpr (XL) {
                    double pb[n];
#pragma unroll
                    for (int l = 0; l < n; l++) pb[l] = ex[sg * n + l];
#pragma unroll
                    for (int l = 0; l < n; l++) acc[l % PS] += row[l] * pb[l];
                } else {
                    double pb[n];   // (see backward_sweep_1w)
#pragma unroll
                    for (int l = 0; l < n; l++) pb[l] = readlane_f64(yval, sg * n + l);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int l = 0; l < n; l++) acc[l % PS] += row[l] * pb[l];
                }
                const double s = (PS == 4) ? (acc[0] + acc[1]) + (acc[2] + acc[PS - 1]) : acc[0];
                yval = (g == gs) ? s : yval;
                if constexpr (XL) ex[tid] = yval;
            }
        }
        {
            const int kk = k0 + g;
            if (tid < C * n && kk < N) K.dY[kk * n + i] = yval;
        }
#pragma unroll
        for (int l = 0; l < n; l++) row[l] = rown[l];
        cv = cvn;
    }
    K.sync();
}

// (Round 4 tried the vector sweeps on the fp64 DPP broadcast gfx950 has -- `v_fmac_f64_dpp ... row_newbcast:L`, the n-vector
// replicated in every row of 16 lanes -- instead of 2 n v_readlane per step.  In isolation a step of n = 6 is 67 cycles against
// 136 (tools/ub/dpp.hip), but the sweep then forms its column of Phicl per knot and lane instead of once per ten knots by groups
// of lanes: bit-identical and 35.9 ms against 34.1 ms per config-2 batch.  The variant without per-knot operands,
// Phi^T p - K^T (Gam^T p), is as fast as the readlane sweep and numerically worse -- it subtracts two large terms that
// Phi - Gam K cancels entry by entry first: 1.5 % more interior point iterations, lock-step tolerances missed.  Neither kept.)
// nu_{k+1} = P_k dy_k + p_k + Pi_k mu_g of every knot (the new costates of the corrector), for the 12/13-state models by
// groups of n lanes: lane i of group g forms row i for knot k0 + g from ITS rows of the P_k and Pi_k records (stored transposed:
// the lanes of a group read consecutive doubles), the next chunk's rows in flight while this one is summed.  A load
// instruction then touches one line of each of C = 64 / n records; with a
// lane per knot (step_phase) each of the 2 n^2 loads of the walk touched 50 records -- ~230 cycles apiece through the
// texture addresser, 80 k of the 1.08 M cycles of a KKT solve.  Same sums in the same order as step_phase.
template <int MODEL> GD void costate_pass_1w(SweepView<MODEL> K, const double* mugn) {
    using T = MT<MODEL>;
    using R = Rec<MODEL>;
    constexpr int n = T::n, C = 64 / n;
    const int tid = K.tid, N = K.N;
    const int g = (tid < C * n) ? tid / n : C - 1, i = (tid < C * n) ? tid % n : 0;
    double mg[n];
#pragma unroll
    for (int l = 0; l < n; l++) mg[l] = mugn[l];
    constexpr int RING = GUSTO_SWEEP_RING;   // chunk buffers: RING - 1 fetches in flight (see backward_sweep_1w)
    double pr[RING][n], pi[RING][n];
    auto fetch = [&](int k0, double* a, double* b) {
        const int k = (k0 + g + 1 < N) ? k0 + g : N - 2;   // (clamped: no load under a branch)
        // (records stored transposed by factor_sweep_mfma: entry (i, l) at l n + i; row-major by the VALU sweep of a
        // -DGUSTO_USE_MFMA=false build)
        const double* pa = K.Paft + (size_t)k * R::SNN + (T::MFMA ? i : i * n);
        const double* pb = K.Piaft + (size_t)k * R::SNN + (T::MFMA ? i : i * n);
#pragma unroll
        for (int l = 0; l < n; l++) { a[l] = pa[T::MFMA ? l * n : l]; b[l] = pb[T::MFMA ? l * n : l]; }
    };
#pragma unroll
    for (int d = 0; d < RING - 1; d++) fetch(d * C, pr[d], pi[d]);
    for (int kb = 0; kb + 1 < N; kb += RING * C) {
        static_for<0, RING>([&](auto DD) {
            constexpr int d = decltype(DD)::value;
            const int k0 = kb + d * C;
            fetch(k0 + (RING - 1) * C, pr[(d + RING - 1) % RING], pi[(d + RING - 1) % RING]);
            if (k0 + 1 < N) {
                const bool ok = tid < C * n && k0 + g + 1 < N;
                const int k = (k0 + g + 1 < N) ? k0 + g : N - 2;
                double s = K.pv[k * n + i] - K.rv[k * n + i];
#pragma unroll
                for (int l = 0; l < n; l++) s += pr[d][l] * K.dY[k * n + l] + pi[d][l] * mg[l];
                if (ok) K.nun[(k + 1) * n + i] = s;
            }
        });
    }
    K.sync();
}
template <int MODEL> __device__ __noinline__ void costate_pass_1w_call(typename Blk<MODEL, true>::Args a) {
    Blk<MODEL, true> B(a, gusto_dyn_lds);
    costate_pass_1w<MODEL>(SweepView<MODEL>::make(B), gusto_dyn_lds + LdsC<MODEL, true>::misc + 48);
}


// Adjoint recursion of the new costates (costate_adjoint): nu_k = Phi_k^T nu_{k+1} + v_k, k = N-2 .. 1, nu_{N-1} = v_{N-1};
// nun[k] holds v_k on entry (step_phase) and nu_k on exit.  Phi_k = 2 M_k - I is the block linearize() stored; of its column
// i only the structural nonzeros are fetched (x = (r, v, attitude, w): M is block upper triangular, MT::Mnz -- one or two
// entries for a position / velocity column, the attitude and rate rows for the others): 36 / 46 doubles per knot instead
// of the 2 n^2 of the P | Pi records.  Groups of n lanes hold consecutive knots as in backward_sweep_1w; the n-vector travels
// between groups through 64 doubles of LDS (each lane needs ITS rows of it: a per-lane address, not a broadcast).
template <int MODEL> constexpr bool adjoint_pattern_ok() {   // the column pattern adjoint_sweep_1w assumes IS MT::Mnz
    using T = MT<MODEL>;
    constexpr int n = T::n, q = n - 9;
    for (int i = 0; i < n; i++)
        for (int l = 0; l < n; l++) {
            const bool want = (i < 6) ? (l == i || (i >= 3 && l == i - 3)) : (l >= 6 && l < ((i < 6 + q) ? 6 + q : n));
            if (want != T::Mnz(l, i)) return false;
        }
    return true;
}
template <int MODEL> GD void adjoint_sweep_1w(SweepView<MODEL> K) {
    using T = MT<MODEL>;
    constexpr int n = T::n, NZ = T::n + T::m, C = 64 / n, q = n - 9, NR = n - 6;
    static_assert(adjoint_pattern_ok<MODEL>(), "column pattern of M of the astrobee models");
    const int tid = K.tid, N = K.N;
    const int g = (tid < C * n) ? tid / n : C - 1, i = (tid < C * n) ? tid % n : 0;
    // rows of column i: {i} and {i - 3} below the attitude block, else rows 6 .. hi - 1, hi = 6 + q for an attitude column, n for a rate column
    int rl[NR];
    bool rok[NR];
    const int hi = (i < 6 + q) ? 6 + q : n;
#pragma unroll
    for (int j = 0; j < NR; j++) {
        rl[j] = (i < 6) ? ((j == 1 && i >= 3) ? i - 3 : i) : 6 + j;
        rok[j] = (i < 6) ? (j == 0 || (j == 1 && i >= 3)) : (6 + j < hi);
        if (!rok[j]) rl[j] = i;
    }
    // where those entries sit in the compact record (SpPG: Phi column by column, rows ascending): column start + rank
    int cs = 0, pj[NR];
#pragma unroll
    for (int i2 = 0; i2 < n; i2++) cs = (i == i2) ? SpPG<MODEL>::pos_phi(0, i2) : cs;
#pragma unroll
    for (int j = 0; j < NR; j++) pj[j] = (i < 6) ? ((i >= 3) ? (j == 1 ? 0 : (j == 0 ? 1 : 0)) : 0) : (rok[j] ? j : 0);
    double* ex = K.sHh;
    constexpr int RING = GUSTO_SWEEP_RING;
    double cb[RING][NR], vb[RING];
    auto fetch = [&](int k0, double* c, double& v) {
        const int kk = (k0 - g >= 1) ? k0 - g : 1;      // (clamped: always loadable, never used past the end)
        const auto pg = K.PGS + (size_t)kk * SpPG<MODEL>::S + cs;   // (column i of Phi_kk: consecutive entries of the compact record)
#pragma unroll
        for (int j = 0; j < NR; j++) { const double a = pg[pj[j]]; c[j] = rok[j] ? a : 0.0; }
        v = K.nun[kk * n + i];
    };
#pragma unroll
    for (int d = 0; d < RING - 1; d++) fetch(N - 2 - d * C, cb[d], vb[d]);
    double val = K.nun[(N - 1) * n + i];       // every group starts from nu_{N-1}; only group C - 1 is read at step 0
    ex[tid] = val;
    K.sync();
    for (int kb = N - 2; kb >= 1; kb -= RING * C) {
        static_for<0, RING>([&](auto DD) {
            constexpr int d = decltype(DD)::value;
            const int k0 = kb - d * C;
            fetch(k0 - (RING - 1) * C, cb[(d + RING - 1) % RING], vb[(d + RING - 1) % RING]);
            if (k0 >= 1) {
#pragma unroll
                for (int gs = 0; gs < C; gs++) {
                    if (k0 - gs >= 1) {
                        const int sg = (gs == 0) ? C - 1 : gs - 1;
                        double pb[NR];
#pragma unroll
                        for (int j = 0; j < NR; j++) pb[j] = ex[sg * n + rl[j]];
                        double s = vb[d];
#pragma unroll
                        for (int j = 0; j < NR; j++) s += cb[d][j] * pb[j];
                        val = (g == gs) ? s : val;
                        ex[tid] = val;
                    }
                }
                const int kk = k0 - g;
                if (tid < C * n && kk >= 1) K.nun[kk * n + i] = val;
            }
        });
    }
    K.sync();
}
template <int MODEL> __device__ __noinline__ void adjoint_sweep_1w_call(typename Blk<MODEL, true>::Args a, double hdt) {
    using T = MT<MODEL>;
    constexpr int n = T::n, m = T::m;
    Blk<MODEL, true> K(a, gusto_dyn_lds);
    adjoint_sweep_1w<MODEL>(SweepView<MODEL>::make(K));
    if (K.tid == 0) {  // x_1 stationarity: gx_0 + nu_0 + F_0^T nu_1 = 0 (as step_phase closes the other kernels)
        const double* gxs = gusto_dyn_lds + LdsC<MODEL, true>::misc + 16;
        double Ad[n * n], x0[n], u0[m];
#pragma unroll
        for (int i = 0; i < n; i++) x0[i] = K.Xp[i];
#pragma unroll
        for (int i = 0; i < m; i++) u0[i] = K.Up[i];
        Dyn<MODEL>::A(K.P.mp, x0, u0, Ad);
#pragma unroll
        for (int i = 0; i < n; i++) {
            double s = gxs[i] + K.nun[n + i];
#pragma unroll
            for (int j = 0; j < n; j++) if (T::Anz(j, i)) s += hdt * Ad[j * n + i] * K.nun[n + j];
            K.nun[i] = -s;
        }
    }
    K.sync();
}

// MT::SWEEP_CALL (measured per model: astrobeeSE3 +9 %, the manifold model -10 %): the sweep as a real call.  Inlined, its 50-stage loop shares one register allocation with the whole
// interior point iteration and the allocator spills INSIDE the loop; called, the loop gets the register file to itself
// and the caller's live values are saved once around the call.
template <int MODEL> __device__ __noinline__ void factor_sweep_1w_call(typename Blk<MODEL, true>::Args a, Prof* pf) {
    Blk<MODEL, true> B(a, gusto_dyn_lds);
    SweepView<MODEL> K = SweepView<MODEL>::make(B);
    if constexpr (MT<MODEL>::MFMA) {
        if constexpr (!costate_adjoint<MODEL>()) factor_sweep_mfma<MODEL, false>(K, gusto_dyn_lds + LdsC<MODEL, true>::misc + 8, *pf);
        else if (costate_adjoint_now<MODEL>()) factor_sweep_mfma<MODEL, true>(K, gusto_dyn_lds + LdsC<MODEL, true>::misc + 8, *pf);
        else factor_sweep_mfma<MODEL, false>(K, gusto_dyn_lds + LdsC<MODEL, true>::misc + 8, *pf);
    }
    else factor_sweep_1w<MODEL>(K, gusto_dyn_lds + LdsC<MODEL, true>::misc + 8, *pf);
}
// the multi-wave sweeps of the TrajOpt kernels whose phases are real calls (MT::SWEEP_CALL with defect controls): a register
// allocation of their own, like the one-wave sweeps of the 12/13-state GuSTO kernels
template <int MODEL, class BLK> __device__ __noinline__ void factor_sweep_mw_call(typename BLK::Args a, Prof* pf) {
    BLK K(a, gusto_dyn_lds);
    if constexpr (MT<MODEL>::m >= GUSTO_COOP_CHOL_MIN) {
        if (K.NTr <= 64) { factor_sweep_w1<MODEL>(K, gusto_dyn_lds + BLK::C::misc + 8, *pf); return; }
    }
    factor_sweep_mw<MODEL>(K, gusto_dyn_lds + BLK::C::misc + 8, *pf);
}
// (a problem of ONE wave -- every TrajOpt launch with N <= 64 -- takes the one-wave vector sweeps: groups of n lanes for
// consecutive knots, one batch of loads per chunk of the chain instead of a trip to the Phicl record per knot)
template <class BLK> __device__ __noinline__ void backward_sweep_mw_call(typename BLK::Args a) {
    BLK K(a, gusto_dyn_lds);
    if (K.NTr <= 64) backward_sweep_1w<const BLK&>(K); else backward_sweep_mw(K);
}
template <class BLK> __device__ __noinline__ void forward_sweep_mw_call(typename BLK::Args a) {
    BLK K(a, gusto_dyn_lds);
    if (K.NTr <= 64) forward_sweep_1w<const BLK&>(K); else forward_sweep_mw(K);
}
template <int MODEL, class BLK> GD void factor_sweep(BLK& K, double* fail, Prof& pf) {
    if constexpr (!BLK::ONE && MT<MODEL>::SWEEP_CALL && MT<MODEL>::NDEF > 0) factor_sweep_mw_call<MODEL, BLK>(K.args(), &pf);
    else if constexpr (!BLK::ONE) factor_sweep_mw<MODEL>(K, fail, pf);
    else if constexpr (MT<MODEL>::SWEEP_CALL) factor_sweep_1w_call<MODEL>(K.args(), &pf);
    else if constexpr (MT<MODEL>::MFMA) factor_sweep_mfma<MODEL, false>(SweepView<MODEL>::make(K), fail, pf);   // (inlined builds, -DGUSTO_SWEEP_INLINE: P | Pi costates)
    else if constexpr (MT<MODEL>::PG2 && LdsC<MODEL, true>::KD_LDS) {
        factor_sweep_pg2<MODEL>(SweepView<MODEL>::make(K), fail, pf);
    }
    else factor_sweep_1w<MODEL>(SweepView<MODEL>::make(K), fail, pf);
}
template <int MODEL> __device__ __noinline__ void backward_sweep_1w_call(typename Blk<MODEL, true>::Args a) {
    Blk<MODEL, true> B(a, gusto_dyn_lds);
    backward_sweep_1w(SweepView<MODEL>::make(B));
}
template <int MODEL> __device__ __noinline__ void forward_sweep_1w_call(typename Blk<MODEL, true>::Args a) {
    Blk<MODEL, true> B(a, gusto_dyn_lds);
    forward_sweep_1w(SweepView<MODEL>::make(B));
}
template <int MODEL, class BLK> GD void backward_sweep(BLK& K) {
    if constexpr (!BLK::ONE && MT<MODEL>::SWEEP_CALL && MT<MODEL>::NDEF > 0) backward_sweep_mw_call<BLK>(K.args());
    else if constexpr (!BLK::ONE) backward_sweep_mw(K);
    else if constexpr (MT<MODEL>::SWEEP_CALL) backward_sweep_1w_call<MODEL>(K.args());
    else backward_sweep_1w(SweepView<MODEL>::make(K));
}
template <int MODEL, class BLK> GD void forward_sweep(BLK& K) {
    if constexpr (!BLK::ONE && MT<MODEL>::SWEEP_CALL && MT<MODEL>::NDEF > 0) forward_sweep_mw_call<BLK>(K.args());
    else if constexpr (!BLK::ONE) forward_sweep_mw(K);
    else if constexpr (MT<MODEL>::SWEEP_CALL) forward_sweep_1w_call<MODEL>(K.args());
    else forward_sweep_1w(SweepView<MODEL>::make(K));
}

// Gd^-1 -> sP for the models whose goal system does not fit one lane's registers, by the whole workgroup: in-place
// right-looking Cholesky of sGd (identity on the coordinates without a point goal), L^-1 one column per lane, then
// L^-T L^-1 one entry per lane.  Scratch: sHh.
template <int MODEL, class BLK> GD void inv_spd_block(BLK& K, double* fail) {
    constexpr int n = BLK::n;
    const int tid = K.tid, nt = K.nt();
    double* A = K.sGd;
    double* Li = K.sHh;   // (free between the factor sweep and the next one; NZ^2 >= n^2 doubles)
    if constexpr (BLK::ONE) {
        // One wave: the matrix lives in registers, lane r holds row r, and what a step needs from another lane comes by
        // v_readlane -- no LDS round trip and no fence inside the factorisation (the LDS version below pays three per column
        // and walks the forward substitution through dependent LDS reads: 28 k cycles per KKT solve for n = 12 / 13).  The
        // same operations in the same order per entry (terms the LDS version skips enter as exact zeros): bit-identical.
        const int r = tid < n ? tid : n - 1;   // (lanes >= n mirror the last row and are never read)
        double a[n];
#pragma unroll
        for (int c = 0; c < n; c++) { a[c] = A[r * n + c]; if (c == r && !K.is_goal(r)) a[c] = 1.0; }
        bool bad = false;
        static_for<0, n>([&](auto J) {
            constexpr int j = decltype(J)::value;
            const double ajj = readlane_f64(a[j], j);
            if (!(ajj > 0.0)) bad = true;
            const double d = rsqrt_nr(ajj);
            a[j] = (r >= j) ? a[j] * d : a[j];                  // column j of L
            static_for<j + 1, n>([&](auto C) {                  // trailing update of the lower triangle
                constexpr int c = decltype(C)::value;
                const double lcj = readlane_f64(a[j], c);
                double t = a[c];
                t -= a[j] * lcj;
                a[c] = (r >= c) ? t : a[c];
            });
        });
        if (bad) *fail = 1.0;
        // column r of L^-1 by forward substitution; row i of L comes from lane i
        double x[n];
        static_for<0, n>([&](auto I) {
            constexpr int i = decltype(I)::value;
            const double ri = rcp_nr(readlane_f64(a[i], i));
            double acc = 0;
            static_for<0, i>([&](auto L_) {
                constexpr int l = decltype(L_)::value;
                acc -= readlane_f64(a[l], i) * x[l];
            });
            x[i] = (i == r) ? ri : ((i > r) ? acc * ri : 0.0);
        });
        if (tid < n) {
#pragma unroll
            for (int i = 0; i < n; i++) Li[i * n + tid] = x[i];
        }
        K.sync();
#pragma unroll
        for (int t = 0; t < (n * n + 63) / 64; t++) {          // L^-T L^-1, an entry per lane: independent LDS reads
            const int e = tid + 64 * t, ec = e < n * n ? e : n * n - 1;
            const int i = ec / n, j = ec % n;
            double acc = 0;
#pragma unroll
            for (int l = 0; l < n; l++) acc += Li[l * n + i] * Li[l * n + j];
            if (e < n * n) K.sP[e] = acc;
        }
        return;
    }
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e % n;
        if (i == j && !K.is_goal(i)) A[e] = 1.0;
        Li[e] = 0.0;
    }
    K.sync();
    for (int j = 0; j < n; j++) {
        const double ajj = A[j * n + j];
        if (!(ajj > 0.0)) *fail = 1.0;
        const double d = rsqrt_nr(ajj);
        K.sync();
        if (tid >= j && tid < n) A[tid * n + j] *= d;          // column j of L (row j: a_jj * d = sqrt(a_jj))
        K.sync();
        for (int e = tid; e < n * n; e += nt) {                 // trailing update of the lower triangle
            const int i = e / n, c = e % n;
            if (c > j && i >= c) A[e] -= A[i * n + j] * A[c * n + j];
        }
        K.sync();
    }
    if (tid < n) {   // column tid of L^-1 by forward substitution (1 / l_ii by Newton reciprocal)
        const int c = tid;
        Li[c * n + c] = rcp_nr(A[c * n + c]);
        for (int i = c + 1; i < n; i++) {
            double acc = 0;
            for (int l = c; l < i; l++) acc -= A[i * n + l] * Li[l * n + c];
            Li[i * n + c] = acc * rcp_nr(A[i * n + i]);
        }
    }
    K.sync();
    for (int e = tid; e < n * n; e += nt) {
        const int i = e / n, j = e % n;
        double acc = 0;
        for (int l = (i > j ? i : j); l < n; l++) acc += Li[l * n + i] * Li[l * n + j];
        K.sP[e] = acc;
    }
}

// The phase between the two vector sweeps of a right-hand side: feed-forward d0 = S^-1 lu, the goal multiplier
// mu_g = Gd^-1 theta, then d_k = d0 + D_k mu_g and ct_k = c_k - Gam_k d_k.  A function of its own so that the models
// with MT::SWEEP_CALL can run it as a real call (own register allocation; everything it touches lives in LDS / HBM).
// NCH > 0 (round 6, segw.hpp: the matrix-core kernels' segmented solve): theta is summed per chain, the chains' vectors are folded from
// the back, the first interface gives mu_g, then every interface its state xi and the increment dlam on its costate, front to back;
// a knot in front of an interface takes that interface's dlam where one of the last chain takes mu_g.
template <int MODEL, class BLK, int NCH = 0> GD void mid_phase(BLK& K, int k, bool act, double hdt, double* red, double* mugn, Prof* pf = nullptr) {
    constexpr bool SEG = NCH > 0;
#define MT_(i) do { if (pf) pf->tick(i); } while (0)
    MT_(PF_MID);
    using T = MT<MODEL>;
    using R = Rec<MODEL>;
    constexpr int n = T::n, m = T::m, NZ = n + m;
    const int N = K.N;
    // feed-forward d0 = S^-1 lu and the goal multiplier
    double th[n], d0[m];
    // (12/13-state models: the D record of the knot, 78 entries from the slot workspace, is walked once and kept for d_k = d0 + D mu_g
    // below -- this phase is a real call with registers of its own -- instead of walked again after the reductions)
    constexpr bool KEEP_D = T::SWEEP_CALL && BLK::ONE && !BLK::C::KD_LDS;
    double Dk[KEEP_D ? m * n : 1];
#pragma unroll
    for (int i = 0; i < (KEEP_D ? m * n : 1); i++) Dk[i] = 0;
#pragma unroll
    for (int i = 0; i < n; i++) th[i] = 0;
#pragma unroll
    for (int i = 0; i < m; i++) d0[i] = 0;
    if (act) {
        double tt[n], lu[m], Gamk[n * m];
        // (12/13-state models: the last knot's goal term M rd from the M this phase recomputes anyway, by EVERY lane on its own
        // knot's data and selected below -- as a branch per goal coordinate it was n single-lane walks of the [Phi Gam]
        // record, one memory round trip each (the workspace of these models lives in HBM: ~3 k cycles), twice per iteration)
        constexpr bool GT_RECOMP = !T::LTI && n > 8 && T::NDEF == 0;
        double gterm[n], gsub[n];
        if (k >= 1) {
            if constexpr (T::PG2 || (!T::LTI && n > 8 && T::NDEF == 0)) {
                double Mk_[n * n];
                load_M_Gam(K, k, Mk_, Gamk);
                if constexpr (GT_RECOMP) {
#pragma unroll
                    for (int j = 0; j < n; j++) {
                        double g = 0.0;
#pragma unroll
                        for (int i = 0; i < n; i++) if (T::Mnz(j, i)) g += Mk_[j * n + i] * K.rd_(k, i);
                        gterm[j] = g; gsub[j] = K.misc[64 + j] - K.Xw[k * n + j];
                    }
                }
            } else {
                const auto pg = K.PGk(k);
#pragma unroll
                for (int i = 0; i < n; i++)
#pragma unroll
                    for (int j = 0; j < m; j++) Gamk[i * m + j] = T::Gnz(i, j) ? pg[i * NZ + n + j] : 0.0;
            }
        } else {
            Dyn<MODEL>::B(K.P.mp, Gamk);
#pragma unroll
            for (int i = 0; i < n * m; i++) Gamk[i] *= hdt;
            if constexpr (T::NDEF > 0) {
#pragma unroll
                for (int i = 0; i < T::NDEF; i++) Gamk[i * m + (m - T::NDEF) + i] = 1.0;
            }
        }
#pragma unroll
        for (int i = 0; i < n; i++) tt[i] = K.pv[k * n + i];   // pt_k = p_k + r_k
#pragma unroll
        for (int i = 0; i < m; i++) {
            double s = K.qu_(k, i);
#pragma unroll
            for (int l = 0; l < n; l++) if (T::Gnz(l, i)) s += Gamk[l * m + i] * tt[l];
            lu[i] = s;
        }
#pragma unroll
        for (int i = 0; i < m; i++) {
            double s = 0;
#pragma unroll
            for (int l = 0; l < m; l++) s += K.kdS(k, i, l) * lu[l];
            d0[i] = s;
        }
        // theta_j = sum_k Pi_k^T c_k - D_k^T lu_k  (+ C M rd_{N-1} - rg at the last knot)
        // LTI models: the last knot's goal term is evaluated by EVERY lane on its own knot's data and selected afterwards --
        // as a branch it is single-lane work (loads and all) that the whole wave waits for, twice per iteration
        if constexpr (T::LTI) {
            double rdl[n], Mg[n * n];
            if constexpr (T::PG2) {   // (closed form, exactly the stored block: 0.5 ((2 M - I) + I) = M entry by entry)
                double Gg[n * m];
                load_M_Gam(K, k, Mg, Gg);
            } else {
                const auto pg = K.PGk(0);
#pragma unroll
                for (int j = 0; j < n; j++)
#pragma unroll
                    for (int i = 0; i < n; i++) Mg[j * n + i] = T::Mnz(j, i) ? 0.5 * (pg[j * NZ + i] + (i == j ? 1.0 : 0.0)) : 0.0;
            }
#pragma unroll
            for (int i = 0; i < n; i++) rdl[i] = K.rd_(k, i);
#pragma unroll
            for (int j = 0; j < n; j++) {
                double g = 0.0;
#pragma unroll
                for (int i = 0; i < n; i++) if (T::Mnz(j, i)) g += Mg[j * n + i] * rdl[i];
                gterm[j] = g; gsub[j] = K.misc[64 + j] - K.Xw[k * n + j];
            }
        }
        // (the record of D is walked in storage order, row by row.  Lane k reads ITS knot's record, so one load touches 50
        // cache lines, and the 16 entries of a line are served by the L1 only while the wave's live set stays ~50 lines:
        // column by column it is every line of the block x 50 knots, which does not fit next to the other waves.  The
        // knot-major records were also tried entry-major (one coalesced load per entry): every load then goes to the
        // L2 and the stage-parallel phases got 40-170 % slower.)
        double thd[n];
#pragma unroll
        for (int j = 0; j < n; j++) thd[j] = K.nun[k * n + j];   // Pi_k^T c_k from the factor sweep (nun is free until the corrector's costates)
#pragma unroll
        for (int i = 0; i < m; i++)
#pragma unroll
            for (int j = 0; j < n; j++) {
                const double dij = K.kd(k, R::oD + i * n + j);
                if constexpr (KEEP_D) Dk[i * n + j] = dij;
                thd[j] -= dij * lu[i];
            }
#pragma unroll
        for (int j = 0; j < n; j++) {
            double s = thd[j];
            if constexpr (T::LTI || GT_RECOMP) {
                s = (k == N - 1 && K.is_goal(j)) ? (s + gterm[j]) - gsub[j] : s;   // (N >= 2: the last knot has k >= 1)
            } else
            if (k == N - 1 && K.is_goal(j)) {
                const auto pg = K.PGk(k);
#pragma unroll
                for (int i = 0; i < n; i++) s += 0.5 * (pg[j * NZ + i] + (i == j ? 1.0 : 0.0)) * K.rd_(k, i);
                s -= K.misc[64 + j] - K.Xw[k * n + j];
            }
            th[j] = s;
        }
    }
    MT_(PF_M_TH);
    if constexpr (SEG) {
        // lane i < n forms row i of every matrix-vector product; a vector travels to all lanes by v_readlane.  The rows a level
        // needs are requested together, ahead of the products (one LDS round trip per level, not one per term).
        using SB = SegB<MODEL, NCH>;
        constexpr int NI = NCH - 1;
        const LPtr<double> L = K.lds;
        const int sb = K.P.ll.seg;
        const int ri = K.tid < n ? K.tid : 0;
        // theta per chain (the last chain's with the goal terms).  Through LDS -- helper 1's block is idle here --: lane (c, j) adds entry
        // j of chain c's knots, then the NCH n sums travel by v_readlane.  (As NCH n masked values through wave_reduce_n this was
        // 52 x 23 DPP / readlane instructions for four chains: 4.8 k cycles per solve; now ~1 k.)
        double r[NCH * n];
        {
            constexpr int MAXL = (64 + NCH - 1) / NCH + 1;   // knots of a chain at most (N <= 64)
            const int tb = sb + SB::sPG2(0);
            if (act) {
#pragma unroll
                for (int j = 0; j < n; j++) L[tb + k * n + j] = th[j];
            }
            K.sync();
            const int t = K.tid < NCH * n ? K.tid : NCH * n - 1, tc = t / n, tj = t % n;
            const int k0 = seg_lo(tc, N, NCH), k1 = seg_lo(tc + 1, N, NCH);
            double v[MAXL];
#pragma unroll
            for (int q = 0; q < MAXL; q++) v[q] = L[tb + ((k0 + q < k1) ? k0 + q : k0) * n + tj];
            __builtin_amdgcn_sched_barrier(0);
            double s = 0;
#pragma unroll
            for (int q = 0; q < MAXL; q++) s += (k0 + q < k1) ? v[q] : 0.0;
#pragma unroll
            for (int e = 0; e < NCH * n; e++) r[e] = readlane_f64(s, e);
        }
        MT_(PF_M_RED);
        auto row = [&](int o, double* v) {
#pragma unroll
            for (int l = 0; l < n; l++) v[l] = L[o + ri * n + l];
        };
        auto col = [&](int o, double* v) {
#pragma unroll
            for (int l = 0; l < n; l++) v[l] = L[o + l * n + ri];
        };
        auto bcast = [&](double x, double* v) {
#pragma unroll
            for (int l = 0; l < n; l++) v[l] = readlane_f64(x, l);
        };
        auto dot = [&](const double* a, const double* x) {
            double s = 0;
#pragma unroll
            for (int l = 0; l < n; l++) s += a[l] * x[l];
            return s;
        };
        // p_hat of an interface = the costate offset in front of what lies behind it, seen from the costate iterate lam0 there
        auto lam0 = [&](int j, const double* pc, double* ph) {
            const int o = seg_lo(j + 1, N, NCH) * n;
#pragma unroll
            for (int l = 0; l < n; l++) ph[l] = pc[l] - K.nu[o + l];
        };
        auto ufrom = [&](int o, double* v) {
#pragma unroll
            for (int l = 0; l < n; l++) v[l] = L[o + l];
        };
        auto pick = [&](const double* vU) {   // this lane's entry of a vector every lane holds
            double e = 0;
#pragma unroll
            for (int l = 0; l < n; l++) e = (ri == l) ? vU[l] : e;
            return e;
        };
        if constexpr (NCH == 2) {   // one interface: mu_g, then its state and costate increment
            const int I = sb + SB::IF(0);
            double pf1[n], ph[n], rT[n], rS[n], rP[n], cT[n], rG[n], rA1[n], rA2[n], rA3[n];
            row(I + SB::Tt, rT); row(I + SB::Sg, rS); row(I + SB::Pa, rP); col(I + SB::Tt, cT);
            row(sb + SB::Gci, rG); row(sb + SB::A1, rA1); row(I + SB::A2, rA2); row(I + SB::A3, rA3);
            ufrom(sb + SB::PBV(1), pf1);
            lam0(0, pf1, ph);
            __builtin_amdgcn_sched_barrier(0);
            const double w1 = dot(rT, r) - dot(rS, ph), v3 = dot(rP, r) + dot(cT, ph);
            double w1U[n], muU[n];
            bcast(w1, w1U);
            double mu = dot(rG, r + n) + dot(rA1, w1U);
            mu = K.is_goal(ri) ? mu : 0.0;
            bcast(mu, muU);
            const double xiv = w1 - dot(rA2, muU), dlv = v3 + dot(rA3, muU);
            if (K.tid < n) { mugn[K.tid] = mu; L[sb + SB::XI(0) + K.tid] = xiv; L[sb + SB::LAM(0) + K.tid] = dlv; }
        } else {
            // four chains, merged as a tree (segw.hpp: seg_fold_tree_*): (C0 | C1) and (C2 | C3), then the pairs at interface 1
            static_assert(NCH == 4, "two or four chains");
            const int I0 = sb + SB::IF(0), I1 = sb + SB::IF(1), I2 = sb + SB::IF(2), C1 = sb + SB::CH(1), C2 = sb + SB::CH(2);
            double u0, w0, u2, w2, u0U[n], u2U[n], w2U[n];
            {   // the two outer interfaces, each from its own chains' vectors
                double pfa[n], pfb[n], ph0[n], ph2[n], rT[n], rS[n], rP[n], cT[n], rT2[n], rS2[n], rP2[n], cT2[n];
                row(I0 + SB::Tt, rT); row(I0 + SB::Sg, rS); row(I0 + SB::Pa, rP); col(I0 + SB::Tt, cT);
                row(I2 + SB::Tt, rT2); row(I2 + SB::Sg, rS2); row(I2 + SB::Pa, rP2); col(I2 + SB::Tt, cT2);
                ufrom(sb + SB::PBV(1), pfa); ufrom(sb + SB::PBV(3), pfb);
                lam0(0, pfa, ph0); lam0(2, pfb, ph2);
                __builtin_amdgcn_sched_barrier(0);
                u0 = dot(rT, r) - dot(rS, ph0); w0 = dot(rP, r) + dot(cT, ph0);
                u2 = dot(rT2, r + 2 * n) - dot(rS2, ph2); w2 = dot(rP2, r + 2 * n) + dot(cT2, ph2);
                bcast(u0, u0U); bcast(u2, u2U); bcast(w2, w2U);
            }
            double th01U[n], th23U[n], p23U[n];
            {   // the pairs: theta of (C0 C1), theta and front costate offset of (C2 C3)
                double cP1[n], cP3[n], rP2_[n];
                col(C1 + SB::Pif, cP1); col(I2 + SB::PIc, cP3); row(C2 + SB::Pif, rP2_);
                const double pf2 = L[sb + SB::PBV(2) + ri];
                __builtin_amdgcn_sched_barrier(0);
                const double th01 = pick(r + n) + dot(cP1, u0U), th23 = pick(r + 3 * n) + dot(cP3, u2U), p23 = pf2 + dot(rP2_, w2U);
                bcast(th01, th01U); bcast(th23, th23U); bcast(p23, p23U);
            }
            double xi1, dl1, muU[n], xi1U[n], dl1U[n];
            {   // interface 1: mu_g, its state and costate increment
                double ph1[n], rT[n], rS[n], rP[n], cT[n], rG[n], rA1[n], rA2[n], rA3[n];
                row(I1 + SB::Tt, rT); row(I1 + SB::Sg, rS); row(I1 + SB::Pa, rP); col(I1 + SB::Tt, cT);
                row(sb + SB::Gci, rG); row(sb + SB::A1, rA1); row(I1 + SB::A2, rA2); row(I1 + SB::A3, rA3);
                lam0(1, p23U, ph1);
                __builtin_amdgcn_sched_barrier(0);
                const double w1 = dot(rT, th01U) - dot(rS, ph1), v3 = dot(rP, th01U) + dot(cT, ph1);
                double w1U[n];
                bcast(w1, w1U);
                double mu = dot(rG, th23U) + dot(rA1, w1U);
                mu = K.is_goal(ri) ? mu : 0.0;
                bcast(mu, muU);
                xi1 = w1 - dot(rA2, muU); dl1 = v3 + dot(rA3, muU);
                if (K.tid < n) { mugn[K.tid] = mu; L[sb + SB::XI(1) + K.tid] = xi1; L[sb + SB::LAM(1) + K.tid] = dl1; }
                bcast(xi1, xi1U); bcast(dl1, dl1U);
            }
            {   // the outer interfaces: 0 hangs on interface 1's multiplier, 2 on its state and on mu_g
                double rA2[n], rA3[n], cP2[n], rT[n], rP[n], rB2[n], rB3[n], zU[n];
                row(I0 + SB::A2, rA2); row(I0 + SB::A3, rA3); col(C2 + SB::Pif, cP2);
                row(I2 + SB::Tt, rT); row(I2 + SB::Pa, rP); row(I2 + SB::A2, rB2); row(I2 + SB::A3, rB3);
                __builtin_amdgcn_sched_barrier(0);
                const double xi0 = u0 - dot(rA2, dl1U), dl0 = w0 + dot(rA3, dl1U);
                const double z = dot(cP2, xi1U);    // Pi_2' xi_1
                bcast(z, zU);
                const double xi2 = u2 + dot(rT, zU) - dot(rB2, muU), dl2 = w2 + dot(rP, zU) + dot(rB3, muU);
                if (K.tid < n) {
                    L[sb + SB::XI(0) + K.tid] = xi0; L[sb + SB::LAM(0) + K.tid] = dl0;
                    L[sb + SB::XI(2) + K.tid] = xi2; L[sb + SB::LAM(2) + K.tid] = dl2;
                }
            }
        }
    } else
    if constexpr (BLK::ONE) {
        if (K.goalmask != 0) wave_reduce_n<n>(th, OpSum());
        else {
#pragma unroll
            for (int j = 0; j < n; j++) th[j] = 0.0;
        }
    } else {
#pragma unroll
    for (int j = 0; j < n; j++) th[j] = (K.goalmask != 0) ? block_reduce<BLK::ONE>(th[j], OpSum(), red) : 0.0;
    }
    if constexpr (!SEG) {
    MT_(PF_M_RED);
    if constexpr (n > 4) {
        if (k < n) {   // mu_g = Gd^-1 theta, a lane per row (theta is wave-uniform after the reductions): as lane 0's loop it was n^2
                       // LDS reads and FMAs that every other lane waited for, twice per interior point iteration.  Same sums:
                       // bit-identical.  (Not for the 3-state model, two waves per SIMD: measured 2 % slower there.)
            double s = 0;
#pragma unroll
            for (int l = 0; l < n; l++) s += K.sP[k * n + l] * th[l];
            mugn[k] = K.is_goal(k) ? s : 0.0;
        }
    } else if (k == 0) {
#pragma unroll
        for (int j = 0; j < n; j++) {
            double s = 0;
#pragma unroll
            for (int l = 0; l < n; l++) s += K.sP[j * n + l] * th[l];
            mugn[j] = K.is_goal(j) ? s : 0.0;
        }
    }
    }
    K.sync();
    MT_(PF_M_MU);
    if (act) {  // d_k = d0 + D_k mu_g ; ct_k = c_k - Gam_k d_k
        double dk[m];
        double ml[SEG ? n : 1];   // (segmented solve: the multiplier of this knot's chain, taken once)
        if constexpr (SEG) {
            const int mo = seg_mult_off<MODEL, NCH>(k, N, K.P.ll.seg);
#pragma unroll
            for (int j = 0; j < n; j++) ml[j] = K.lds[mo + j];
            __builtin_amdgcn_sched_barrier(0);
        }
#pragma unroll
        for (int i = 0; i < m; i++) {
            double s = d0[i];
#pragma unroll
            for (int j = 0; j < n; j++) {
                double mlt;
                if constexpr (SEG) mlt = ml[j]; else mlt = mugn[j];
                s += (KEEP_D ? Dk[i * n + j] : K.kd(k, R::oD + i * n + j)) * mlt;
            }
            dk[i] = s;
            K.dv_(k, i) = s;
        }
        double Gamk[n * m];
        if (k >= 1) {
            if constexpr (T::PG2 || (!T::LTI && n > 8 && T::NDEF == 0)) {
                double Mk_[n * n];
                load_M_Gam(K, k, Mk_, Gamk);
            } else {
                const auto pg = K.PGk(k);
#pragma unroll
                for (int i = 0; i < n; i++)
#pragma unroll
                    for (int j = 0; j < m; j++) Gamk[i * m + j] = T::Gnz(i, j) ? pg[i * NZ + n + j] : 0.0;
            }
        } else {
            Dyn<MODEL>::B(K.P.mp, Gamk);
#pragma unroll
            for (int i = 0; i < n * m; i++) Gamk[i] *= hdt;
            if constexpr (T::NDEF > 0) {
#pragma unroll
                for (int i = 0; i < T::NDEF; i++) Gamk[i * m + (m - T::NDEF) + i] = 1.0;
            }
        }
#pragma unroll
        for (int i = 0; i < n; i++) {
            double s = K.cv[k * n + i];
#pragma unroll
            for (int l = 0; l < m; l++) if (T::Gnz(i, l)) s -= Gamk[i * m + l] * dk[l];
            K.dY[k * n + i] = s;
        }
    }
    MT_(PF_M_DK);
    K.sync();
    MT_(PF_M_SYNC);
#undef MT_
}
template <int MODEL, class BLK, int NCH = 0> __device__ __noinline__ void mid_phase_call(typename BLK::Args a, int k, bool act, double hdt, Prof* pf) {
    BLK K(a, gusto_dyn_lds);
    using C = typename BLK::C;
    mid_phase<MODEL, BLK, NCH>(K, k, act, hdt, gusto_dyn_lds + C::misc, gusto_dyn_lds + C::misc + 48, pf);
}

// The phase after the forward sweep of a right-hand side: primal step of this knot, the new costates, row steps and
// the fraction to the boundary (before the workgroup reductions).  A function of its own for MT::SWEEP_CALL models.
struct StepOut { double amax, c0, c1, c2; };
#if GUSTO_SEG_W2   // (segw.hpp, included further down: the row passes below share a knot's obstacle rows with the helper waves)
template <int MODEL, int NCH, class BLK> GD void segw_post_rows(BLK& K, int cmd, double a0, double kappa, double omega, double Delta, double mu_t, double tau);
template <int MODEL, int NCH, class Op> GD void segw_rows_resid_add(const LPtr<double> L, int sb, int k, int nhelp, Op& op, double* Hx, double* rdx, double* gx0);
template <int MODEL, int NCH, class Op> GD void segw_rows_step_add(const LPtr<double> L, int sb, int k, int h0, Op& op, double* gAx, double* gBx);
GD void segw_join();
#endif
// x_1 stationarity, gx_0 + nu_0 + F_0^T nu_1 = 0: the costate of the first knot from the one behind it (ONE lane's work: lane 0 of
// the step phase, or of the helper wave's costate pass)
template <int MODEL, class BLK> GD void costate_close_x1(BLK& K, double hdt, const double* gxs) {
    using T = MT<MODEL>;
    constexpr int n = T::n, m = T::m;
    double Ad[n * n], x0[n], u0[m];
#pragma unroll
    for (int i = 0; i < n; i++) x0[i] = K.Xp[i];
#pragma unroll
    for (int i = 0; i < m; i++) u0[i] = K.Up[i];
    if constexpr (BLK::C::LC_LDS) {
        const double c2[2] = {K.lcl[0], K.lcl[1]};
        Dyn<MODEL>::A_cached(c2, Ad);
    } else Dyn<MODEL>::A(K.P.mp, x0, u0, Ad);
#pragma unroll
    for (int i = 0; i < n; i++) {
        double s = gxs[i] + K.nun[n + i];
#pragma unroll
        for (int j = 0; j < n; j++) if (T::Anz(j, i)) s += hdt * Ad[j * n + i] * K.nun[n + j];
        K.nun[i] = -s;
    }
}
template <int MODEL, class BLK, bool CLOSE = true, int NCH = 0>
GD StepOut step_phase(BLK& K, const RowCtx<MODEL>& ctx, const RowState& rs, int k, bool act, int pass, int ncomp, double hdt,
                      double tau, double mu_t, const double* mugn, const double* gxs) {
    using T = MT<MODEL>;
    using R = Rec<MODEL>;
    constexpr int n = T::n, m = T::m;
    constexpr bool ADJ = costate_adjoint<MODEL>() && BLK::ONE && NCH == 0;   // new costates by the adjoint recursion (adjoint_sweep_1w; never with a wave per chain)
    bool adj_rt = false;   // ... unless the horizon is too coarse for it or the solve has run long (P | Pi records then)
    if constexpr (ADJ) adj_rt = costate_adjoint_now<MODEL>();
    const int N = K.N;
    double l_amax = 1.0, l_c0 = 0, l_c1 = 0, l_c2 = 0;
    if (act) {
        double dxs[n], dus[m], dyp[n];
#pragma unroll
        for (int i = 0; i < n; i++) { dyp[i] = (k >= 1) ? K.dY[(k - 1) * n + i] : 0.0; dxs[i] = 0; }
#pragma unroll
        for (int i = 0; i < m; i++) {
            double s = -K.dv_(k, i);
#pragma unroll
            for (int l = 0; l < n; l++) s -= K.kd(k, R::oK + i * n + l) * dyp[l];
            dus[i] = s;
            K.dUs_(k, i) = s;
        }
        if (k >= 1) {
            double a[n], Bd[n * m], Mk[n * n], Gamk[n * m];
            Dyn<MODEL>::B(K.P.mp, Bd);
            load_M_Gam(K, k, Mk, Gamk);
#pragma unroll
            for (int i = 0; i < n; i++) {
                double s = dyp[i] + K.rd_(k, i);
#pragma unroll
                for (int l = 0; l < m; l++) if (T::Bnz(i, l)) s += (hdt * Bd[i * m + l]) * dus[l];
                a[i] = s;
            }
#pragma unroll
            for (int i = 0; i < n; i++) {
                double s = 0;
#pragma unroll
                for (int l = 0; l < n; l++) if (T::Mnz(i, l)) s += Mk[i * n + l] * a[l];
                dxs[i] = s;
            }
        }
#pragma unroll
        for (int i = 0; i < n; i++) K.dXs_(k, i) = dxs[i];
        // nu_{k+1} = P_k dy_k + p_k + Pi_k mu_g  (one-wave 12/13-state models: costate_pass_1w has done it)
        if constexpr (!(BLK::ONE && T::SWEEP_CALL))
        if (k + 1 < N && (pass == 1 || ncomp == 0)) {
#pragma unroll
            for (int i = 0; i < n; i++) {
                double s = K.pv[k * n + i] - K.rv[k * n + i];
#pragma unroll
                for (int l = 0; l < n; l++)
                    if constexpr (BLK::C::KD_LDS)   // (packed record: upper triangle of P, then Pi; factor_sweep_1w)
                        s += K.pprec(k)[sidx(i, l, n)] * K.dY[k * n + l] + K.pprec(k)[n * (n + 1) / 2 + i * n + l] * mugn[l];
                    else
                    s += K.Paft[(size_t)k * R::SNN + i * n + l] * K.dY[k * n + l] +
                         K.Piaft[(size_t)k * R::SNN + i * n + l] * mugn[l];
                K.nun[(k + 1) * n + i] = s;
            }
        }
        double xs[n], us[m];
#pragma unroll
        for (int i = 0; i < n; i++) xs[i] = K.Xw[k * n + i];
#pragma unroll
        for (int i = 0; i < m; i++) us[i] = K.Uw[k * m + i];
        double gAx[n], gBx[n], gAu[m], gBu[m];   // (pass 0) row part of the corrector's right-hand side, see OpStep
#pragma unroll
        for (int i = 0; i < n; i++) { gAx[i] = 0; gBx[i] = 0; }
#pragma unroll
        for (int i = 0; i < m; i++) { gAu[i] = 0; gBu[i] = 0; }
        // (small models: the state of the rows every knot has, in one batch of loads -- row by row, each row's loads wait
        // behind the stores of the row before it and the pass pays one memory round trip per row)
        constexpr int NP = (n <= 8 && T::NDEF == 0) ? T::NFIX + T::NHU : 0;
        RowPre<NP> pre;
        if constexpr (NP > 0) {
            const int slot_u = T::NFIX + K.P.n_obs + 2 * n;
            pre.load(rs, T::NFIX, slot_u, [&](int var) {
                return var == RS_T || var == RS_LAM || var == RS_S || var == RS_LAMB || (pass && (var == RS_KA || var == RS_KB));
            });
        }
        double hdx[ADJ ? n : 1];
#pragma unroll
        for (int i = 0; i < (ADJ ? n : 1); i++) hdx[i] = 0;
        OpStep<NP, RowState, ADJ> op{rs, dxs, dus, pass, mu_t, tau, gAx, gAu, gBx, gBu, &pre, hdx};
        ctx.tick(0);
#if GUSTO_SEG_W2
        if constexpr (NCH > 0) {
            // the knot's obstacle rows shared with the helper waves (segw.hpp); a pass that ends with new costates gives helper 1 those
            // instead.  (The step dxs / dus is in the workspace: the post drains this wave's stores.)
            const bool cs = pass == 1 || ncomp == 0;
            segw_post_rows<MODEL, NCH>(K, cs ? SEGW_STEP_CS : SEGW_STEP, (double)pass, ctx.kappa, ctx.omega, ctx.Delta, mu_t, tau);
            RowCtx<MODEL> cm = ctx;
            // (the helpers take all obstacle rows -- unless the only one has the costates first: then half of them stay here)
            cm.mask = (cs && NCH == 2) ? (cm.mask & seg_obs_share(0, 2)) : 0;
            visit_rows<MODEL>(cm, xs, us, op);
            segw_join();
            segw_rows_step_add<MODEL, NCH>(K.lds, K.P.ll.seg, k, (cs && NCH > 2) ? 1 : 0, op, gAx, gBx);
        } else
#endif
        visit_rows<MODEL>(ctx, xs, us, op);
        ctx.tick(3);
        if constexpr (ADJ) {
            // v_k = M_k^T (H_x dx_k + gx_k [+ mu_g at the last knot]) -> nun[k], the inhomogeneity of nu_k = Phi_k^T nu_{k+1} + v_k:
            // stationarity in x_k of the Newton system, H_x dx_k + gx_k + F_k^T nu_{k+1} - G_k^T nu_k = 0 with F = I + dt/2 A,
            // G = I - dt/2 A = M^-1 (resid_phase: "+ E^T nu"), gx_k the row part of this right-hand side (RHS phase)
            if (adj_rt && k >= 1 && (pass == 1 || ncomp == 0)) {
                double Mk[n * n], Gamk[n * m], w[n];
                load_M_Gam(K, k, Mk, Gamk);
#pragma unroll
                for (int i = 0; i < n; i++) {
                    double s = hdx[i] + (pass ? K.gAx_(k, i) + mu_t * K.gBx_(k, i) : 0.0);
                    if (k == N - 1 && K.is_goal(i)) s += mugn[i];
                    w[i] = s;
                }
#pragma unroll
                for (int i = 0; i < n; i++) {
                    double s = 0;
#pragma unroll
                    for (int l = 0; l < n; l++) if (T::Mnz(l, i)) s += Mk[l * n + i] * w[l];
                    K.nun[k * n + i] = s;
                }
            }
        }
        l_amax = op.amax.value(); l_c0 = op.c0; l_c1 = op.c1; l_c2 = op.c2;
        if (pass == 0) {
#pragma unroll
            for (int i = 0; i < n; i++) { K.gAx_(k, i) = gAx[i]; K.gBx_(k, i) = gBx[i]; }
#pragma unroll
            for (int i = 0; i < m; i++) { K.gAu_(k, i) = gAu[i]; K.gBu_(k, i) = gBu[i]; }
        }
    }
    K.sync();
    // (adjoint costates: nu_1 is known after the sweep; adjoint_sweep_1w_call closes with this block.  Two waves per problem, CLOSE =
    // false: the helper wave's costate pass runs beside this phase and closes itself -- seg.hpp: segw_helper)
    if constexpr (CLOSE)
    if (!(ADJ && adj_rt) && k == 0 && (pass == 1 || ncomp == 0)) costate_close_x1<MODEL>(K, hdt, gxs);
    return StepOut{l_amax, l_c0, l_c1, l_c2};
}
// the row context of knot k, as ipm_solve builds it (the called phases rebuild theirs from three numbers)
template <int MODEL, class BLK>
GD void make_row_ctx(const BLK& K, int k, bool act, double kappa, double omega, double Delta, RowCtx<MODEL>& ctx, RowState& rs) {
    constexpr int n = BLK::n;
    ctx.P = &K.P; ctx.N = K.N; ctx.k = k; ctx.nslot = K.P.wl.nslot; ctx.kappa = kappa; ctx.omega = omega; ctx.Delta = Delta;
    ctx.xp = K.Xp + (act ? k : 0) * n; ctx.mask = act ? K.obs_mask[k] : 0; ctx.obs_nh = K.obs_nh; ctx.obs_c0 = K.obs_c0;
    ctx.goal_lo = K.goal_lo; ctx.goal_hi = K.goal_hi; ctx.boxmask = K.boxmask;
    rs = RowState{K.rowstate, K.P.wl.nslot, K.N, act ? k : 0};
}
struct RowScal { double kappa, omega, Delta; };
template <int MODEL, class BLK, bool CLOSE = true, int NCH = 0>
__device__ __noinline__ StepOut step_phase_call(typename BLK::Args a, RowScal sc, int k, bool act, int pass, int ncomp,
                                                double hdt, double tau, double mu_t, Prof* pf) {
    BLK K(a, gusto_dyn_lds);
    using C = typename BLK::C;
    RowCtx<MODEL> ctx;
    RowState rs;
    make_row_ctx<MODEL>(K, k, act, sc.kappa, sc.omega, sc.Delta, ctx, rs);
#ifdef GUSTO_PROFILE
    ctx.pf = pf; ctx.pfb = PF_S0;
#endif
    return step_phase<MODEL, BLK, CLOSE, NCH>(K, ctx, rs, k, act, pass, ncomp, hdt, tau, mu_t, gusto_dyn_lds + C::misc + 48,
                                              gusto_dyn_lds + C::misc + 16);
}

// The residual phase of an interior point iteration: residuals, condensed Hessian blocks, dual residual, the predictor's
// row sums and the LQR stage cost QQ_k of this knot (before the workgroup reductions).  A function of its own for
// MT::SWEEP_CALL models.
struct ResidOut { double resp, resd, comp, numax; };
template <int MODEL, class BLK, int NCH = 0>
GD ResidOut resid_phase(BLK& K, const RowCtx<MODEL>& ctx, const RowState& rs, int k, bool act, double hdt, double wk,
                        double alpha_prev, const double* mug) {
    using T = MT<MODEL>;
    using R = Rec<MODEL>;
    constexpr int n = T::n, m = T::m, NZ = n + m, NHX = n * (n + 1) / 2, NHU = m * (m + 1) / 2, NQ = NZ * (NZ + 1) / 2;
    const int N = K.N;
    double l_resp = 0, l_resd = 0, l_comp = 0, l_numax = 0;
    if (act) {
        double xs[n], us[m], Hx[NHX], Hu[NHU], rdx[n], rdu[m], rdk[n], Ad[n * n], Bd[n * m];
#pragma unroll
        for (int i = 0; i < n; i++) xs[i] = K.Xw[k * n + i];
#pragma unroll
        for (int i = 0; i < m; i++) us[i] = K.Uw[k * m + i];
#pragma unroll
        for (int i = 0; i < NHX; i++) Hx[i] = 0;
#pragma unroll
        for (int i = 0; i < NHU; i++) Hu[i] = 0;
#pragma unroll
        for (int i = 0; i < n; i++) { rdx[i] = 0; rdk[i] = 0; }
#pragma unroll
        for (int i = 0; i < m; i++) rdu[i] = 0;
        {
            double xpk[n], upk[m];
#pragma unroll
            for (int i = 0; i < n; i++) xpk[i] = K.Xp[k * n + i];
#pragma unroll
            for (int i = 0; i < m; i++) upk[i] = K.Up[k * m + i];
            if constexpr (BLK::C::LC_LDS) {
                const double c2[2] = {K.lcl[2 * k], K.lcl[2 * k + 1]};
                Dyn<MODEL>::A_cached(c2, Ad);
            } else Dyn<MODEL>::A(K.P.mp, xpk, upk, Ad);
            Dyn<MODEL>::B(K.P.mp, Bd);
        }
        if (k >= 1) {
#pragma unroll
            for (int i = 0; i < n; i++) {
                rdk[i] = K.Xw[(k - 1) * n + i] - xs[i] + hdt * (K.pv[(k - 1) * n + i] + K.pv[k * n + i]);
                if constexpr (T::NDEF > 0) rdk[i] += K.Uw[(k - 1) * m + (m - T::NDEF) + i];   // + d_{k-1}: the defect of the interval
                l_resp = nanmax(l_resp, fabs(rdk[i]));
            }
        }
#pragma unroll
        for (int i = 0; i < n; i++) K.rd_(k, i) = rdk[i];
        double gx0[n], gu0[m];
#pragma unroll
        for (int i = 0; i < n; i++) gx0[i] = 0;
#pragma unroll
        for (int i = 0; i < m; i++) gu0[i] = 0;
        // (small models only: the 12/13-state kernels are far beyond the register file already -- 5 KB of scratch
        // per lane and > 1200 spilled SGPRs -- and more live values there have produced wrong code)
        constexpr int NP = (n <= 8 && T::NDEF == 0) ? T::NFIX + T::NHU : 0;
        RowPre<NP> pre;
        if constexpr (NP > 0) {
            const int slot_u = T::NFIX + K.P.n_obs + 2 * n;
            const bool upd = alpha_prev != 0.0;
            pre.load(rs, T::NFIX, slot_u, [&](int var) {
                return var == RS_T || var == RS_LAM || var == RS_S || var == RS_LAMB ||
                       (upd && (var == RS_DT || var == RS_DL || var == RS_DS));
            });
        }
        constexpr bool LRTR = MODEL == GUSTO_ASTROBEE_SE3;   // (the manifold model has no trust region row)
        OpResidHess<n, m, NP, LRTR> op{rs, Hx, Hu, rdx, rdu, gx0, gu0, alpha_prev, &pre};
        ctx.tick(0);   // (profile builds: PF_R0.. = prologue | fixed rows | obstacle rows | control rows | stage cost)
#if GUSTO_SEG_W2
        if constexpr (NCH > 0) {   // the knot's obstacle rows shared with the helper waves (segw.hpp)
            segw_post_rows<MODEL, NCH>(K, SEGW_ROWS_R, alpha_prev, ctx.kappa, ctx.omega, ctx.Delta, 0.0, 0.0);
            RowCtx<MODEL> cm = ctx;
            cm.mask = 0;   // (the helpers take all obstacle rows)
            visit_rows<MODEL>(cm, xs, us, op);
            segw_join();
            segw_rows_resid_add<MODEL, NCH>(K.lds, K.P.ll.seg, k, NCH - 1, op, Hx, rdx, gx0);
        } else
#endif
        visit_rows<MODEL>(ctx, xs, us, op);
        ctx.tick(3);
        // row part of the predictor right-hand side, parked in the (currently free) step arrays
#pragma unroll
        for (int i = 0; i < n; i++) K.dXs_(k, i) = gx0[i];
#pragma unroll
        for (int i = 0; i < m; i++) K.dUs_(k, i) = gu0[i];
        l_comp = op.comp;
        l_resp = nanmax(l_resp, op.maxrp);
#pragma unroll
        for (int i = 0; i < m; i++) {   // (TrajOpt: the defects carry REG times the control cost)
            const double wi_ = (i < m - T::NDEF) ? wk : TRAJOPT_DEFECT_REG * wk;
            Hu[sidx(i, i, m)] += 2 * wi_; rdu[i] += 2 * wi_ * us[i];
        }
        // + E^T nu: F_k^T nu_{k+1} - G_k^T nu_k on x, b_k^T (nu_{k+1} + nu_k) on u
        {
            double vs[n], vd[n];
#pragma unroll
            for (int i = 0; i < n; i++) {
                const double n1 = (k + 1 < N) ? K.nu[(k + 1) * n + i] : 0.0, n0 = (k >= 1) ? K.nu[k * n + i] : 0.0;
                vs[i] = n1 + n0; vd[i] = n1 - n0;
                l_numax = fmax(l_numax, fabs(K.nu[k * n + i]));
            }
#pragma unroll
            for (int i = 0; i < n; i++) {
                double s = vd[i];
#pragma unroll
                for (int j = 0; j < n; j++) if (T::Anz(j, i)) s += hdt * Ad[j * n + i] * vs[j];
                rdx[i] += s;
            }
#pragma unroll
            for (int i = 0; i < m; i++) {
                double s = 0;
#pragma unroll
                for (int j = 0; j < n; j++) if (T::Bnz(j, i)) s += hdt * Bd[j * m + i] * vs[j];
                rdu[i] += s;
            }
            if constexpr (T::NDEF > 0) {   // d_k sits in the trapezoid row of the interval (k, k+1) only: + nu_{k+1}
#pragma unroll
                for (int i = 0; i < T::NDEF; i++) rdu[(m - T::NDEF) + i] += (k + 1 < N) ? K.nu[(k + 1) * n + i] : 0.0;
            }
        }
        if (k == N - 1) {
#pragma unroll
            for (int i = 0; i < n; i++) {
                if (K.is_goal(i)) {
                    rdx[i] += mug[i];
                    l_resp = nanmax(l_resp, fabs(K.misc[64 + i] - xs[i]));
                }
            }
        }
        if (k >= 1) {
#pragma unroll
            for (int i = 0; i < n; i++) l_resd = nanmax(l_resd, fabs(rdx[i]));
        }
#pragma unroll
        for (int i = 0; i < m; i++) l_resd = nanmax(l_resd, fabs(rdu[i]));

        // (3) QQ = [Qt, Qt b; ., Hu + b^T Qt b], Qt = M^T Hx M (built even on the last trip: cheap)
        {   // knot 0 (x_1 is pinned: Qt = 0, only H_u survives) runs the SAME code with M := 0 -- as an else-branch it was
            // NQ + n stores issued for one lane while the other 49 waited
            double Mk[n * n], Qt[NHX], Qb[n * m];
            {
                double Gamk[n * m];
                load_M_Gam(K, k, Mk, Gamk);
                const double mz = (k >= 1) ? 1.0 : 0.0;
#pragma unroll
                for (int i = 0; i < n; i++)
#pragma unroll
                    for (int j = 0; j < n; j++) if (T::Mnz(i, j)) Mk[i * n + j] *= mz;
            }
            // H_x = (block diagonal part, T::Hnz) + diag(trh) + trs * trg trg^T (LRTR): the dyad goes through M as a
            // vector, M^T (trs g g^T) M = trs (M^T g)(M^T g)^T, and the products below skip the structural zeros of H_x
            double gt[LRTR ? n : 1];
            if constexpr (LRTR) {
#pragma unroll
                for (int i = 0; i < n; i++) {
                    Hx[sidx(i, i, n)] += op.trh[i];
                    double s = 0;
#pragma unroll
                    for (int l = 0; l < n; l++) if (T::Mnz(l, i)) s += Mk[l * n + i] * op.trg[l];
                    gt[i] = s;
                }
            }
#pragma unroll
            for (int j = 0; j < n; j++) {  // column j of Hx M, then column j of the upper triangle of M^T (Hx M)
                double tcol[n];
                bool tnz[n];   // (compile-time after unrolling: entry i of the column is structurally nonzero)
#pragma unroll
                for (int i = 0; i < n; i++) {
                    double s = 0;
                    tnz[i] = false;
#pragma unroll
                    for (int l = 0; l < n; l++)
                        if (T::Mnz(l, j) && T::Hnz(i, l)) { s += Hx[sidx(i, l, n)] * Mk[l * n + j]; tnz[i] = true; }
                    tcol[i] = s;
                }
#pragma unroll
                for (int i = 0; i <= j; i++) {
                    double s = 0;
#pragma unroll
                    for (int l = 0; l < n; l++) if (T::Mnz(l, i) && tnz[l]) s += Mk[l * n + i] * tcol[l];
                    if constexpr (LRTR) s += op.trs * gt[i] * gt[j];
                    Qt[sidx(i, j, n)] = s;
                    K.qq_put(k, sidx(i, j, NZ), s);
                }
            }
#pragma unroll
            for (int i = 0; i < n; i++)
#pragma unroll
                for (int j = 0; j < m; j++) {
                    double s = 0;
#pragma unroll
                    for (int l = 0; l < n; l++) if (T::Bnz(l, j)) s += Qt[sidx(i, l, n)] * (hdt * Bd[l * m + j]);
                    Qb[i * m + j] = s;
                    K.qq_put(k, sidx(i, n + j, NZ), s);
                }
#pragma unroll
            for (int i = 0; i < m; i++)
#pragma unroll
                for (int j = i; j < m; j++) {
                    double s = Hu[sidx(i, j, m)];
#pragma unroll
                    for (int l = 0; l < n; l++) if (T::Bnz(l, i)) s += (hdt * Bd[l * m + i]) * Qb[l * m + j];
                    K.qq_put(k, sidx(n + i, n + j, NZ), s);
                }
#pragma unroll
            for (int i = 0; i < n; i++) {
                double s = 0, c = 0.0 - rdk[i];
#pragma unroll
                for (int l = 0; l < n; l++) { s += Qt[sidx(i, l, n)] * rdk[l]; if (T::Mnz(i, l)) c += 2.0 * Mk[i * n + l] * rdk[l]; }
                K.qrd_(k, i) = s;
                K.cv[k * n + i] = c;
            }
        }
    }
    return ResidOut{l_resp, l_resd, l_comp, l_numax};
}
template <int MODEL, class BLK, int NCH = 0>
__device__ __noinline__ ResidOut resid_phase_call(typename BLK::Args a, RowScal sc, int k, bool act, double hdt, double wk,
                                                  double alpha_prev, Prof* pf) {
    BLK K(a, gusto_dyn_lds);
    using C = typename BLK::C;
    RowCtx<MODEL> ctx;
    RowState rs;
    make_row_ctx<MODEL>(K, k, act, sc.kappa, sc.omega, sc.Delta, ctx, rs);
#ifdef GUSTO_PROFILE
    ctx.pf = pf; ctx.pfb = PF_R0;
#endif
    return resid_phase<MODEL, BLK, NCH>(K, ctx, rs, k, act, hdt, wk, alpha_prev, gusto_dyn_lds + C::misc + 32);
}

// The KKT solve as Riccati segments joined by coarse LQR stages (round 6).  seg.hpp (-DGUSTO_SEG2=1, off): two chains interleaved
// in ONE wave, freeflyerSE2 -- parity-green, slower: the sequential phases are ISSUE-bound, not latency-bound
// (profiles/r06_two_chains.txt).  segw.hpp (on): a WAVE per chain, the matrix-core kernels.
}  // namespace gusto
#if GUSTO_SEG2
#include "seg.hpp"
#else
namespace gusto { template <int MODEL> constexpr bool seg2_model() { return false; } }
#endif
#if GUSTO_SEG_W2
#include "segw.hpp"
#endif
namespace gusto {

// ---- the interior point method ---------------------------------------------------------------------
// Register discipline: nothing per-thread stays live across a sequential sweep.  Every stage-parallel block
// re-reads the iterate (Xw/Uw), the linearisation point (Xp/Up) and the stage matrices it needs from LDS / L2
// and leaves its results in LDS, so the sweeps get the whole register file for latency hiding.
template <int MODEL, class BLK, int NCH = 0> GD void ipm_solve(BLK& K, double Delta, double omega, double muw, IpmOut& out, Prof& pf) {
    using T = MT<MODEL>;
    using R = Rec<MODEL>;
    constexpr int n = T::n, m = T::m, NZ = n + m, NHX = n * (n + 1) / 2, NHU = m * (m + 1) / 2, NQ = NZ * (NZ + 1) / 2;
    int k = K.tid;
    const int N = K.N;
    const bool act = k < N;
    const gusto_ipm_opts& io = K.P.io;
    const double kappa = 1.0 / fmax(1.0, omega);
    const double wk = kappa * ((k == 0 || k == N - 1) ? 0.5 * K.dt : K.dt);
    const double hdt = 0.5 * K.dt;
    double* red = K.misc;        // [0..7] block_reduce scratch
    double* fail = K.misc + 8;   // factorisation failure flag
    double* gxs = K.misc + 16;   // gx of knot 0 (n values)
    double* mug = K.misc + 32;   // goal multipliers (state-index space)
    double* mugn = K.misc + 48;  // ... of the current Newton step
    // the horizon split into Riccati segments: NCH chains, a wave each (segw.hpp: scp_kernel_w2, the matrix-core kernels; launch_scp
    // starts it for N >= NCH GUSTO_SEG_MIN_N only), or two chains in the one wave (seg.hpp: A = stages 0 .. seg_s - 1, B = seg_s .. N - 1)
    constexpr bool SEGB = seg2_big<MODEL>() && BLK::ONE && NCH > 0;
    constexpr bool SEG = (seg2_model<MODEL>() && BLK::ONE) || SEGB;
    [[maybe_unused]] const bool seg = SEG && (SEGB || N >= 2 * GUSTO_SEG_MIN_N);
    [[maybe_unused]] const int seg_s = seg_split(N);
#if GUSTO_SEG_W2
    if constexpr (SEGB) segw_open<MODEL, NCH>(K);
#endif

    RowCtx<MODEL> ctx;
    ctx.P = &K.P; ctx.N = N; ctx.k = k; ctx.nslot = K.P.wl.nslot; ctx.kappa = kappa; ctx.omega = omega; ctx.Delta = Delta;
    ctx.xp = K.Xp + (act ? k : 0) * n; ctx.mask = act ? K.obs_mask[k] : 0; ctx.obs_nh = K.obs_nh; ctx.obs_c0 = K.obs_c0;
    ctx.goal_lo = K.goal_lo; ctx.goal_hi = K.goal_hi; ctx.boxmask = K.boxmask;
    RowState rs{K.rowstate, K.P.wl.nslot, N, act ? k : 0};
    // The knot index is made opaque at every phase boundary: otherwise the compiler hoists each phase's address
    // arithmetic out of the interior point loop and its registers (hundreds) stay live across the sweeps.
#define GUSTO_REFRESH_K()                                     \
    do {                                                      \
        asm volatile("" : "+v"(k));                           \
        ctx.k = k; rs.k = act ? k : 0;                        \
        ctx.xp = K.Xp + (act ? k : 0) * n;                    \
    } while (0)

    auto load_iter = [&](double* xs, double* us) {
#pragma unroll
        for (int i = 0; i < n; i++) xs[i] = K.Xw[k * n + i];
#pragma unroll
        for (int i = 0; i < m; i++) us[i] = K.Uw[k * m + i];
    };

    // warm start at traj_prev (scp_gusto.jl:100-102) with x_1 pinned to x_init; slacks interior
    double ncomp_l = 0;
    if (act) {
        double xs[n], us[m];
#pragma unroll
        for (int i = 0; i < n; i++) { xs[i] = (k == 0) ? K.x_init[i] : K.Xp[k * n + i]; K.Xw[k * n + i] = xs[i]; K.nu[k * n + i] = 0; }
#pragma unroll
        for (int i = 0; i < m; i++) { us[i] = K.Up[k * m + i]; K.Uw[k * m + i] = us[i]; }
        OpInit op{rs, muw};
        visit_rows<MODEL>(ctx, xs, us, op);
        ncomp_l = op.ncomp;
    }
    if (k == 0) {
        *fail = 0.0;
        for (int i = 0; i < n; i++) { mug[i] = 0; mugn[i] = 0; }
    }
    const double ncomp = block_reduce<BLK::ONE>(ncomp_l, OpSum(), red);
    K.sync();
    pf.tick(PF_INIT);

    constexpr bool LINC = BLK::C::LC_LDS;
    if constexpr (LINC) {   // (the linearisation point is fixed for the solve: its sin / cos once, two numbers per knot in LDS)
        if (act) {
            double xpk[n], c2[2];
#pragma unroll
            for (int i = 0; i < n; i++) xpk[i] = K.Xp[k * n + i];
            Dyn<MODEL>::lin_cache(K.P.mp, xpk, c2);
            K.lcl[2 * k] = c2[0]; K.lcl[2 * k + 1] = c2[1];
        }
        K.sync();
    }
    int status = GUSTO_SOLVER_FAILED, it = 0, n_acc = 0;
    bool adj_ok = true;   // the adjoint costates of the 12/13-state kernels are still in use (see below)
    if constexpr (SEGB) adj_ok = false;   // (the segmented solve takes its costates from the P | Pi records: a record (0 | I) in front of every interface)
    double res_p = 0, res_d = 0, mu = 0, alpha_prev = 0.0, mu_start = 0.0;
    for (it = 0;; it++) {
        GUSTO_REFRESH_K();
        // (1) linearised xdot at each knot: a_k = f_k + A_k (x_k - xp_k) + B (u_k - up_k)
        if (act) {
            double xs[n], us[m], xpk[n], upk[m], fp[n], Ad[n * n], Bd[n * m];
            load_iter(xs, us);
#pragma unroll
            for (int i = 0; i < n; i++) xpk[i] = K.Xp[k * n + i];
#pragma unroll
            for (int i = 0; i < m; i++) upk[i] = K.Up[k * m + i];
            if constexpr (LINC) {
                const double c2[2] = {K.lcl[2 * k], K.lcl[2 * k + 1]};
                Dyn<MODEL>::f_cached(K.P.mp, c2, upk, fp);
                Dyn<MODEL>::A_cached(c2, Ad);
            } else {
            Dyn<MODEL>::f(K.P.mp, xpk, upk, fp);
            Dyn<MODEL>::A(K.P.mp, xpk, upk, Ad);
            }
            Dyn<MODEL>::B(K.P.mp, Bd);
#pragma unroll
            for (int i = 0; i < n; i++) {
                double s = fp[i];
#pragma unroll
                for (int j = 0; j < n; j++) if (T::Anz(i, j)) s += Ad[i * n + j] * (xs[j] - xpk[j]);
#pragma unroll
                for (int j = 0; j < m; j++) if (T::Bnz(i, j)) s += Bd[i * m + j] * (us[j] - upk[j]);
                K.pv[k * n + i] = s;
            }
        }
        K.sync();
        // (2) residuals, condensed Hessian blocks, dual residual; (3) the LQR stage cost of this knot
        ResidOut ro;
#ifdef GUSTO_PROFILE
        ctx.pf = &pf; ctx.pfb = PF_R0;
#endif
        if constexpr (T::SWEEP_CALL) ro = resid_phase_call<MODEL, BLK, (SEGB ? NCH : 0)>(K.args(), RowScal{kappa, omega, Delta}, k, act, hdt, wk, alpha_prev, &pf);
        else ro = resid_phase<MODEL>(K, ctx, rs, k, act, hdt, wk, alpha_prev, mug);
        const double l_resp = ro.resp, l_resd = ro.resd, l_comp = ro.comp, l_numax = ro.numax;
        res_p = block_reduce<BLK::ONE>(l_resp, OpNanMax(), red);
        res_d = block_reduce<BLK::ONE>(l_resd, OpNanMax(), red);
        const double comp = block_reduce<BLK::ONE>(l_comp, OpSum(), red);
        const double numax = block_reduce<BLK::ONE>(l_numax, OpMax(), red);
        mu = ncomp > 0 ? comp / ncomp : 0.0;
        K.sync();
        pf.tick(PF_RESID);
        if (res_p <= io.tol && res_d <= io.tol * (1 + numax) && mu <= 0.1 * io.tol) { status = GUSTO_SOLVER_OPTIMAL; break; }
        const bool acceptable = res_p <= io.tol_acc && res_d <= io.tol_acc * (1 + numax) && mu <= io.tol_acc;
        n_acc = acceptable ? n_acc + 1 : 0;   // (gusto_ipm_opts.acc_iter: a solve that cycles at the acceptable level does not run to the cap)
        if (it >= io.max_iter || (io.acc_iter > 0 && n_acc >= io.acc_iter)) {
            if (acceptable) status = GUSTO_SOLVER_ALMOST;
            break;
        }
        if (!isfinite(res_p) || !isfinite(res_d) || !isfinite(mu)) break;
        if (it == 0) mu_start = mu;
        if (mu > IPM_DIVERGED * fmax(1.0, mu_start)) break;   // (diverging: an infeasible subproblem, common.hpp)
        pf.tick(PF_BUILD);
        // (the costates of this iteration: adjoint recursion or P | Pi records, see costate_adjoint_rt)
        bool adj_now = false;
        if constexpr (costate_adjoint<MODEL>() && BLK::ONE) {
            // (latched: a solve that has gone over to the P | Pi records stays there -- n_acc falls back to 0 whenever an iterate
            // leaves the acceptable level, and a solve must not flip between the two forms of its costates)
            adj_ok = adj_ok && it < GUSTO_ADJ_MAX_IT && n_acc < GUSTO_ADJ_ACC && costate_adjoint_rt<MODEL>(K.P.mp, K.dt);
            adj_now = adj_ok;
            if (k == 0) K.misc[ADJ_FLAG] = adj_now ? 1.0 : 0.0;
            K.sync();
        }
        // (4) factorise
        bool seg_done = false;
        [[maybe_unused]] bool seg_fail = false;
#if GUSTO_SEG_W2
        if constexpr (SEGB) {
            // chain A on the helper wave, chain B here; after the join the helper goes on with the coarse stage while this wave
            // builds the predictor's right-hand side -- the second join waits in front of the first backward sweep
            segw_post<MODEL, NCH>(K, SEGW_FACTOR);
            factor_sweep_seg_call<MODEL, NCH>(K.args(), &pf);
            segw_join();
            pf.tick(PF_FACTOR);
            GUSTO_REFRESH_K();
            seg_done = true;
            if (*fail != 0.0) { if constexpr (NCH == 4) segw_join(); segw_join(); break; }   // (the barriers of the waves stay paired)
        }
#endif
#if GUSTO_SEG2
        if constexpr (SEG && !SEGB) {
            if (seg) {
                factor_sweep_pg2s<MODEL>(SweepView<MODEL>::make(K), fail, pf, seg_s);
                pf.tick(PF_FACTOR);
                GUSTO_REFRESH_K();
                seg_coarse_factor<MODEL>(K, fail);
                seg_done = true;
            }
        }
#endif
        if (!seg_done) {
        factor_sweep<MODEL>(K, fail, pf);
        pf.tick(PF_FACTOR);
        GUSTO_REFRESH_K();
        if (k == 0) {  // Gd^-1 with an identity block on the coordinates without a point goal -> sP
            if constexpr (n <= 8) {
                double G[n * n], Li[n * n];
#pragma unroll
                for (int i = 0; i < n; i++)
#pragma unroll
                    for (int j = 0; j < n; j++) G[i * n + j] = (i == j && !K.is_goal(i)) ? 1.0 : K.sGd[i * n + j];
                if (!chol_inv<n>(G, Li)) *fail = 1.0;
#pragma unroll
                for (int i = 0; i < n; i++)
#pragma unroll
                    for (int j = 0; j < n; j++) {
                        double s2 = 0;
#pragma unroll
                        for (int l = 0; l < n; l++) if (l >= i && l >= j) s2 += Li[l * n + i] * Li[l * n + j];
                        K.sP[i * n + j] = s2;
                    }
            }
        }
        if constexpr (n > 8) inv_spd_block<MODEL>(K, fail);   // (the whole workgroup: one lane took 170 k cycles for n = 12)
        }
        if constexpr (!SEGB) {
        K.sync();
        if (*fail != 0.0) break;
        }
        pf.tick(PF_POSTF);

        // (5) predictor (mu_t = 0) and centred corrector share the factorisation
        double sigma = 0, mu_t = 0, alpha = 1.0;
        for (int pass = 0; pass < 2; pass++) {
            GUSTO_REFRESH_K();
            if (act) {  // right-hand side: qt_k = gy - K^T qu, qu = gu + b^T gy, gy = Qt rd + M^T gx
                double xs[n], us[m], gx[n], gu[m], quk[m], gy[n], Bd[n * m];
                load_iter(xs, us);
                Dyn<MODEL>::B(K.P.mp, Bd);
#pragma unroll
                for (int i = 0; i < n; i++) { gx[i] = 0; gy[i] = 0; }
#pragma unroll
                for (int i = 0; i < m; i++) gu[i] = 2 * ((i < m - T::NDEF) ? wk : TRAJOPT_DEFECT_REG * wk) * us[i];
                pf.tick(PF_F1);
                if (pass == 0) {   // the predictor's row sums were accumulated by the residual pass
#pragma unroll
                    for (int i = 0; i < n; i++) gx[i] = K.dXs_(k, i);
#pragma unroll
                    for (int i = 0; i < m; i++) gu[i] += K.dUs_(k, i);
                } else {   // the corrector's row sums were accumulated by the predictor's step pass: coef = A + mu_t B per row
#pragma unroll
                    for (int i = 0; i < n; i++) gx[i] = K.gAx_(k, i) + mu_t * K.gBx_(k, i);
#pragma unroll
                    for (int i = 0; i < m; i++) gu[i] += K.gAu_(k, i) + mu_t * K.gBu_(k, i);
                }
                pf.tick(PF_F2);
                if (k >= 1) {
                    double Mk[n * n], Gamk[n * m];
                    load_M_Gam(K, k, Mk, Gamk);
#pragma unroll
                    for (int i = 0; i < n; i++) {
                        double s = K.qrd_(k, i);
#pragma unroll
                        for (int l = 0; l < n; l++) if (T::Mnz(l, i)) s += Mk[l * n + i] * gx[l];
                        gy[i] = s;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < n; i++) gxs[i] = gx[i];
                }
                pf.tick(PF_F3);
#pragma unroll
                for (int i = 0; i < m; i++) {
                    double s = gu[i];
#pragma unroll
                    for (int l = 0; l < n; l++) if (T::Bnz(l, i)) s += (hdt * Bd[l * m + i]) * gy[l];
                    quk[i] = s;
                    K.qu_(k, i) = s;
                }
                {   // qt_k = gy - K^T qu (the record of K walked in storage order, see mid_phase)
                    double qt[n];
#pragma unroll
                    for (int i = 0; i < n; i++) qt[i] = gy[i];
#pragma unroll
                    for (int l = 0; l < m; l++)
#pragma unroll
                        for (int i = 0; i < n; i++) qt[i] -= K.kd(k, R::oK + l * n + i) * quk[l];
#pragma unroll
                    for (int i = 0; i < n; i++) K.pv[k * n + i] = qt[i] + (k >= 1 ? K.rv[(k - 1) * n + i] : 0.0);   // qq_k = qt_k + r_{k-1}
                }
                pf.tick(PF_F8);
            }
            K.sync();
            pf.tick(PF_RHS);
            // the two vector sweeps of this right-hand side and the phase between them: the chains' waves side by side (segw.hpp), two
            // chains in the one wave (seg.hpp), or the sequential recursion
            bool swp = false;
#if GUSTO_SEG_W2
            if constexpr (SEGB) {
                if (pass == 0) {   // the merges of this factorisation are done (four chains: the barrier between the tree's two levels first)
                    if constexpr (NCH == 4) segw_join();
                    segw_join();
                    pf.tick(PF_POSTF);   // (the wait for it)
                    if (*fail != 0.0) { seg_fail = true; break; }
                }
                segw_post<MODEL, NCH>(K, SEGW_BACK); backward_sweep_seg_call<MODEL, NCH>(K.args()); segw_join();
                pf.tick(PF_BACK);
                GUSTO_REFRESH_K();
                mid_phase_call<MODEL, BLK, NCH>(K.args(), k, act, hdt, &pf);
                pf.tick(PF_MID);
                segw_post<MODEL, NCH>(K, SEGW_FWD); forward_sweep_seg_call<MODEL, NCH>(K.args()); segw_join();
                swp = true;
            }
#endif
#if GUSTO_SEG2
            if constexpr (SEG && !SEGB) {
                if (seg) {
                    backward_sweep_seg<MODEL>(SweepView<MODEL>::make(K), seg_s);
                    pf.tick(PF_BACK);
                    GUSTO_REFRESH_K();
                    mid_phase_seg<MODEL>(K, k, act, hdt, seg_s, mugn, &pf);
                    pf.tick(PF_MID);
                    forward_sweep_seg<MODEL>(SweepView<MODEL>::make(K), seg_s);
                    swp = true;
                }
            }
#endif
            if (!swp) {
            backward_sweep<MODEL>(K);
            pf.tick(PF_BACK);
            GUSTO_REFRESH_K();
            if constexpr (T::SWEEP_CALL) mid_phase_call<MODEL, BLK>(K.args(), k, act, hdt, &pf);
            else mid_phase<MODEL>(K, k, act, hdt, red, mugn, &pf);
            pf.tick(PF_MID);
            forward_sweep<MODEL>(K);
            }
            if constexpr (BLK::ONE && T::SWEEP_CALL && !SEGB)
                if (!adj_now)
                if (pass == 1 || ncomp == 0) costate_pass_1w_call<MODEL>(K.args());
            pf.tick(PF_FWD);
            GUSTO_REFRESH_K();
            // primal step of this knot, the new costates, row steps + fraction to the boundary
            const double tau = pass ? fmax(0.995, 1.0 - mu) : 1.0;
            StepOut so;
#ifdef GUSTO_PROFILE
            ctx.pf = &pf; ctx.pfb = PF_S0;
#endif
#if GUSTO_SEG_W2
            if constexpr (SEGB)   // (the helper waves take obstacle rows and, in the pass that ends with them, the new costates: step_phase)
                so = step_phase_call<MODEL, BLK, false, NCH>(K.args(), RowScal{kappa, omega, Delta}, k, act, pass, ncomp, hdt, tau, mu_t, &pf);
            else
#endif
            if constexpr (T::SWEEP_CALL) so = step_phase_call<MODEL, BLK>(K.args(), RowScal{kappa, omega, Delta}, k, act, pass, ncomp, hdt, tau, mu_t, &pf);
            else {
                // (segmented solve: the costates of chain A's knots hang on dlam, those of chain B's on mu_g)
                const double* mult = mugn;
#if GUSTO_SEG2
                if constexpr (SEG && !SEGB) mult = (seg && k < seg_s) ? (const double*)(K.misc + (SegC<MODEL>::LAM - BLK::C::misc)) : mult;
#endif
                so = step_phase<MODEL>(K, ctx, rs, k, act, pass, ncomp, hdt, tau, mu_t, mult, gxs);
            }
            if constexpr (costate_adjoint<MODEL>() && BLK::ONE)
                if (adj_now && (pass == 1 || ncomp == 0)) adjoint_sweep_1w_call<MODEL>(K.args(), hdt);
            (void)adj_now;
            const double l_amax = so.amax, l_c0 = so.c0, l_c1 = so.c1, l_c2 = so.c2;
            const double a_max = block_reduce<BLK::ONE>(l_amax, OpMin(), red);
            alpha = a_max;
            if (pass == 0) {
                const double q0 = block_reduce<BLK::ONE>(l_c0, OpSum(), red), q1 = block_reduce<BLK::ONE>(l_c1, OpSum(), red),
                             q2 = block_reduce<BLK::ONE>(l_c2, OpSum(), red);
                const double ca = q0 + a_max * (q1 + a_max * q2);   // complementarity after the affine step
                const double mu_aff = ncomp > 0 ? ca / ncomp : 0.0;
                const double rr = (mu > 0) ? mu_aff / mu : 0.0;
                sigma = rr * rr * rr;
                if (io.sigma_max > 0) sigma = fmin(sigma, io.sigma_max);   // (gusto_hip.h: the corrector always aims at a real reduction of mu)
                mu_t = fmax(sigma * mu, io.mu_floor);
                if (ncomp == 0) break;  // equality-constrained QP: the predictor already is the Newton step
            }
        }
        if (seg_fail) break;
        pf.tick(PF_STEP);
        // (6) update
        GUSTO_REFRESH_K();
        K.sync();
        if (act) {
            double xs[n], us[m];
            load_iter(xs, us);
#pragma unroll
            for (int i = 0; i < n; i++) {
                xs[i] += alpha * K.dXs_(k, i);
                K.Xw[k * n + i] = xs[i];
                K.nu[k * n + i] += alpha * (K.nun[k * n + i] - K.nu[k * n + i]);
            }
#pragma unroll
            for (int i = 0; i < m; i++) { us[i] += alpha * K.dUs_(k, i); K.Uw[k * m + i] = us[i]; }
        }
        alpha_prev = alpha;   // the row state is advanced by the next residual pass
        if constexpr (n > 4) {
            if (k < n) mug[k] += alpha * (mugn[k] - mug[k]);
        } else if (k == 0) {
            for (int i = 0; i < n; i++) mug[i] += alpha * (mugn[i] - mug[i]);
        }
        K.sync();
        pf.tick(PF_UPDATE);
    }
    // JuMP.objective_value: cost + all slacks, in unscaled units
    double l_obj = 0;
    if (act) {
        double xs[n], us[m];
        load_iter(xs, us);
#pragma unroll
        for (int i = 0; i < m; i++) l_obj += ((i < m - T::NDEF) ? wk : TRAJOPT_DEFECT_REG * wk) * us[i] * us[i];
        OpSlackSum op{rs};
        visit_rows<MODEL>(ctx, xs, us, op);
        l_obj += op.sum;
    }
    const double obj = block_reduce<BLK::ONE>(l_obj, OpSum(), red);
    K.sync();
#undef GUSTO_REFRESH_K
    out.status = status; out.iters = it; out.obj = obj / kappa; out.res_p = res_p; out.res_d = res_d; out.mu = mu;
}

}  // namespace gusto
