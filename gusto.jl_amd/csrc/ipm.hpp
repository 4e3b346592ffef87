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
//   stage-parallel.  Goal point rows C x_N = g carry a multiplier mu_g whose sensitivities Pi_k = dp_k/dmu_g
//   ride along the factor sweep, so mu_g is known after the backward sweep.
//
// The reference hands this problem to JuMP -> Ipopt/Gurobi (scp_gusto.jl:82-104); the interior point method
// here is a Mehrotra predictor-corrector on the same problem (same optimum, it is strictly convex in U).
#pragma once
#include "rows.hpp"

namespace gusto {

struct IpmOut {
    int status, iters;
    double obj, res_p, res_d, mu;
};

template <int MODEL> struct Blk {
    using T = MT<MODEL>;
    static constexpr int n = T::n, m = T::m, NZ = n + m;
    const KParams& P;
    int b, tid, NT, N;
    double dt;
    // LDS
    double *Xw, *Uw, *Xp, *Up, *dY, *rd, *pv, *cv, *rv, *qrd, *nu, *nun, *qu, *dv;
    double *sP, *sPi, *sPG, *sT, *sHh, *sZ, *sK, *sD, *sW, *sV, *sGd, *misc;
    // global workspace of this problem
    double *rowstate, *obs_nh, *obs_c0, *PG, *QQ, *Paft, *Piaft, *Kg, *Sinvg, *Dg, *Phicl;
    uint64_t* obs_mask;
    const double *x_init, *goal_lo, *goal_hi;
    int ng, gidx[n];
    double gval[n];

    GD Blk(const KParams& P_, double* lds) : P(P_) {
        b = blockIdx.x; tid = threadIdx.x; NT = blockDim.x; N = P.N;
        const LdsLayout& L = P.ll;
        Xw = lds + L.Xw; Uw = lds + L.Uw; Xp = lds + L.Xp; Up = lds + L.Up; dY = lds + L.dY; rd = lds + L.rd;
        pv = lds + L.pv; cv = lds + L.cv; rv = lds + L.rv; qrd = lds + L.qrd; nu = lds + L.nu; nun = lds + L.nun;
        qu = lds + L.qu; dv = lds + L.dv;
        sP = lds + L.sP; sPi = lds + L.sPi; sPG = lds + L.sPG; sT = lds + L.sT; sHh = lds + L.sHh; sZ = lds + L.sZ;
        sK = lds + L.sK; sD = lds + L.sD; sW = lds + L.sW; sV = lds + L.sV; sGd = lds + L.sGd; misc = lds + L.misc;
        double* w = P.ws + (size_t)b * P.wl.total;
        const WsLayout& W = P.wl;
        rowstate = w + W.rowstate; obs_nh = w + W.obs_nh; obs_c0 = w + W.obs_c0;
        obs_mask = reinterpret_cast<uint64_t*>(w + W.obs_mask);
        PG = w + W.PG; QQ = w + W.QQ; Paft = w + W.Paft; Piaft = w + W.Piaft; Kg = w + W.K; Sinvg = w + W.Sinv;
        Dg = w + W.D; Phicl = w + W.Phicl;
        x_init = P.x_init + (size_t)b * n; goal_lo = P.goal_lo + (size_t)b * n; goal_hi = P.goal_hi + (size_t)b * n;
        dt = P.tf[b] / (N - 1);  // Trajectory(X,U,Tf): dt = Tf/(N-1), types.jl:235
        ng = 0;
#pragma unroll
        for (int i = 0; i < n; i++) { gidx[i] = 0; gval[i] = 0; }
        for (int i = 0; i < n; i++)
            if (goal_lo[i] == goal_hi[i]) { gidx[ng] = i; gval[ng] = goal_lo[i]; ng++; }
    }
    GD const double* PGk(int k) const { return PG + (size_t)(T::LTI ? 0 : k) * n * NZ; }
};

// M_k and Gam_k of knot k (k >= 1) from the stored [Phi | Gam] block
template <int MODEL> GD void load_M_Gam(const Blk<MODEL>& K, int k, double* M, double* Gam) {
    constexpr int n = Blk<MODEL>::n, m = Blk<MODEL>::m, NZ = n + m;
    const double* pg = K.PGk(k);
#pragma unroll
    for (int i = 0; i < n; i++) {
#pragma unroll
        for (int j = 0; j < n; j++) M[i * n + j] = 0.5 * (pg[i * NZ + j] + (i == j ? 1.0 : 0.0));
#pragma unroll
        for (int j = 0; j < m; j++) Gam[i * m + j] = pg[i * NZ + n + j];
    }
}

// initialize_model_params!/update_model_params! (freeflyer_se2.jl:116-147): linearise at (Xp,Up); also evaluates
// the signed distances of the linearisation point and freezes which obstacle rows are active this trip.
template <int MODEL> GD void linearize(Blk<MODEL>& K, double toggle) {
    using T = MT<MODEL>;
    constexpr int n = T::n, m = T::m, NZ = n + m;
    const int k = K.tid;
    if (k < K.N) {
        const double* xp = K.Xp + k * n;
        const double* up = K.Up + k * m;
        if (!T::LTI || k == 0) {
            double A[n * n], G[n * n], M[n * n], B[n * m];
            Dyn<MODEL>::A(K.P.mp, xp, up, A);
            Dyn<MODEL>::B(K.P.mp, B);
            const double h = 0.5 * K.dt;
#pragma unroll
            for (int i = 0; i < n; i++)
#pragma unroll
                for (int j = 0; j < n; j++) G[i * n + j] = (i == j ? 1.0 : 0.0) - h * A[i * n + j];
            inv_gauss_jordan<n>(G, M);
            double* pg = K.PG + (size_t)(T::LTI ? 0 : k) * n * NZ;
#pragma unroll
            for (int i = 0; i < n; i++) {
#pragma unroll
                for (int j = 0; j < n; j++) pg[i * NZ + j] = 2.0 * M[i * n + j] - (i == j ? 1.0 : 0.0);
#pragma unroll
                for (int j = 0; j < m; j++) {
                    double s = 0;
#pragma unroll
                    for (int l = 0; l < n; l++) s += M[i * n + l] * (h * B[l * m + j]);
                    pg[i * NZ + n + j] = 2.0 * s;
                }
            }
        }
        uint64_t mask = 0;
        if constexpr (T::HAS_OBS) {
            for (int i = 0; i < K.P.n_obs; i++) {
                double nh[T::WS];
                const double dist = signed_distance<T::WS>(K.P, 0, xp, i, nh);
                if (dist < toggle) {
                    mask |= (uint64_t)1 << i;
                    double c0 = K.P.mp.clearance - dist;
#pragma unroll
                    for (int j = 0; j < T::WS; j++) {
                        K.obs_nh[((size_t)i * T::WS + j) * K.N + k] = nh[j];
                        c0 += nh[j] * xp[j];
                    }
                    K.obs_c0[(size_t)i * K.N + k] = c0;
                }
            }
        }
        K.obs_mask[k] = mask;
    }
    __syncthreads();
}

// ---- Riccati factorisation of the condensed KKT system (cooperative, sequential in k) ---------------
template <int MODEL> GD void factor_sweep(Blk<MODEL>& K, double* fail) {
    using T = MT<MODEL>;
    constexpr int n = T::n, m = T::m, NZ = n + m, NQ = NZ * (NZ + 1) / 2, NPG = n * NZ;
    constexpr int QPT = (NQ + 63) / 64, PPT = (NPG + 63) / 64;
    const int tid = K.tid, NT = K.NT, N = K.N, ng = K.ng;
    for (int e = tid; e < n * n; e += NT) { K.sP[e] = 0; K.sPi[e] = 0; K.sGd[e] = 0; }
    // stage N-1 operands
    double qq[QPT], pgn[PPT];
#pragma unroll
    for (int r = 0; r < QPT; r++) { const int e = tid + r * NT; qq[r] = (e < NQ) ? K.QQ[(size_t)(N - 1) * NQ + e] : 0.0; }
    {
        const double* pg = K.PGk(N - 1);
        for (int e = tid; e < NPG; e += NT) K.sPG[((N - 1) & 1) * NPG + e] = pg[e];
    }
    __syncthreads();
    for (int k = N - 1; k >= 0; k--) {
        const double* PGs = K.sPG + (k & 1) * NPG;
        // prefetch the operands of knot k-1 while this knot is processed
        double qqn[QPT];
#pragma unroll
        for (int r = 0; r < QPT; r++) {
            const int e = tid + r * NT;
            qqn[r] = (k > 0 && e < NQ) ? K.QQ[(size_t)(k - 1) * NQ + e] : 0.0;
        }
        if (k > 1) {
            const double* pg = K.PGk(k - 1);
#pragma unroll
            for (int r = 0; r < PPT; r++) { const int e = tid + r * NT; pgn[r] = (e < NPG) ? pg[e] : 0.0; }
        }
        // value function after knot k
        for (int e = tid; e < n * n; e += NT) K.Paft[(size_t)k * n * n + e] = K.sP[e];
        for (int e = tid; e < n * ng; e += NT) K.Piaft[(size_t)k * n * n + e] = K.sPi[e];
        // phase 1: T = P [Phi Gam],  Z = [Phi Gam]^T Pi (+ E at the last knot)
        for (int e = tid; e < NPG + NZ * ng; e += NT) {
            if (e < NPG) {
                const int i = e / NZ, j = e % NZ;
                double s = 0;
#pragma unroll
                for (int l = 0; l < n; l++) s += K.sP[i * n + l] * PGs[l * NZ + j];
                K.sT[e] = s;
            } else {
                const int e2 = e - NPG, j = e2 / ng, g = e2 % ng;
                double s = 0;
#pragma unroll
                for (int l = 0; l < n; l++) s += PGs[l * NZ + j] * K.sPi[l * ng + g];
                if (k == N - 1) {  // E = [M^T C^T; b^T M^T C^T]: M = (Phi + I)/2, M b = Gam/2
                    const int gi = K.gidx[g];
                    s += 0.5 * (PGs[gi * NZ + j] + ((j == gi) ? 1.0 : 0.0));
                }
                K.sZ[e2] = s;
            }
        }
        __syncthreads();
        // phase 2: Hh = QQ + [Phi Gam]^T T (one triangle, mirrored)
#pragma unroll
        for (int r = 0; r < QPT; r++) {
            int e = tid + r * NT;
            if (e < NQ) {
                int i = 0, rem = e;
                while (rem >= NZ - i) { rem -= NZ - i; i++; }
                const int j = i + rem;
                double s = qq[r];
#pragma unroll
                for (int l = 0; l < n; l++) s += PGs[l * NZ + i] * K.sT[l * NZ + j];
                K.sHh[i * NZ + j] = s;
                K.sHh[j * NZ + i] = s;
            }
        }
        __syncthreads();
        // phase 3: block Cholesky of [S Hyu^T; Hyu Hyy]: L = chol(S), W = L^-1 Hyu^T, V = L^-1 Zu,
        // K = L^-T W, D = L^-T V.  One thread per column; the m x m factor is recomputed by each of them.
        if (tid < n + ng || tid < m * m) {
            double S[m * m], Li[m * m];
#pragma unroll
            for (int i = 0; i < m; i++)
#pragma unroll
                for (int j = 0; j < m; j++) S[i * m + j] = K.sHh[(n + i) * NZ + n + j];
            if (!chol_inv<m>(S, Li)) *fail = 1.0;
            for (int c = tid; c < n + ng; c += NT) {
                double col[m], w[m], kk[m];
                const bool isK = c < n;
                const int g = c - n;
#pragma unroll
                for (int l = 0; l < m; l++) col[l] = isK ? K.sHh[c * NZ + n + l] : K.sZ[(n + l) * ng + g];
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
#pragma unroll
                for (int i = 0; i < m; i++) {
                    if (isK) { K.sW[i * n + c] = w[i]; K.sK[i * n + c] = kk[i]; K.Kg[(size_t)k * m * n + i * n + c] = kk[i]; }
                    else { K.sV[i * ng + g] = w[i]; K.sD[i * ng + g] = kk[i]; K.Dg[(size_t)k * m * n + i * ng + g] = kk[i]; }
                }
            }
            for (int e = tid; e < m * m; e += NT) {  // S^-1 = L^-T L^-1 (feed-forward only)
                const int i = e / m, j = e % m;
                double s = 0;
#pragma unroll
                for (int l = 0; l < m; l++) if (l >= i && l >= j) s += Li[l * m + i] * Li[l * m + j];
                K.Sinvg[(size_t)k * m * m + e] = s;
            }
        }
        __syncthreads();
        // phase 4: P' = Hyy - W^T W, Pi' = Zy - W^T V, Phicl = Phi - Gam K, Gd += V^T V.  The Schur complements are
        // never formed through an explicit S^-1: with barrier weights ~1/mu in Hyy that loses every digit.
        for (int e = tid; e < 2 * n * n + n * ng + ng * ng; e += NT) {
            if (e < n * n) {
                const int i = e / n, j = e % n;
                double a = K.sHh[i * NZ + j];
#pragma unroll
                for (int l = 0; l < m; l++) a -= K.sW[l * n + i] * K.sW[l * n + j];
                K.sP[e] = a;
            } else if (e < 2 * n * n) {
                const int e2 = e - n * n, i = e2 / n, j = e2 % n;
                double s = PGs[i * NZ + j];
#pragma unroll
                for (int l = 0; l < m; l++) s -= PGs[i * NZ + n + l] * K.sK[l * n + j];
                K.Phicl[(size_t)k * n * n + e2] = s;
            } else if (e < 2 * n * n + n * ng) {
                const int e2 = e - 2 * n * n, i = e2 / ng, g = e2 % ng;
                double s = K.sZ[i * ng + g];
#pragma unroll
                for (int l = 0; l < m; l++) s -= K.sW[l * n + i] * K.sV[l * ng + g];
                K.sPi[e2] = s;
            } else {
                const int e2 = e - 2 * n * n - n * ng, g = e2 / ng, h = e2 % ng;
                double s = 0;
#pragma unroll
                for (int l = 0; l < m; l++) s += K.sV[l * ng + g] * K.sV[l * ng + h];
                K.sGd[e2] += s;
            }
        }
        // operands of the next knot: knot 0 has Phi = 0, Gam = b_0 (x_1 is pinned)
        if (k > 1) {
#pragma unroll
            for (int r = 0; r < PPT; r++) { const int e = tid + r * NT; if (e < NPG) K.sPG[((k - 1) & 1) * NPG + e] = pgn[r]; }
        } else if (k == 1) {
            double B[n * m];
            Dyn<MODEL>::B(K.P.mp, B);
            for (int e = tid; e < NPG; e += NT) {
                const int i = e / NZ, j = e % NZ;
                double v = 0.0;
#pragma unroll
                for (int q = 0; q < n * m; q++) if (j >= n && q == i * m + (j - n)) v = 0.5 * K.dt * B[q];
                K.sPG[0 * NPG + e] = v;
            }
        }
#pragma unroll
        for (int r = 0; r < QPT; r++) qq[r] = qqn[r];
        __syncthreads();
    }
}

// p_{k-1} = Phicl_k^T (p_k + r_k) + qt_k, k = N-1..1; pv[k] holds qt_k on entry and p_k on exit
template <int MODEL> GD void backward_sweep(Blk<MODEL>& K) {
    constexpr int n = Blk<MODEL>::n;
    const int tid = K.tid, N = K.N;
    double p = 0.0, col[n], coln[n];
    if (tid < n) {
#pragma unroll
        for (int l = 0; l < n; l++) col[l] = K.Phicl[(size_t)(N - 1) * n * n + l * n + tid];
    }
    for (int k = N - 1; k >= 1; k--) {
        double* buf = K.sT + (k & 1) * n;
        if (tid < n) {
            if (k > 1) {
#pragma unroll
                for (int l = 0; l < n; l++) coln[l] = K.Phicl[(size_t)(k - 1) * n * n + l * n + tid];
            }
            buf[tid] = p + K.rv[k * n + tid];
        }
        __syncthreads();
        if (tid < n) {
            double s = K.pv[k * n + tid];
            K.pv[k * n + tid] = p;
#pragma unroll
            for (int l = 0; l < n; l++) s += col[l] * buf[l];
            p = s;
#pragma unroll
            for (int l = 0; l < n; l++) col[l] = coln[l];
        }
    }
    if (tid < n) K.pv[tid] = p;
    __syncthreads();
}

// dy_k = Phicl_k dy_{k-1} + ct_k, k = 0..N-1; dY[k] holds ct_k on entry and dy_k on exit
template <int MODEL> GD void forward_sweep(Blk<MODEL>& K) {
    constexpr int n = Blk<MODEL>::n;
    const int tid = K.tid, N = K.N;
    double y = 0.0, row[n], rown[n];
    if (tid < n) {
#pragma unroll
        for (int l = 0; l < n; l++) row[l] = K.Phicl[(size_t)tid * n + l];
    }
    for (int k = 0; k < N; k++) {
        double* buf = K.sT + (k & 1) * n;
        if (tid < n) {
            if (k + 1 < N) {
#pragma unroll
                for (int l = 0; l < n; l++) rown[l] = K.Phicl[(size_t)(k + 1) * n * n + tid * n + l];
            }
            buf[tid] = y;
        }
        __syncthreads();
        if (tid < n) {
            double s = K.dY[k * n + tid];
#pragma unroll
            for (int l = 0; l < n; l++) s += row[l] * buf[l];
            y = s;
            K.dY[k * n + tid] = y;
#pragma unroll
            for (int l = 0; l < n; l++) row[l] = rown[l];
        }
    }
    __syncthreads();
}

// ---- the interior point method ---------------------------------------------------------------------
template <int MODEL> GD void ipm_solve(Blk<MODEL>& K, double Delta, double omega, IpmOut& out) {
    using T = MT<MODEL>;
    constexpr int n = T::n, m = T::m, NZ = n + m, NHX = n * (n + 1) / 2, NHU = m * (m + 1) / 2, NQ = NZ * (NZ + 1) / 2;
    const int k = K.tid, N = K.N, ng = K.ng;
    const bool act = k < N;
    const gusto_ipm_opts& io = K.P.io;
    const double kappa = 1.0 / fmax(1.0, omega);
    const double wk = kappa * ((k == 0 || k == N - 1) ? 0.5 * K.dt : K.dt);
    const double hdt = 0.5 * K.dt;
    double* red = K.misc;       // [0..7] block_reduce scratch
    double* fail = K.misc + 8;  // factorisation failure flag
    double* gxs = K.misc + 16;  // gx of knot 0 (n values)

    RowCtx<MODEL> ctx;
    ctx.P = &K.P; ctx.N = N; ctx.k = k; ctx.nslot = K.P.wl.nslot; ctx.kappa = kappa; ctx.omega = omega; ctx.Delta = Delta;
    ctx.xp = K.Xp + (act ? k : 0) * n; ctx.mask = act ? K.obs_mask[k] : 0; ctx.obs_nh = K.obs_nh; ctx.obs_c0 = K.obs_c0;
    ctx.goal_lo = K.goal_lo; ctx.goal_hi = K.goal_hi;
    RowState rs{K.rowstate, K.P.wl.nslot, N, act ? k : 0};

    // warm start at traj_prev (scp_gusto.jl:100-102) with x_1 pinned to x_init; slacks interior
    double xs[n], us[m], xpk[n], upk[m], fp[n];
#pragma unroll
    for (int i = 0; i < n; i++) { xs[i] = 0; xpk[i] = 0; fp[i] = 0; }
#pragma unroll
    for (int i = 0; i < m; i++) { us[i] = 0; upk[i] = 0; }
    double ncomp_l = 0;
    if (act) {
#pragma unroll
        for (int i = 0; i < n; i++) { xpk[i] = K.Xp[k * n + i]; xs[i] = (k == 0) ? K.x_init[i] : xpk[i]; K.Xw[k * n + i] = xs[i]; K.nu[k * n + i] = 0; }
#pragma unroll
        for (int i = 0; i < m; i++) { upk[i] = K.Up[k * m + i]; us[i] = upk[i]; K.Uw[k * m + i] = us[i]; }
        Dyn<MODEL>::f(K.P.mp, xpk, upk, fp);
        OpInit op{rs};
        visit_rows<MODEL>(ctx, xs, us, op);
        ncomp_l = op.ncomp;
    }
    if (k == 0) *fail = 0.0;
    const double ncomp = block_reduce(ncomp_l, OpSum(), red);
    double mug[n], mugn[n];
#pragma unroll
    for (int i = 0; i < n; i++) { mug[i] = 0; mugn[i] = 0; }

    int status = GUSTO_SOLVER_FAILED, it = 0;
    double res_p = 0, res_d = 0, mu = 0;
    for (it = 0;; it++) {
        // (1) linearised xdot at each knot: a_k = f_k + A_k (x_k - xp_k) + B (u_k - up_k)
        double Ad[n * n], Bd[n * m];
        if (act) {
            Dyn<MODEL>::A(K.P.mp, xpk, upk, Ad);
            Dyn<MODEL>::B(K.P.mp, Bd);
#pragma unroll
            for (int i = 0; i < n; i++) {
                double s = fp[i];
#pragma unroll
                for (int j = 0; j < n; j++) s += Ad[i * n + j] * (xs[j] - xpk[j]);
#pragma unroll
                for (int j = 0; j < m; j++) s += Bd[i * m + j] * (us[j] - upk[j]);
                K.pv[k * n + i] = s;
            }
        }
        __syncthreads();
        // (2) residuals, condensed Hessian blocks, dual residual
        double Hx[NHX], Hu[NHU], rdx[n], rdu[m], rdk[n];
#pragma unroll
        for (int i = 0; i < NHX; i++) Hx[i] = 0;
#pragma unroll
        for (int i = 0; i < NHU; i++) Hu[i] = 0;
#pragma unroll
        for (int i = 0; i < n; i++) { rdx[i] = 0; rdk[i] = 0; }
#pragma unroll
        for (int i = 0; i < m; i++) rdu[i] = 0;
        double l_resp = 0, l_resd = 0, l_comp = 0, l_numax = 0;
        double rg[n];
#pragma unroll
        for (int i = 0; i < n; i++) rg[i] = 0;
        if (act) {
            if (k >= 1) {
#pragma unroll
                for (int i = 0; i < n; i++) {
                    rdk[i] = K.Xw[(k - 1) * n + i] - xs[i] + hdt * (K.pv[(k - 1) * n + i] + K.pv[k * n + i]);
                    l_resp = nanmax(l_resp, fabs(rdk[i]));
                }
            }
#pragma unroll
            for (int i = 0; i < n; i++) K.rd[k * n + i] = rdk[i];
            OpResidHess<n, m> op{rs, Hx, Hu, rdx, rdu};
            visit_rows<MODEL>(ctx, xs, us, op);
            l_comp = op.comp;
            l_resp = nanmax(l_resp, op.maxrp);
#pragma unroll
            for (int i = 0; i < m; i++) { Hu[sidx(i, i, m)] += 2 * wk; rdu[i] += 2 * wk * us[i]; }
            // + E^T nu: F_k^T nu_{k+1} - G_k^T nu_k on x, b_k^T (nu_{k+1} + nu_k) on u
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
                for (int j = 0; j < n; j++) s += hdt * Ad[j * n + i] * vs[j];
                rdx[i] += s;
            }
#pragma unroll
            for (int i = 0; i < m; i++) {
                double s = 0;
#pragma unroll
                for (int j = 0; j < n; j++) s += hdt * Bd[j * m + i] * vs[j];
                rdu[i] += s;
            }
            if (k == N - 1) {
                for (int j = 0; j < ng; j++) {
                    const int gi = K.gidx[j];
#pragma unroll
                    for (int i = 0; i < n; i++) if (i == gi) { rdx[i] += mug[j]; rg[j] = K.gval[j] - xs[i]; }
                    l_resp = nanmax(l_resp, fabs(rg[j]));
                }
            }
            if (k >= 1) {
#pragma unroll
                for (int i = 0; i < n; i++) l_resd = nanmax(l_resd, fabs(rdx[i]));
            }
#pragma unroll
            for (int i = 0; i < m; i++) l_resd = nanmax(l_resd, fabs(rdu[i]));
        }
        res_p = block_reduce(l_resp, OpNanMax(), red);
        res_d = block_reduce(l_resd, OpNanMax(), red);
        const double comp = block_reduce(l_comp, OpSum(), red);
        const double numax = block_reduce(l_numax, OpMax(), red);
        mu = ncomp > 0 ? comp / ncomp : 0.0;
        if (res_p <= io.tol && res_d <= io.tol * (1 + numax) && mu <= 0.1 * io.tol) { status = GUSTO_SOLVER_OPTIMAL; break; }
        if (it >= io.max_iter) {
            if (res_p <= io.tol_acc && res_d <= io.tol_acc * (1 + numax) && mu <= io.tol_acc) status = GUSTO_SOLVER_ALMOST;
            break;
        }
        if (!isfinite(res_p) || !isfinite(res_d) || !isfinite(mu)) break;

        // (3) stage cost of the LQR in (dy_{k-1}, du_k): QQ = [Qt, Qt b; ., Hu + b^T Qt b], Qt = M^T Hx M
        double Mk[n * n], Gamk[n * m];
        if (act) {
            double* qqg = K.QQ + (size_t)k * NQ;
            if (k >= 1) {
                load_M_Gam<MODEL>(K, k, Mk, Gamk);
                double tmp[n * n], Qt[NHX], Qb[n * m];
#pragma unroll
                for (int i = 0; i < n; i++)
#pragma unroll
                    for (int j = 0; j < n; j++) {
                        double s = 0;
#pragma unroll
                        for (int l = 0; l < n; l++) s += Hx[sidx(i, l, n)] * Mk[l * n + j];
                        tmp[i * n + j] = s;
                    }
#pragma unroll
                for (int i = 0; i < n; i++)
#pragma unroll
                    for (int j = i; j < n; j++) {
                        double s = 0;
#pragma unroll
                        for (int l = 0; l < n; l++) s += Mk[l * n + i] * tmp[l * n + j];
                        Qt[sidx(i, j, n)] = s;
                        qqg[sidx(i, j, NZ)] = s;
                    }
#pragma unroll
                for (int i = 0; i < n; i++)
#pragma unroll
                    for (int j = 0; j < m; j++) {
                        double s = 0;
#pragma unroll
                        for (int l = 0; l < n; l++) s += Qt[sidx(i, l, n)] * (hdt * Bd[l * m + j]);
                        Qb[i * m + j] = s;
                        qqg[sidx(i, n + j, NZ)] = s;
                    }
#pragma unroll
                for (int i = 0; i < m; i++)
#pragma unroll
                    for (int j = i; j < m; j++) {
                        double s = Hu[sidx(i, j, m)];
#pragma unroll
                        for (int l = 0; l < n; l++) s += (hdt * Bd[l * m + i]) * Qb[l * m + j];
                        qqg[sidx(n + i, n + j, NZ)] = s;
                    }
#pragma unroll
                for (int i = 0; i < n; i++) {
                    double s = 0, c = -rdk[i];
#pragma unroll
                    for (int l = 0; l < n; l++) { s += Qt[sidx(i, l, n)] * rdk[l]; c += 2.0 * Mk[i * n + l] * rdk[l]; }
                    K.qrd[k * n + i] = s;
                    K.cv[k * n + i] = c;
                }
            } else {
#pragma unroll
                for (int e = 0; e < NQ; e++) qqg[e] = 0.0;
#pragma unroll
                for (int i = 0; i < m; i++)
#pragma unroll
                    for (int j = i; j < m; j++) qqg[sidx(n + i, n + j, NZ)] = Hu[sidx(i, j, m)];
#pragma unroll
                for (int i = 0; i < n; i++) { K.qrd[i] = 0; K.cv[i] = 0; }
#pragma unroll
                for (int i = 0; i < n * m; i++) Gamk[i] = hdt * Bd[i];
            }
        }
        __syncthreads();
        // (4) factorise
        factor_sweep<MODEL>(K, fail);
        if (k == 0 && ng > 0) {
            if (!inv_spd_rt(K.sGd, K.sP, K.sGd + n * n, ng)) *fail = 1.0;  // sP <- Gd^-1
        }
        if (act) {  // r_k = P_k c_k
#pragma unroll
            for (int i = 0; i < n; i++) {
                double s = 0;
#pragma unroll
                for (int l = 0; l < n; l++) s += K.Paft[(size_t)k * n * n + i * n + l] * K.cv[k * n + l];
                K.rv[k * n + i] = s;
            }
        }
        __syncthreads();
        if (*fail != 0.0) break;

        // (5) predictor (mu_t = 0) and centred corrector share the factorisation
        double sigma = 0, mu_t = 0, alpha = 1.0;
        double dxs[n], dus[m];
        for (int pass = 0; pass < 2; pass++) {
            double gx[n], gu[m], quk[m], Kk[m * n], Dk[m * n];
#pragma unroll
            for (int i = 0; i < n; i++) gx[i] = 0;
#pragma unroll
            for (int i = 0; i < m; i++) { gu[i] = 0; quk[i] = 0; }
            if (act) {
#pragma unroll
                for (int i = 0; i < m; i++) gu[i] = 2 * wk * us[i];
                OpRhs op{rs, gx, gu, pass, mu_t};
                visit_rows<MODEL>(ctx, xs, us, op);
#pragma unroll
                for (int e = 0; e < m * n; e++) { Kk[e] = K.Kg[(size_t)k * m * n + e]; Dk[e] = (e < m * ng) ? K.Dg[(size_t)k * m * n + e] : 0.0; }
                double gy[n];
#pragma unroll
                for (int i = 0; i < n; i++) gy[i] = 0;
                if (k >= 1) {
#pragma unroll
                    for (int i = 0; i < n; i++) {
                        double s = K.qrd[k * n + i];
#pragma unroll
                        for (int l = 0; l < n; l++) s += Mk[l * n + i] * gx[l];
                        gy[i] = s;
                    }
                } else {
#pragma unroll
                    for (int i = 0; i < n; i++) gxs[i] = gx[i];
                }
#pragma unroll
                for (int i = 0; i < m; i++) {
                    double s = gu[i];
#pragma unroll
                    for (int l = 0; l < n; l++) s += (hdt * Bd[l * m + i]) * gy[l];
                    quk[i] = s;
                    K.qu[k * m + i] = s;
                }
#pragma unroll
                for (int i = 0; i < n; i++) {
                    double s = gy[i];
#pragma unroll
                    for (int l = 0; l < m; l++) s -= Kk[l * n + i] * quk[l];
                    K.pv[k * n + i] = s;
                }
            }
            __syncthreads();
            backward_sweep<MODEL>(K);
            // feed-forward, goal multiplier
            double d0[m], th[n], lu[m];
#pragma unroll
            for (int i = 0; i < n; i++) th[i] = 0;
#pragma unroll
            for (int i = 0; i < m; i++) { d0[i] = 0; lu[i] = 0; }
            if (act) {
                double tt[n];
#pragma unroll
                for (int i = 0; i < n; i++) tt[i] = K.pv[k * n + i] + K.rv[k * n + i];
#pragma unroll
                for (int i = 0; i < m; i++) {
                    double s = quk[i];
#pragma unroll
                    for (int l = 0; l < n; l++) s += Gamk[l * m + i] * tt[l];
                    lu[i] = s;
                }
#pragma unroll
                for (int i = 0; i < m; i++) {
                    double s = 0;
#pragma unroll
                    for (int l = 0; l < m; l++) s += K.Sinvg[(size_t)k * m * m + i * m + l] * lu[l];
                    d0[i] = s;
                }
                for (int j = 0; j < ng; j++) {
                    double s = 0;
                    for (int i = 0; i < n; i++) s += K.Piaft[(size_t)k * n * n + i * ng + j] * K.cv[k * n + i];
#pragma unroll
                    for (int i = 0; i < m; i++) s -= K.Dg[(size_t)k * m * n + i * ng + j] * lu[i];
                    if (k == N - 1) {  // + C M rd_{N-1} - rg
                        const int gi = K.gidx[j];
                        for (int i = 0; i < n; i++) s += Mk[gi * n + i] * rdk[i];
                        s -= rg[j];
                    }
#pragma unroll
                    for (int i = 0; i < n; i++) if (i == j) th[i] = s;
                }
            }
            for (int j = 0; j < ng; j++) {
                double v = 0;
#pragma unroll
                for (int i = 0; i < n; i++) if (i == j) v = th[i];
                v = block_reduce(v, OpSum(), red);
#pragma unroll
                for (int i = 0; i < n; i++) if (i == j) th[i] = v;
            }
            for (int j = 0; j < ng; j++) {
                double s = 0;
                for (int l = 0; l < ng; l++) {
                    double tl = 0;
#pragma unroll
                    for (int i = 0; i < n; i++) if (i == l) tl = th[i];
                    s += K.sP[j * ng + l] * tl;
                }
#pragma unroll
                for (int i = 0; i < n; i++) if (i == j) mugn[i] = s;
            }
            double dk[m];
#pragma unroll
            for (int i = 0; i < m; i++) dk[i] = 0;
            if (act) {
#pragma unroll
                for (int i = 0; i < m; i++) {
                    double s = d0[i];
                    for (int j = 0; j < ng; j++) {
                        double mj = 0;
#pragma unroll
                        for (int q = 0; q < n; q++) if (q == j) mj = mugn[q];
                        s += Dk[i * ng + j] * mj;
                    }
                    dk[i] = s;
                }
#pragma unroll
                for (int i = 0; i < n; i++) {
                    double s = K.cv[k * n + i];
#pragma unroll
                    for (int l = 0; l < m; l++) s -= Gamk[i * m + l] * dk[l];
                    K.dY[k * n + i] = s;
                }
            }
            __syncthreads();
            forward_sweep<MODEL>(K);
            // primal step of this knot and the new costates
#pragma unroll
            for (int i = 0; i < n; i++) dxs[i] = 0;
#pragma unroll
            for (int i = 0; i < m; i++) dus[i] = 0;
            if (act) {
                double dyp[n];
#pragma unroll
                for (int i = 0; i < n; i++) dyp[i] = (k >= 1) ? K.dY[(k - 1) * n + i] : 0.0;
#pragma unroll
                for (int i = 0; i < m; i++) {
                    double s = -dk[i];
#pragma unroll
                    for (int l = 0; l < n; l++) s -= Kk[i * n + l] * dyp[l];
                    dus[i] = s;
                }
                if (k >= 1) {
                    double a[n];
#pragma unroll
                    for (int i = 0; i < n; i++) {
                        double s = dyp[i] + rdk[i];
#pragma unroll
                        for (int l = 0; l < m; l++) s += (hdt * Bd[i * m + l]) * dus[l];
                        a[i] = s;
                    }
#pragma unroll
                    for (int i = 0; i < n; i++) {
                        double s = 0;
#pragma unroll
                        for (int l = 0; l < n; l++) s += Mk[i * n + l] * a[l];
                        dxs[i] = s;
                    }
                }
                if (k + 1 < N) {  // nu_{k+1} = P_k dy_k + p_k + Pi_k mu_g
#pragma unroll
                    for (int i = 0; i < n; i++) {
                        double s = K.pv[k * n + i];
#pragma unroll
                        for (int l = 0; l < n; l++) s += K.Paft[(size_t)k * n * n + i * n + l] * K.dY[k * n + l];
                        for (int j = 0; j < ng; j++) {
                            double mj = 0;
#pragma unroll
                            for (int q = 0; q < n; q++) if (q == j) mj = mugn[q];
                            s += K.Piaft[(size_t)k * n * n + i * ng + j] * mj;
                        }
                        K.nun[(k + 1) * n + i] = s;
                    }
                }
            }
            __syncthreads();
            if (k == 0) {  // x_1 stationarity: gx_0 + nu_0 + F_0^T nu_1 = 0
#pragma unroll
                for (int i = 0; i < n; i++) {
                    double s = gxs[i] + K.nun[n + i];
#pragma unroll
                    for (int j = 0; j < n; j++) s += hdt * Ad[j * n + i] * K.nun[n + j];
                    K.nun[i] = -s;
                }
            }
            // row steps + fraction to the boundary
            const double tau = pass ? fmax(0.995, 1.0 - mu) : 1.0;
            double l_amax = 1.0;
            if (act) {
                OpStep op{rs, dxs, dus, pass, mu_t, tau};
                visit_rows<MODEL>(ctx, xs, us, op);
                l_amax = op.amax;
            }
            const double a_max = block_reduce(l_amax, OpMin(), red);
            if (pass == 0) {
                double l_ca = 0;
                if (act) {
                    OpAff op{rs, a_max};
                    visit_rows<MODEL>(ctx, xs, us, op);
                    l_ca = op.comp;
                }
                const double ca = block_reduce(l_ca, OpSum(), red);
                const double mu_aff = ncomp > 0 ? ca / ncomp : 0.0;
                const double rr = (mu > 0) ? mu_aff / mu : 0.0;
                sigma = rr * rr * rr;
                mu_t = fmax(sigma * mu, io.mu_floor);
                alpha = a_max;
                if (ncomp == 0) break;  // equality-constrained QP: the predictor already is the Newton step
            } else {
                alpha = a_max;
            }
        }
        // (6) update
        if (act) {
#pragma unroll
            for (int i = 0; i < n; i++) {
                xs[i] += alpha * dxs[i];
                K.Xw[k * n + i] = xs[i];
                K.nu[k * n + i] += alpha * (K.nun[k * n + i] - K.nu[k * n + i]);
            }
#pragma unroll
            for (int i = 0; i < m; i++) { us[i] += alpha * dus[i]; K.Uw[k * m + i] = us[i]; }
            OpUpdate op{rs, alpha};
            visit_rows<MODEL>(ctx, xs, us, op);
        }
#pragma unroll
        for (int i = 0; i < n; i++) mug[i] += alpha * (mugn[i] - mug[i]);
        __syncthreads();
    }
    // JuMP.objective_value: cost + all slacks, in unscaled units
    double l_obj = 0;
    if (act) {
#pragma unroll
        for (int i = 0; i < m; i++) l_obj += wk * us[i] * us[i];
        OpSlackSum op{rs};
        visit_rows<MODEL>(ctx, xs, us, op);
        l_obj += op.sum;
    }
    const double obj = block_reduce(l_obj, OpSum(), red);
    __syncthreads();
    out.status = status; out.iters = it; out.obj = obj / kappa; out.res_p = res_p; out.res_d = res_d; out.mu = mu;
}

}  // namespace gusto
