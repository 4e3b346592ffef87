// factor1w.hpp -- the one-wave factor sweeps (the Riccati recursion of the condensed KKT system, ipm.hpp) of a problem with
// N <= 64 knots: factor_sweep_1w (dubins_car), factor_sweep_pg2 (the double integrator, freeflyerSE2: software-pipelined over the
// value-function and goal-sensitivity chains) and factor_sweep_mfma (the 12/13-state models: every product of a stage on
// v_mfma_f64_16x16x4_f64), with the choice of the costate form that goes with them (P | Pi records or the adjoint recursion).
// Included by ipm.hpp, which holds the workgroup views (Blk, SweepView) these sweeps run on.
// Reference path: the convex subproblem handed to JuMP.optimize!, scp_gusto.jl:104,178-314.
#pragma once

namespace gusto {

// ---- one-wave variants: every lane's role and LDS addresses are fixed before the knot loop ---------------

// LDS buffer holding [Phi Gam] of knot k: knot 0 has its own ([0 | b_0], x_1 is pinned), LTI models one more,
// time-varying models double-buffer.
template <int MODEL, class BLK> GD double* pg_buf(const BLK& K, int k) {
    using T = MT<MODEL>;
    using R = Rec<MODEL>;
    constexpr int NPG = T::n * (T::n + T::m);
    if constexpr (LdsC<MODEL, true>::PG_LDS) return K.pgl + k * NPG;   // (every knot's block lives in LDS: read in place)
    else if constexpr (T::LTI) return K.sPG + (k == 0 ? 2 * NPG : 0);
    else return K.sPG + (k & 1) * NPG;
}

// Two LDS phases per knot (the dependency chain is P_k -> H -> P_{k-1}):
//   AB: H = QQ + [Phi Gam]^T (P [Phi Gam]) entry-per-lane (each lane forms the column of P [Phi Gam] it needs itself
//       instead of waiting for a shared T), Z = [Phi Gam]^T Pi (+E), r_k = P_k c_k, Pi_k^T c_k
//   CD: every lane factors the m x m block S itself, forms the columns of W = L^-1 Hyu^T, V = L^-1 Zu it needs and
//       writes P' = Hyy - W^T W, Pi' = Zy - W^T V, Phicl = Phi - Gam K, Gd += V^T V (Schur complements never
//       through an explicit S^-1: with barrier weights ~1/mu in Hyy that loses every digit).
template <int MODEL> GD void factor_sweep_1w(SweepView<MODEL> K, double* fail, Prof& pf) {
    using T = MT<MODEL>;
    using R = Rec<MODEL>;
    using C = LdsC<MODEL, true>;
    constexpr int n = T::n, m = T::m, NZ = n + m, NQ = NZ * (NZ + 1) / 2, NPG = n * NZ, NN = n * n, NZN = NZ * n;
    constexpr int RT = (NPG + 63) / 64, RQ = (NQ + 63) / 64, RN = (NN + 63) / 64, RZ = (NZN + 63) / 64, RKD = R::SKD / 64;
    const int tid = K.tid, N = K.N;
    int zJ[RZ], zG[RZ], hI[RQ], hJ[RQ], nI[RN], nJ[RN], tL[RT], tC[RT];
#pragma unroll
    for (int r = 0; r < RT; r++) { const int e = tid + 64 * r; tL[r] = (e < NPG) ? e / NZ : 0; tC[r] = (e < NPG) ? e % NZ : 0; }
#pragma unroll
    for (int r = 0; r < RZ; r++) { const int e = tid + 64 * r; zJ[r] = (e < NZN) ? e / n : 0; zG[r] = (e < NZN) ? e % n : 0; }
#pragma unroll
    for (int r = 0; r < RQ; r++) { const int e = tid + 64 * r, ij = (e < NQ) ? K.lut[e] : 0; hI[r] = ij >> 8; hJ[r] = ij & 255; }
#pragma unroll
    for (int r = 0; r < RN; r++) { const int e = tid + 64 * r; nI[r] = (e < NN) ? e / n : 0; nJ[r] = (e < NN) ? e % n : 0; }
    if constexpr (T::LTI) {   // knot-0 operands [0 | b_0] (third buffer) and the one [Phi Gam] block of an LTI model
        double B[n * m];
        Dyn<MODEL>::B(*K.mpp, B);
#pragma unroll
        for (int r = 0; r < RT; r++) {
            const int e = tid + 64 * r;
            if (e < NPG) {
                const int i = e / NZ, j = e % NZ;
                double v = 0.0;
#pragma unroll
                for (int q = 0; q < n * m; q++) if (j >= n && q == i * m + (j - n)) v = 0.5 * K.dt * B[q];
                K.sPG[2 * NPG + e] = v;
                K.sPG[e] = K.PGk(N - 1)[e];
            }
        }
    } else if constexpr (!C::PG_LDS) {   // time-varying: block N-1 now, the others one knot ahead; block 0 of the global array IS [0 | b_0]
#pragma unroll
        for (int r = 0; r < RT; r++) {
            const int e = tid + 64 * r;
            if (e < NPG) K.sPG[((N - 1) & 1) * NPG + e] = K.PGk(N - 1)[e];
        }
    }
    static_assert(!T::MFMA, "the matrix-core models run factor_sweep_mfma");
    double qq[RQ], pgn[RT];
#pragma unroll
    for (int r = 0; r < RQ; r++) {
        const int e = tid + 64 * r;
        if constexpr (C::KD_LDS) qq[r] = K.kdl[(N - 1) * C::KDS + (e < NQ ? e : 0)];
        else qq[r] = K.QQ[(size_t)(N - 1) * R::SQQ + e];
    }
#pragma unroll
    for (int r = 0; r < RT; r++) pgn[r] = 0.0;
#pragma unroll
    for (int r = 0; r < RN; r++) {
        const int e = tid + 64 * r;
        if (e < NN) {
            K.sP[e] = 0; K.sPi[e] = 0; K.sGd[e] = 0;
            K.Paft[(size_t)(N - 1) * R::SNN + e] = 0.0;   // value function after the last knot
            K.Piaft[(size_t)(N - 1) * R::SNN + e] = 0.0;
        }
        if constexpr (C::KD_LDS) { if (e < R::SNN) K.pprec(N - 1)[e] = 0.0; }   // (packed P | Pi record)
    }
    K.sync();
    for (int k = N - 1; k >= 0; k--) {
        const double* PGs = pg_buf<MODEL>(K, k);
        double qqn[RQ];
#pragma unroll
        for (int r = 0; r < RQ; r++) {   // unconditional (clamped) prefetch: see phase CD
            const int e = tid + 64 * r;
            if constexpr (C::KD_LDS) qqn[r] = K.kdl[((k > 0) ? k - 1 : 0) * C::KDS + (e < NQ ? e : 0)];   // (slot k-1 still holds QQ_{k-1})
            else qqn[r] = K.QQ[(size_t)((k > 0) ? k - 1 : 0) * R::SQQ + e];   // (padded record: every lane has an entry)
        }
        if (!T::LTI && !C::PG_LDS) {
            const auto pg = K.PGk((k > 0) ? k - 1 : 0);
#pragma unroll
            for (int r = 0; r < RT; r++) { const int e = tid + 64 * r; pgn[r] = pg[(e < NPG) ? e : NPG - 1]; }
        }
        pf.tick(PF_FPRE);
        double hreg[RQ];   // this lane's entries of H, kept for the S = H_uu broadcast of phase CD
        // ---- phase AB -------------------------------------------------------------------------------
        // Small models: every LDS operand of the three products (H, Z, r) is requested before the first FMA, so the
        // stage pays ONE LDS latency here instead of three back-to-back read -> wait -> compute chains.
        // Models whose [Phi Gam] has <= 2 nonzeros per column (MT::PG2): contractions over the two structural rows of
        // a column only.
        if constexpr (T::PG2 && RZ == 1 && RQ == 1) {
            // H[i][j] = QQ + sum_{a,b} PG[ra(i)][i] P[ra(i)][rb(j)] PG[rb(j)][j]: four entries of P per lane, one phase
            const int zc = zJ[0], zg = zG[0], hc = hI[0], hj = hJ[0];
            // the second structural row of a column is always h3 = n/2 rows below the first (pg_r1 = pg_r0 + h3), so each
            // operand pair / quadruple is ONE lane-dependent LDS address plus immediates; r_k = P c reads column ri of the
            // (exactly symmetric) P, the same access pattern as Pi^T c: one address for both kinds of lane
            constexpr int h3 = n / 2;
            static_assert(T::pg_r1(0) == T::pg_r0(0) + h3 && T::pg_r1(n) == T::pg_r0(n) + h3 && T::pg_r1(n - 1) == T::pg_r0(n - 1) + h3, "PG2 row pairs");
            const int z0 = T::pg_r0(zc), i0 = T::pg_r0(hc), j0 = T::pg_r0(hj);
            const bool isr = tid < n;
            const int ri = isr ? tid : ((tid < 2 * n) ? tid - n : 0);
            double ra[n], rb[n];
            const double* pa = PGs + i0 * NZ + hc;
            const double* pb = PGs + j0 * NZ + hj;
            const double a0 = pa[0], a1 = pa[h3 * NZ], b0 = pb[0], b1 = pb[h3 * NZ];
            const auto pp = K.sP + (i0 * n + j0);
            const double p00 = pp[0], p01 = pp[h3], p10 = pp[h3 * n], p11 = pp[h3 * n + h3];
            const auto pz = K.sPi + (z0 * n + zg);
            const double* pv2 = PGs + z0 * NZ + zc;
            const double zb0 = pz[0], zb1 = pz[h3 * n], zv0 = pv2[0], zv1 = pv2[h3 * NZ];
            const auto pr = (isr ? K.sP : K.sPi) + ri;
#pragma unroll
            for (int l = 0; l < n; l++) { ra[l] = pr[l * n]; rb[l] = K.cv[k * n + l]; }
            __builtin_amdgcn_sched_barrier(0);
            {
                const double h = qq[0] + a0 * (b0 * p00 + b1 * p01) + a1 * (b0 * p10 + b1 * p11);
                hreg[0] = h;
                if (tid < NQ) { K.sHh[hc * NZ + hj] = h; K.sHh[hj * NZ + hc] = h; }
            }
            {
                double z = zv0 * zb0 + zv1 * zb1;
                // E = [M^T C^T; b^T M^T C^T], M = (Phi + I)/2, M b = Gam/2; column g only for goal coordinates
                if (k == N - 1 && K.is_goal(zg)) z += 0.5 * (PGs[zg * NZ + zc] + ((zc == zg) ? 1.0 : 0.0));
                if (tid < NZN) K.sZ[tid] = z;
            }
            {   // r_k = P_k c_k and Pi_k^T c_k for the stage-parallel blocks
                double rr = 0;
#pragma unroll
                for (int l = 0; l < n; l++) rr += ra[l] * rb[l];
                if (tid < 2 * n) (isr ? K.rv : K.nun)[k * n + ri] = rr;
            }
        } else
        // Two steps, T = P [Phi Gam] then H = QQ + [Phi Gam]^T T: each lane contracts ONE index per step (2 x n FMAs
        // and 4 n operands instead of n^2 + n FMAs and n^2 + 2 n operands), at the price of one more trip through LDS.
        if constexpr (2 * (RT + RZ + 1) * n <= 96) {
            double tp[RT][n], tg[RT][n], za[RZ][n], zb[RZ][n], ra[n], rb[n];
            const bool isr = tid < n;
            const int ri = isr ? tid : ((tid < 2 * n) ? tid - n : 0);
#pragma unroll
            for (int r = 0; r < RT; r++)
#pragma unroll
                for (int q = 0; q < n; q++) { tp[r][q] = K.sP[tL[r] * n + q]; tg[r][q] = PGs[q * NZ + tC[r]]; }
#pragma unroll
            for (int r = 0; r < RZ; r++)
#pragma unroll
                for (int l = 0; l < n; l++) { za[r][l] = PGs[l * NZ + zJ[r]]; zb[r][l] = K.sPi[l * n + zG[r]]; }
#pragma unroll
            for (int l = 0; l < n; l++) { ra[l] = *(isr ? K.sP + ri * n + l : K.sPi + l * n + ri); rb[l] = K.cv[k * n + l]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < RT; r++) {
                double t = 0;
#pragma unroll
                for (int q = 0; q < n; q++) t += tp[r][q] * tg[r][q];
                if (tid + 64 * r < NPG) K.sT[tid + 64 * r] = t;
            }
#pragma unroll
            for (int r = 0; r < RZ; r++) {
                double z = 0;
#pragma unroll
                for (int l = 0; l < n; l++) z += za[r][l] * zb[r][l];
                // E = [M^T C^T; b^T M^T C^T], M = (Phi + I)/2, M b = Gam/2; column g only for goal coordinates
                if (k == N - 1 && K.is_goal(zG[r])) z += 0.5 * (PGs[zG[r] * NZ + zJ[r]] + ((zJ[r] == zG[r]) ? 1.0 : 0.0));
                if (tid + 64 * r < NZN) K.sZ[tid + 64 * r] = z;
            }
            {   // r_k = P_k c_k and Pi_k^T c_k for the stage-parallel blocks
                double rr = 0;
#pragma unroll
                for (int l = 0; l < n; l++) rr += ra[l] * rb[l];
                if (tid < 2 * n) (isr ? K.rv : K.nun)[k * n + ri] = rr;
            }
            K.sync();
            double pgi[RQ][n], tj[RQ][n];
#pragma unroll
            for (int r = 0; r < RQ; r++)
#pragma unroll
                for (int l = 0; l < n; l++) { pgi[r][l] = PGs[l * NZ + hI[r]]; tj[r][l] = K.sT[l * NZ + hJ[r]]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < RQ; r++) {
                double h = qq[r];
#pragma unroll
                for (int l = 0; l < n; l++) h += pgi[r][l] * tj[r][l];
                hreg[r] = h;
                if (tid + 64 * r < NQ) { K.sHh[hI[r] * NZ + hJ[r]] = h; K.sHh[hJ[r] * NZ + hI[r]] = h; }
            }
        } else
        if constexpr (n * n + 2 * (RQ + RZ + 1) * n <= 112) {
            double pm[n * n], pgj[RQ][n], pgi[RQ][n], za[RZ][n], zb[RZ][n], ra[n], rb[n];
            const bool isr = tid < n;
            const int ri = isr ? tid : ((tid < 2 * n) ? tid - n : 0);
#pragma unroll
            for (int e = 0; e < n * n; e++) pm[e] = K.sP[e];
#pragma unroll
            for (int r = 0; r < RQ; r++)
#pragma unroll
                for (int l = 0; l < n; l++) { pgj[r][l] = PGs[l * NZ + hJ[r]]; pgi[r][l] = PGs[l * NZ + hI[r]]; }
#pragma unroll
            for (int r = 0; r < RZ; r++)
#pragma unroll
                for (int l = 0; l < n; l++) { za[r][l] = PGs[l * NZ + zJ[r]]; zb[r][l] = K.sPi[l * n + zG[r]]; }
#pragma unroll
            for (int l = 0; l < n; l++) { ra[l] = *(isr ? K.sP + ri * n + l : K.sPi + l * n + ri); rb[l] = K.cv[k * n + l]; }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int r = 0; r < RQ; r++) {
                double s = qq[r];
#pragma unroll
                for (int l = 0; l < n; l++) {
                    double t = 0;   // (P [Phi Gam])[l][j]
#pragma unroll
                    for (int q = 0; q < n; q++) t += pm[l * n + q] * pgj[r][q];
                    s += pgi[r][l] * t;
                }
                hreg[r] = s;
                if (tid + 64 * r < NQ) { K.sHh[hI[r] * NZ + hJ[r]] = s; K.sHh[hJ[r] * NZ + hI[r]] = s; }
            }
#pragma unroll
            for (int r = 0; r < RZ; r++) {
                double s = 0;
#pragma unroll
                for (int l = 0; l < n; l++) s += za[r][l] * zb[r][l];
                // E = [M^T C^T; b^T M^T C^T], M = (Phi + I)/2, M b = Gam/2; column g only for goal coordinates
                if (k == N - 1 && K.is_goal(zG[r])) s += 0.5 * (PGs[zG[r] * NZ + zJ[r]] + ((zJ[r] == zG[r]) ? 1.0 : 0.0));
                if (tid + 64 * r < NZN) K.sZ[tid + 64 * r] = s;
            }
            {   // r_k = P_k c_k and Pi_k^T c_k for the stage-parallel blocks
                double s = 0;
#pragma unroll
                for (int l = 0; l < n; l++) s += ra[l] * rb[l];
                if (tid < 2 * n) (isr ? K.rv : K.nun)[k * n + ri] = s;
            }
        } else {
            // large models: the same two steps, one round of 64 entries at a time (n^2 operands per lane do not fit
            // the register file: the one-step form spilled 5-6 KB per lane and cost n^2 + n FMAs per entry)
#pragma unroll
            for (int r = 0; r < RT; r++) {
                double tp[n], tg[n];
#pragma unroll
                for (int q = 0; q < n; q++) { tp[q] = K.sP[tL[r] * n + q]; tg[q] = PGs[q * NZ + tC[r]]; }
                __builtin_amdgcn_sched_barrier(0);
                double t = 0;
#pragma unroll
                for (int q = 0; q < n; q++) t += tp[q] * tg[q];
                if (tid + 64 * r < NPG) K.sT[tid + 64 * r] = t;
            }
            K.sync();
#pragma unroll
            for (int r = 0; r < RQ; r++) {
                double pgi[n], tj[n];
#pragma unroll
                for (int l = 0; l < n; l++) { pgi[l] = PGs[l * NZ + hI[r]]; tj[l] = K.sT[l * NZ + hJ[r]]; }
                __builtin_amdgcn_sched_barrier(0);
                double h = qq[r];
#pragma unroll
                for (int l = 0; l < n; l++) h += pgi[l] * tj[l];
                hreg[r] = h;
                if (tid + 64 * r < NQ) { K.sHh[hI[r] * NZ + hJ[r]] = h; K.sHh[hJ[r] * NZ + hI[r]] = h; }
            }
            K.sync();   // T is dead from here: Z (below) is written into the same LDS words (LdsC: sT == sZ)

#pragma unroll
            for (int r = 0; r < RZ; r++) {
                if (tid + 64 * r < NZN) {
                    double a[n], bb[n];
#pragma unroll
                    for (int l = 0; l < n; l++) { a[l] = PGs[l * NZ + zJ[r]]; bb[l] = K.sPi[l * n + zG[r]]; }
                    __builtin_amdgcn_sched_barrier(0);
                    double s = 0;
#pragma unroll
                    for (int l = 0; l < n; l++) s += a[l] * bb[l];
                    // E = [M^T C^T; b^T M^T C^T], M = (Phi + I)/2, M b = Gam/2; column g only for goal coordinates
                    if (k == N - 1 && K.is_goal(zG[r])) s += 0.5 * (PGs[zG[r] * NZ + zJ[r]] + ((zJ[r] == zG[r]) ? 1.0 : 0.0));
                    K.sZ[tid + 64 * r] = s;
                }
            }
        
            for (int e = tid; e < 2 * n; e += 64) {   // r_k = P_k c_k and Pi_k^T c_k for the stage-parallel blocks
                const bool isr = e < n;
                const int i = isr ? e : e - n;
                double a[n], bb[n];
#pragma unroll
                for (int l = 0; l < n; l++) { a[l] = isr ? K.sP[i * n + l] : K.sPi[l * n + i]; bb[l] = K.cv[k * n + l]; }
                __builtin_amdgcn_sched_barrier(0);
                double s = 0;
#pragma unroll
                for (int l = 0; l < n; l++) s += a[l] * bb[l];
                (isr ? K.rv : K.nun)[k * n + i] = s;
            }
        
        }
        K.sync();
        pf.tick(PF_FAB);
        // ---- phase CD -------------------------------------------------------------------------------
        // Every global store below is unconditional (idle lanes and k == 0 aim at the dummy pad): with no VMEM op
        // under a branch the compiler can count the stores issued after the QQ prefetch and wait with vmcnt(#stores)
        // instead of draining them all with vmcnt(0) at the end of every stage.
        {
            double S[m * m], Li[m * m];
            // S = H_uu straight from the registers of the lanes that formed it (v_readlane): the Cholesky starts without
            // waiting for the trip of H through LDS, which only the solve operands below still take
#pragma unroll
            for (int i = 0; i < m; i++)
#pragma unroll
                for (int j = 0; j < m; j++) {
                    const int e = sidx(n + (i < j ? i : j), n + (i < j ? j : i), NZ);
                    S[i * m + j] = readlane_f64(hreg[e / 64], e % 64);
                }
            // the operands of the solves do not depend on the Cholesky factor: request them first, they land while
            // the (latency-bound, wave-uniform) factorisation runs
            double hi[RN][m], hj[RN][m], zi[RN][m], zj[RN][m], gi[RN][m], pn_[RN], ph_[RN], pin_[RN], gd_[RN];
#pragma unroll
            for (int r = 0; r < RN; r++) {
                const int i = nI[r], j = nJ[r];   // (0, 0) on idle lanes: every LDS address stays valid
#pragma unroll
                for (int l = 0; l < m; l++) {
                    hi[r][l] = K.sHh[i * NZ + n + l]; hj[r][l] = K.sHh[j * NZ + n + l];
                    zi[r][l] = K.sZ[(n + l) * n + i]; zj[r][l] = K.sZ[(n + l) * n + j];
                    gi[r][l] = PGs[i * NZ + n + l];
                }
                pn_[r] = K.sHh[i * NZ + j]; ph_[r] = PGs[i * NZ + j]; pin_[r] = K.sZ[i * n + j];
                gd_[r] = K.sGd[(tid + 64 * r < NN) ? tid + 64 * r : 0];
            }
            if constexpr (RN * m <= 6) __builtin_amdgcn_sched_barrier(0);   // (larger models: leave the order to the compiler)
            if (!chol_inv<m>(S, Li)) *fail = 1.0;
            pf.tick(PF_F4);
            // take the prefetched QQ_{k-1} here, BEFORE this stage's stores are issued: the wait then covers the
            // prefetch and the previous stage's stores (a whole phase old), not a store issued a moment ago
#pragma unroll
            for (int r = 0; r < RQ; r++) qq[r] = qqn[r];
            __builtin_amdgcn_sched_barrier(0);
            double kdv[RN];
#pragma unroll
            for (int r = 0; r < RN; r++) {
                const int e2 = tid + 64 * r;
                const bool on = e2 < NN;
                const int i = nI[r], j = nJ[r];
                double pn = pn_[r], ph = ph_[r], pin = pin_[r], gd = gd_[r];
                double wi[m], wj[m], vi[m], vj[m], kj[m], dj[m];
#pragma unroll
                for (int a = 0; a < m; a++) {   // W = L^-1 Hyu^T, V = L^-1 Zu (columns i and j)
                    double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
                    for (int l = 0; l <= a; l++) {
                        s1 += Li[a * m + l] * hi[r][l]; s2 += Li[a * m + l] * hj[r][l];
                        s3 += Li[a * m + l] * zi[r][l]; s4 += Li[a * m + l] * zj[r][l];
                    }
                    wi[a] = s1; wj[a] = s2; vi[a] = s3; vj[a] = s4;
                }
#pragma unroll
                for (int a = 0; a < m; a++) {   // K = L^-T W, D = L^-T V (column j)
                    double s1 = 0, s2 = 0;
#pragma unroll
                    for (int l = a; l < m; l++) { s1 += Li[l * m + a] * wj[l]; s2 += Li[l * m + a] * vj[l]; }
                    kj[a] = s1; dj[a] = s2;
                }
#pragma unroll
                for (int l = 0; l < m; l++) { pn -= wi[l] * wj[l]; ph -= gi[r][l] * kj[l]; pin -= wi[l] * vj[l]; gd += vi[l] * vj[l]; }
                pf.tick(PF_F6);
                if (on) { K.sP[e2] = pn; K.sPi[e2] = pin; K.sGd[e2] = gd; }
                // padded stage records (Rec<MODEL>): every lane stores, idle lanes land in the padding; record -1 of
                // Paft/Piaft exists for k == 0
                if constexpr (C::PHICL_LDS) { if (on) K.Phicl[k * K.SPH + e2] = ph; }
                else if constexpr (!C::KD_LDS) K.Phicl[(size_t)k * R::SNN + e2] = ph;   // (KD_LDS: rebuilt from K by the sweeps)
                // (idle lanes all aim at ONE padding slot, the entry after the matrix: the stores stay unconditional, but the
                // record dirties 10 sectors of 32 B in the L2 instead of 16)
                static_assert(R::SNN > NN, "padding slot");
                if constexpr (C::KD_LDS) {
                    // one record per knot for both: P is symmetric, its upper triangle (n (n + 1) / 2 entries) and Pi (n^2)
                    // fit the 64 doubles; the step phase walks one record instead of two
                    constexpr int NH = n * (n + 1) / 2;
                    static_assert(NH + NN < R::SNN, "P | Pi record");
                    const int ep = (on && i <= j) ? sidx(i, j, n) : K.PP_DUMMY;
                    const int eq = on ? NH + e2 : K.PP_DUMMY;
                    K.pprec(k - 1)[ep] = pn;
                    K.pprec(k - 1)[eq] = pin;
                } else {
                    const int eo = on ? e2 : NN;
                    K.Paft[(size_t)(k - 1) * R::SNN + eo] = pn;
                    K.Piaft[(size_t)(k - 1) * R::SNN + eo] = pin;
                }
                // lanes of rows 0..m-1 hold K[i][j], rows m..2m-1 hold D[i-m][j]  (n >= 2m for every model): entry
                // i*n + j of the K|D|S^-1 record
                static_assert(n >= 2 * m, "K/D store mapping");
                // register r of a lane holds entry e = tid + 64 r, i.e. row i = e / n in [64 r / n, (64 r + 63) / n]: rows
                // outside that range are pruned at compile time (and with them the back-substitutions that feed them)
                if constexpr (C::KD_LDS) {
                    // LDS copy of K | D: every lane of column j holds all of K[:, j] and D[:, j]; the lanes of row 0 (entry
                    // e2 = j) write them -- six ds_write under one lane mask instead of a select chain per record entry
                    static_assert(RN == 1, "one register of (i, j) entries");
                    if (e2 < n) {
#pragma unroll
                        for (int a = 0; a < m; a++) {
                            K.kdl[k * C::KDS + a * n + e2] = kj[a];
                            K.kdl[k * C::KDS + m * n + a * n + e2] = dj[a];
                        }
                    }
                    kdv[r] = 0.0;
                } else {
                    const int ilo = (64 * r) / n, ihi = ((64 * r + 63 < NN - 1) ? 64 * r + 63 : NN - 1) / n;
                    double kd = 0.0;
#pragma unroll
                    for (int a = 0; a < m; a++) if (a >= ilo && a <= ihi) kd = (i == a) ? kj[a] : kd;
#pragma unroll
                    for (int a = 0; a < m; a++) if (m + a >= ilo && m + a <= ihi) kd = (i == m + a) ? dj[a] : kd;
                    kdv[r] = kd;
                }
            }
            if constexpr (C::KD_LDS) {   // S^-1 = L^-T L^-1, upper triangle: wave-uniform values, lane 0 writes them
                double sp[m * (m + 1) / 2];
#pragma unroll
                for (int a = 0; a < m; a++)
#pragma unroll
                    for (int c = 0; c <= a; c++) {
                        double s = 0;
#pragma unroll
                        for (int l = a; l < m; l++) s += Li[l * m + a] * Li[l * m + c];
                        sp[sidx(c, a, m)] = s;
                    }
                if (tid == 0) {
#pragma unroll
                    for (int e = 0; e < m * (m + 1) / 2; e++) K.kdl[k * C::KDS + 2 * m * n + e] = sp[e];
                }
            } else {   // S^-1 = L^-T L^-1 (feed-forward only): computed wave-uniformly, the lane of record entry oS + e keeps entry e
                double sv[RKD];
#pragma unroll
                for (int r = 0; r < RKD; r++) sv[r] = 0;
#pragma unroll
                for (int a = 0; a < m; a++)
#pragma unroll
                    for (int c = 0; c <= a; c++) {
                        double s = 0;
#pragma unroll
                        for (int l = a; l < m; l++) s += Li[l * m + a] * Li[l * m + c];
#pragma unroll
                        for (int r = 0; r < RKD; r++) {   // (registers whose 64 entries miss [oS, oS + m^2): nothing to do)
                            if (64 * r + 63 < R::oS || 64 * r >= R::oS + m * m) continue;
                            const int e = tid + 64 * r - R::oS;
                            sv[r] = (e == a * m + c || e == c * m + a) ? s : sv[r];
                        }
                    }
                // the K|D|S^-1 record: entries [0, 2mn) come from the (i, j) lanes above, [2mn, 2mn + m^2) are S^-1
#pragma unroll
                for (int r = 0; r < RKD; r++) {
                    const int e = tid + 64 * r;
                    double v = (r < RN) ? kdv[r < RN ? r : 0] : 0.0;
                    if (64 * r + 63 >= R::oS && 64 * r < R::oS + m * m) { if (e >= R::oS) v = sv[r]; }
                    K.KD[(size_t)k * R::SKD + e] = v;
                }
            }
        }
        if (!T::LTI && !C::PG_LDS && k > 0) {
#pragma unroll
            for (int r = 0; r < RT; r++) { const int e = tid + 64 * r; if (e < NPG) K.sPG[((k - 1) & 1) * NPG + e] = pgn[r]; }
        }
        pf.tick(PF_F7);
        K.sync();
        pf.tick(PF_FCD);
    }
}

// The factor sweep of the double-integrator model (MT::PG2, one wave, K | D | S^-1 in LDS), software pipelined.
// A stage has two chains: the value function (P_k -> H -> chol(H_uu) -> W -> P_{k-1}: the critical path, ~45 dependent
// flops of a wave-uniform 3 x 3 Cholesky in its middle) and the goal sensitivities (Pi_k -> Z -> V = L^-1 Z_u ->
// Pi_{k-1}, Gd, D), which needs L_k and W_k of the first chain but nothing the first chain waits for.  Written stage by
// stage (factor_sweep_1w) the second chain sits behind the Cholesky of its own stage and the wave -- one per SIMD, in-order
// issue -- idles through the factorisation.  Here iteration k runs stage k of the first chain together with the second
// half of stage k+1 of the second (tail: V, Pi, Gd, D from the L and W kept in registers) and the first half of its stage k
// (head: Z = [Phi Gam]^T Pi_k, Pi_k^T c_k): independent instruction streams in one basic block.  To keep it ONE basic block
// nothing is predicated: every LDS / global store is unconditional, lanes without an entry aim at a dummy slot (16 doubles
// of the T buffer, which this path does not use), the last knot's E term is a select, the not-positive-definite flag is
// accumulated and stored once after the sweep.  [Phi Gam] is constant over the sweep except at knot 0 ([0 | b_0], x_1 is
// pinned): its entries live in registers and knot 0 is a peeled copy of the stage.  Same arithmetic as factor_sweep_1w,
// operation for operation: the results are bit-identical.
// LDS traffic is what bounds a stage (measured: a wave issues one ds_read every ~8 cycles whatever its width up to 128 bits,
// a ds_read2 costs two, a ds_write ~13, against 4 cycles for an fp64 FMA), so the operands are laid out for 128-bit reads:
// P is kept with its columns interleaved (j, j + n/2 adjacent: the two entries a lane of H needs from a row are one
// ds_read_b128) and read by ROWS for r_k = P c (P is exactly symmetric), Pi is kept transposed (the column for Pi^T c is a
// row), only the upper triangle of H is written, and what a lane reads back from itself (its entry of Gd, its entry of Z)
// stays in a register.
template <int MODEL> GD void factor_sweep_pg2(SweepView<MODEL> K, double* fail, Prof& pf) {
    using T = MT<MODEL>;
    using R = Rec<MODEL>;
    using C = LdsC<MODEL, true>;
    constexpr int n = T::n, m = T::m, NZ = n + m, NQ = NZ * (NZ + 1) / 2, NPG = n * NZ, NN = n * n, NZN = NZ * n, h3 = n / 2,
                  NH = n * (n + 1) / 2;
    static_assert(T::PG2 && T::LTI && C::KD_LDS && NQ <= 64 && NZN <= 64 && NN <= 64 && n >= 2 * m, "shape");
    static_assert(T::pg_r1(0) == T::pg_r0(0) + h3 && T::pg_r1(n) == T::pg_r0(n) + h3 && T::pg_r1(n - 1) == T::pg_r0(n - 1) + h3, "PG2 row pairs");
    static_assert(NH + NN < R::SNN && !C::BIG, "P | Pi record, dummy slot");
    const int tid = K.tid, N = K.N;
    // ---- lane roles ----
    const int ijh = K.lut[tid < NQ ? tid : 0], hc = ijh >> 8, hj = ijh & 255, i0 = T::pg_r0(hc), j0 = T::pg_r0(hj);   // H[hc][hj]
    const int zc = tid < NZN ? tid / n : 0, zg = tid < NZN ? tid % n : 0, z0 = T::pg_r0(zc);                       // Z[zc][zg]
    const int ri = tid < n ? tid : 0;                                                                             // r[ri], Pi^T c [ri]
    const bool on = tid < NN;
    const int i = on ? tid / n : 0, j = on ? tid % n : 0;                                                         // P[i][j], Pi[i][j]
    // ---- LDS operands (offsets in doubles from the base of the dynamic LDS) ----
    const LPtr<double> L = K.lds;
    typedef double v2d __attribute__((ext_vector_type(2)));
    auto ld2 = [&](int off) { return *(const __attribute__((address_space(3))) v2d*)(L.p + off); };   // ds_read_b128 (off even)
    const int dmy = C::sT0 + (tid & 15);             // dummy slot of this lane (+ immediates < 36 stay inside the T buffer)
    static_assert(n * NZ >= 16 + 36, "dummy slot");
    auto pcol = [](int j_) { return 2 * (j_ % h3) + j_ / h3; };       // column j of P sits at position pcol(j) of its row
    static_assert(C::sP % 2 == 0 && C::sPi % 2 == 0 && n % 2 == 0 && C::vecs % 2 == 0, "16-byte aligned rows");
    const int oPP = C::sP + i0 * n + 2 * j0, oPZ = C::sPi + zg * n + z0, oPr = C::sP + ri * n, oPir = C::sPi + ri * n;
    const int wH1 = tid < NQ ? C::sHh + hc * NZ + hj : dmy;           // (upper triangle only: hc <= hj)
    const int wZ = tid < NZN ? C::sZ + tid : dmy;
    const int oHi = C::sHh + i * NZ + n, oHj = C::sHh + j * NZ + n, oPn = C::sHh + (i < j ? i : j) * NZ + (i < j ? j : i);
    const int oZi = C::sZ + n * n + i, oZj = C::sZ + n * n + j;
    const int wP = on ? C::sP + i * n + pcol(j) : dmy, wPi = on ? C::sPi + j * n + i : dmy, wGd = on ? C::sGd + tid : dmy;
    const int vecs = C::vecs, oCv = vecs + 3 * N * n, oRv = vecs + 4 * N * n, oNun = vecs + 6 * N * n;   // (Blk::rebind_lds)
    const int wRv = tid < n ? oRv + tid : dmy, wNun = tid < n ? oNun + tid : dmy, sRv = tid < n ? n : 0;
    const int kdo = K.kd_off;
    const int wKD = tid < n ? kdo + tid : dmy, sKD = tid < n ? C::KDS : 0;             // K[a][tid] at + a n, D[a][tid] at + (m + a) n
    const int wSi = tid == 0 ? kdo + 2 * m * n : dmy, sSi = tid == 0 ? C::KDS : 0;     // S^-1, upper triangle
    const int ep = (on && i <= j) ? sidx(i, j, n) : R::SNN - 1, eq = on ? NH + tid : R::SNN - 1;   // packed P | Pi record
    // ---- [Phi Gam] entries of this lane: the block of the sweep and the knot-0 block [0 | b_0] ----
    struct PGC { double a0, a1, b0, b1, zv0, zv1; };
    double Bd[n * m];
    Dyn<MODEL>::B(*K.mpp, Bd);
    auto pg_main = [&](int r_, int c_) { return K.PGk(N - 1)[r_ * NZ + c_]; };
    auto pg_zero = [&](int r_, int c_) {   // ([0 | dt/2 B], formed as factor_sweep_1w forms it)
        double v = 0.0;
#pragma unroll
        for (int q = 0; q < n * m; q++) if (c_ >= n && q == r_ * m + (c_ - n)) v = 0.5 * K.dt * Bd[q];
        return v;
    };
    PGC cN, c0;
    cN.a0 = pg_main(i0, hc); cN.a1 = pg_main(i0 + h3, hc); cN.b0 = pg_main(j0, hj); cN.b1 = pg_main(j0 + h3, hj);
    cN.zv0 = pg_main(z0, zc); cN.zv1 = pg_main(z0 + h3, zc);
    c0.a0 = pg_zero(i0, hc); c0.a1 = pg_zero(i0 + h3, hc); c0.b0 = pg_zero(j0, hj); c0.b1 = pg_zero(j0 + h3, hj);
    c0.zv0 = pg_zero(z0, zc); c0.zv1 = pg_zero(z0 + h3, zc);
    // E = [M^T C^T; b^T M^T C^T] of the last knot, M = (Phi + I)/2, M b = Gam/2; column g only for goal coordinates
    const double eterm = pg_main(zg, zc) + ((zc == zg) ? 1.0 : 0.0);
    const bool egoal = K.is_goal(zg);
    // ---- start: P = Pi = Gd = 0 after the last knot, Z = 0 (the tail of "stage N" then leaves Pi_{N-1} = 0) ----
    L[wP] = 0.0; L[wPi] = 0.0; L[wZ] = 0.0;
    double gdR = 0.0, zR = 0.0;   // this lane's entry of Gd (accumulated over the sweep) and of Z (the Pi' term of the next tail)
    if (tid < R::SNN) K.Paft[(size_t)(N - 1) * R::SNN + tid] = 0.0;
    double qq = K.kdl[(N - 1) * C::KDS + (tid < NQ ? tid : 0)];
    double LiP[m * m], wiP[m];   // L^-1 and this lane's column i of W of the stage before (tail operands)
#pragma unroll
    for (int e = 0; e < m * m; e++) LiP[e] = 0.0;
#pragma unroll
    for (int e = 0; e < m; e++) wiP[e] = 0.0;
    bool okall = true;
    K.sync();

    // tail of stage kt (V, Pi_{kt-1}, Gd, D_kt) from (LiP, wiP) and Z of that stage in LDS; returns nothing, writes LDS + record
    auto tail = [&](int kt, const double* zi, const double* zj) {
        double pin = zR, gd = gdR;
        double vi[m], vj[m], dj[m];
#pragma unroll
        for (int a = 0; a < m; a++) {
            double s3 = 0, s4 = 0;
#pragma unroll
            for (int l = 0; l <= a; l++) { s3 += LiP[a * m + l] * zi[l]; s4 += LiP[a * m + l] * zj[l]; }
            vi[a] = s3; vj[a] = s4;
        }
#pragma unroll
        for (int a = 0; a < m; a++) {
            double s2 = 0;
#pragma unroll
            for (int l = a; l < m; l++) s2 += LiP[l * m + a] * vj[l];
            dj[a] = s2;
        }
#pragma unroll
        for (int l = 0; l < m; l++) { pin -= wiP[l] * vj[l]; gd += vi[l] * vj[l]; }
        L[wPi] = pin; gdR = gd;
        K.Paft[(size_t)(kt - 1) * R::SNN + eq] = pin;
#pragma unroll
        for (int a = 0; a < m; a++) L[wKD + kt * sKD + (m + a) * n] = dj[a];
    };

    // ordering point for LDS traffic between lanes that leaves the ALU work free to move (one wave: the hardware keeps
    // its LDS operations in order, only the compiler has to)
    auto msync = [&]() {
#ifdef GUSTO_STRICT_SYNC
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");   // (check build, see blk_sync)
#endif
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    };
    // GUSTO_STAGE_PROF (with GUSTO_PROFILE): time stamps INSIDE a stage, read asynchronously -- s_memtime is issued where the
    // wave is, its result is only waited for at the end of the stage, so the stamps do not drain the LDS queue
#if defined(GUSTO_PROFILE) && defined(GUSTO_STAGE_PROF)
#define STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define STAMPS_END() do { pf.acc[PF_FPRE] += (long long)(t1_ - t0_); pf.acc[PF_FAB] += (long long)(t2_ - t1_); \
                          pf.acc[PF_F4] += (long long)(t3_ - t2_); pf.acc[PF_F5] += (long long)(t4_ - t3_); pf.acc[PF_FCD] += (long long)(t5_ - t4_); } while (0)
#else
#define STAMP(v) do {} while (0)
#define STAMPS_END() do {} while (0)
#endif
    auto stage = [&](int k, const PGC& c, bool last) {
        STAMP(t0_);
        // ---- operands of this iteration, one batch ----
        const v2d pA = ld2(oPP), pB = ld2(oPP + h3 * n);
        const double p00 = pA.x, p01 = pA.y, p10 = pB.x, p11 = pB.y;
        double ra[n], rb[n];
#pragma unroll
        for (int l = 0; l < n; l += 2) {
            const v2d a2 = ld2(oPr + l), b2 = ld2(oCv + k * n + l);
            // row ri of P holds the columns in the order 0, n/2, 1, n/2 + 1, ...: ra[] back in natural order
            ra[l / 2] = a2.x; ra[l / 2 + h3] = a2.y; rb[l] = b2.x; rb[l + 1] = b2.y;
        }
        double zi[m], zj[m];
#pragma unroll
        for (int l = 0; l < m; l++) { zi[l] = L[oZi + l * n]; zj[l] = L[oZj + l * n]; }
        const double qqn = K.kdl[((k > 0) ? k - 1 : 0) * C::KDS + (tid < NQ ? tid : 0)];   // (slot k-1 still holds QQ_{k-1})
        // ---- value function chain, first half: H, r_k = P_k c_k ----
        const double h = qq + c.a0 * (c.b0 * p00 + c.b1 * p01) + c.a1 * (c.b0 * p10 + c.b1 * p11);
        L[wH1] = h;
        {
            double rr = 0;
#pragma unroll
            for (int l = 0; l < n; l++) rr += ra[l] * rb[l];
            L[wRv + k * sRv] = rr;
        }
        STAMP(t1_);
        msync();
        double S[m * m], Li[m * m];
#pragma unroll
        for (int a = 0; a < m; a++)
#pragma unroll
            for (int b = 0; b < m; b++) {
                const int e = sidx(n + (a < b ? a : b), n + (a < b ? b : a), NZ);
                S[a * m + b] = readlane_f64(h, e);
            }
        double hi[m], hjv[m];
#pragma unroll
        for (int l = 0; l < m; l++) { hi[l] = L[oHi + l]; hjv[l] = L[oHj + l]; }
        double pn = L[oPn];
        // ---- goal chain: tail of the stage before, head of this one (independent of the factorisation below) ----
        tail(k + 1 < N ? k + 1 : N - 1, zi, zj);
        STAMP(t2_);
        msync();
        // (operands of the head first, then the factorisation: its ~45 dependent flops run while they are in flight)
        const double zb0 = L[oPZ], zb1 = L[oPZ + h3];
        double pa[n];
#pragma unroll
        for (int l = 0; l < n; l += 2) { const v2d a2 = ld2(oPir + l); pa[l] = a2.x; pa[l + 1] = a2.y; }
        // ---- value function chain, second half: L = chol(H_uu), W = L^-1 H_uy, K = L^-T W, P_{k-1} = H_yy - W^T W ----
        okall = chol_inv<m>(S, Li) && okall;
        STAMP(t3_);
        {
            double z = c.zv0 * zb0 + c.zv1 * zb1;
            const double zE = fma(0.5, eterm, z);
            z = (last && egoal) ? zE : z;
            L[wZ] = z;
            zR = z;
            double rr = 0;
#pragma unroll
            for (int l = 0; l < n; l++) rr += pa[l] * rb[l];
            L[wNun + k * sRv] = rr;
        }
        STAMP(t4_);
        double wi[m], wj[m], kj[m];
#pragma unroll
        for (int a = 0; a < m; a++) {
            double s1 = 0, s2 = 0;
#pragma unroll
            for (int l = 0; l <= a; l++) { s1 += Li[a * m + l] * hi[l]; s2 += Li[a * m + l] * hjv[l]; }
            wi[a] = s1; wj[a] = s2;
        }
#pragma unroll
        for (int a = 0; a < m; a++) {
            double s1 = 0;
#pragma unroll
            for (int l = a; l < m; l++) s1 += Li[l * m + a] * wj[l];
            kj[a] = s1;
        }
#pragma unroll
        for (int l = 0; l < m; l++) pn -= wi[l] * wj[l];
        L[wP] = pn;
        K.Paft[(size_t)(k - 1) * R::SNN + ep] = pn;     // (record -1 exists for k == 0)
#pragma unroll
        for (int a = 0; a < m; a++) L[wKD + k * sKD + a * n] = kj[a];
        {   // S^-1 = L^-T L^-1, upper triangle (wave-uniform values)
#pragma unroll
            for (int a = 0; a < m; a++)
#pragma unroll
                for (int b = 0; b <= a; b++) {
                    double s1 = 0;
#pragma unroll
                    for (int l = a; l < m; l++) s1 += Li[l * m + a] * Li[l * m + b];
                    L[wSi + k * sSi + sidx(b, a, m)] = s1;
                }
        }
        qq = qqn;
#pragma unroll
        for (int e = 0; e < m * m; e++) LiP[e] = Li[e];
#pragma unroll
        for (int e = 0; e < m; e++) wiP[e] = wi[e];
        msync();
        STAMP(t5_);
        STAMPS_END();
    };
#undef STAMP
#undef STAMPS_END

    for (int k = N - 1; k >= 1; k--) stage(k, cN, k == N - 1);
    stage(0, c0, false);
    {   // the goal chain is one half stage behind: tail of stage 0 (Gd, D_0; its Pi lands in record -1)
        double zi[m], zj[m];
#pragma unroll
        for (int l = 0; l < m; l++) { zi[l] = L[oZi + l * n]; zj[l] = L[oZj + l * n]; }
        tail(0, zi, zj);
    }
    L[wGd] = gdR;
    if (!okall) *fail = 1.0;
    K.sync();
    (void)pf;
}

// The corrector's new costates of the one-wave 12/13-state kernels by the ADJOINT recursion nu_k = Phi_k^T nu_{k+1} + M_k^T
// (H_x dx_k + gx_k) instead of nu_{k+1} = P_k dy_k + p_k + Pi_k mu_g: no P | Pi records at all (2 n^2 doubles per knot written
// by the factor sweep and read back by the costate pass, 0.46 of the 2.0 MB a KKT solve moved), see adjoint_sweep_1w.
#ifndef GUSTO_COSTATE_ADJOINT
#define GUSTO_COSTATE_ADJOINT 1
#endif
template <int MODEL> constexpr bool costate_adjoint() { return GUSTO_COSTATE_ADJOINT && MT<MODEL>::SWEEP_CALL && MT<MODEL>::MFMA; }
// Run-time part of the choice.  The recursion multiplies by Phi_k^T = the Cayley transform of dt/2 A_k, and for MRP kinematics
// (astrobeeSE3: A_pp = d(B(p) w)/dp is not skew) that transform has a pole at dt/2 |A_pp| = 1: on coarse horizons the stage
// errors of the Riccati solution are amplified into a noise floor of the dual residual above the 1e-8 stopping test (measured,
// tf = 70: identical interior point iterations to the P | Pi costates for N >= 45, +8 % at N = 40, +60 % and ALMOST statuses at
// N = 28).  There the kernel keeps the P | Pi records (their stores aim at the records only then).  Quaternion kinematics are
// skew (orthogonal Cayley transform): the manifold model showed no such effect down to N = 5.
// ... and per interior point iteration: these costates are not backward stable the way the P | Pi ones are (whatever the Riccati
// solution is off by lands in them instead of in a small stage residual), and once in a few thousand solves that noise keeps
// the dual residual above the 1e-8 test -- the solve then idles at the acceptable level until the iteration cap (measured:
// astrobeeSE3, 2 of 4044 solves of a B = 1024 batch, 60 iterations each, and the batch waits for them: 31.6 -> 48.5 ms).  A solve
// that has sat at the acceptable level for GUSTO_ADJ_ACC iterations without passing the test, or is still running after
// GUSTO_ADJ_MAX_IT (99 % have stopped by then), therefore goes on with the P | Pi records -- and passes the test an iteration later.
// ipm_solve decides and leaves the answer in LDS (misc[9]) for the phases that are real calls.
#ifndef GUSTO_ADJ_MAX_IT
#define GUSTO_ADJ_MAX_IT 20
#endif
#ifndef GUSTO_ADJ_ACC
#define GUSTO_ADJ_ACC 3
#endif
constexpr int ADJ_FLAG = 9;   // slot of the per-iteration choice in the misc block of the workgroup's LDS
template <int MODEL> GD bool costate_adjoint_now() { return gusto_dyn_lds[LdsC<MODEL, true>::misc + ADJ_FLAG] != 0.0; }
template <int MODEL> GD bool costate_adjoint_rt(const gusto_model_params& mp, double dt) {
    if constexpr (MODEL == GUSTO_ASTROBEE_SE3) return 0.5 * dt * mp.hard_limit_omega <= 0.65;
    else return true;
}
// The factor sweep of the 12/13-state models entirely on the matrix cores (MT::MFMA).  Every matrix of a stage is a
// 16 x 16 tile in the accumulator layout of v_mfma_f64_16x16x4_f64 -- entry (row, col) in register row >> 2 of lane
// (row & 3) << 4 | col -- and that layout IS an operand layout: register s of a tile X, used as the A operand of K step
// s, is X^T; used as the B operand it is X.  So   mfma(X.reg[s], Y.reg[s]) summed over s  =  X^T Y   and the whole stage
//   T = P [Phi Gam],  H = QQ + [Phi Gam]^T T,  Z = [Phi Gam]^T Pi (+E),  W = L^-1 H_uy,  V = L^-1 Z_u,
//   P' = H_yy - W^T W,  Pi' = Z_y - W^T V,  Gd += V^T V,  K = L^-T W,  D = L^-T V,  S^-1 = L^-T L^-1,  Phicl = Phi - Gam K
// chains accumulators into operands without a trip through LDS: P, Pi and Gd stay in registers over the 50 stages.
// The only operands gathered from LDS are Phi, Gam (two layouts) and L^-1 (factored wave-uniformly from the H_uu tile
// by v_readlane, as before).  Column 15 of the Phi tile carries c_k, so r_k = P_k c_k is column 15 of P Phi and
// Pi_k^T c_k is row 15 of Phi^T Pi: the two matrix-vector products of the stage come with the tiles.  Rows / columns
// beyond n (m) of a tile are finite don't-cares that never meet a nonzero operand; stores aim them at the padding slot
// of their record.
// (NOPP: no P | Pi records -- the costates come from the adjoint recursion; a template parameter, chosen at run time by the
// caller, so that the stage loop holds no branch around its stores)
// SEG (round 6, seg.hpp): the sweep over the stages kHi .. kLo of ONE chain of a split horizon.  A chain in front of an interface
// (isA) starts from P = 0, Pi = I -- its end state adjoined as a terminal equality; the last chain (kHi = N - 1) starts as the sweep
// always did.  Every chain leaves the P, Pi in front of its first stage and its Gd in the segmented solve's LDS block (offsets oP,
// oPi, oGd from the base of the dynamic LDS; oGd < 0: sGd), and a chain with kLo > 0 does not write record kLo - 1, which belongs to
// the chain in front of it.
template <int MODEL, bool NOPP, bool SEG = false>
GD void factor_sweep_mfma(SweepView<MODEL> K, double* fail, Prof& pf, int kHi_ = 0, int kLo_ = 0, bool isA = false, int oP = 0, int oPi = 0,
                          int oGd = -1) {
    using T = MT<MODEL>;
    using R = Rec<MODEL>;
    constexpr int n = T::n, m = T::m, NZ = n + m, NPG = n * NZ, NN = n * n;
    using SP = SpPG<MODEL>;
    constexpr bool SPR = SP::USE;   // [Phi Gam] from the compact record of its nonzeros (60 / 73 doubles per knot instead of 216 / 247)
    constexpr int KS = (n + 3) / 4, MS = (m + 3) / 4, RT = SPR ? (SP::NS + 63) / 64 : (NPG + 63) / 64, RN = (NN + 63) / 64;
    static_assert(!T::LTI && n <= 15 && m <= 8 && R::SNN > NN && R::SKD > 2 * m * n + m * m, "tile / record shape");
    // -DGUSTO_PROFILE_COARSE: no stamp inside the stage, the sweep is one interval (PF_FACTOR).  The stamps of the fine profile
    // wait for the values they follow and cost ~150 cycles each: with them the sweep reads as 55 % of a KKT solve, without 40 %.
#ifdef GUSTO_PROFILE_COARSE
#define FT_(id) do { } while (0)
#else
#define FT_(id) pf.tick(id)
#endif
    const int tid = K.tid, N = K.N;
    const int kHi = SEG ? kHi_ : N - 1, kLo = SEG ? kLo_ : 0;
    const int mi = tid & 15, mq = tid >> 4;
    constexpr bool adj_rt = NOPP;
#if defined(GUSTO_PROFILE) && !defined(GUSTO_PROFILE_COARSE)   // finer stamps of a stage (slots 40..47): the value is made a VGPR operand first, so the stamp waits for it
#define FX_(id, val) do { asm volatile("" :: "v"(val)); FT_(40 + (id)); } while (0)
#else
#define FX_(id, val) do { } while (0)
#endif
    const bool c15 = mi == 15;
    // per-lane offsets, fixed for the sweep
    int oF[KS], oG[KS], oGA[MS], oN[KS], oNT[KS], oKr[MS], oDr[MS], oSr[MS], qyy[KS], quy[MS], quu[MS];
    bool vF[KS], vG[KS], vGA[MS];
#pragma unroll
    for (int q = 0; q < KS; q++) {
        const int row = mq + 4 * q;
        vF[q] = row < n && (mi < n || c15); vG[q] = row < n && mi < m;
        oF[q] = (row < n && mi < n) ? row * NZ + mi : 0;
        oG[q] = vG[q] ? row * NZ + n + mi : 0;
        oN[q] = (row < n && mi < n) ? row * n + mi : R::SNN - 1;
        // (P and Pi records: read by costate_pass_1w only, whose lane i takes row i -- stored TRANSPOSED, so that the n lanes of
        // a group read n consecutive doubles per load)
        oNT[q] = (row < n && mi < n) ? mi * n + row : R::SNN - 1;
        qyy[q] = (row < n && mi < n) ? sidx(row, mi, NZ) : -1;
    }
#pragma unroll
    for (int s = 0; s < MS; s++) {
        const int row = mq + 4 * s;
        vGA[s] = mi < n && row < m;
        oGA[s] = vGA[s] ? mi * NZ + n + row : 0;
        oKr[s] = (row < m && mi < n) ? R::oK + row * n + mi : R::SKD - 1;
        oDr[s] = (row < m && mi < n) ? R::oD + row * n + mi : R::SKD - 1;
        oSr[s] = (row < m && mi < m) ? R::oS + row * m + mi : R::SKD - 1;
        quy[s] = (row < m && mi < n) ? sidx(mi, n + row, NZ) : -1;
        quu[s] = (row < m && mi < m) ? sidx(n + row, n + mi, NZ) : -1;
    }
    const int oLA0 = (mi < m) ? mi * m : 0;   // L^-1 as the A operand of L^-1 X: lane (i = mi, k = mq + 4 s) holds Li[mi][k]
    double* Lw = K.sHh;                       // m x m scratch for L^-1 (the H / Z buffers of the VALU path are unused here)
    for (int e = tid; e < m * m; e += 64) Lw[e] = 0.0;
    int doff[RT];   // (SPR) where entry tid + 64 r of the compact record sits in the dense n x NZ operand buffer
    if constexpr (SPR) {
#pragma unroll
        for (int r = 0; r < RT; r++) {
            const int e = tid + 64 * r;
            int o = 0;
#pragma unroll
            for (int i = 0; i < n; i++) {
#pragma unroll
                for (int j = 0; j < n; j++) if (T::Mnz(i, j)) o = (e == SP::pos_phi(i, j)) ? i * NZ + j : o;
#pragma unroll
                for (int j = 0; j < m; j++) if (T::Gnz(i, j)) o = (e == SP::pos_gam(i, j)) ? i * NZ + n + j : o;
            }
            doff[r] = o;
        }
        for (int e = tid; e < 2 * NPG; e += 64) K.sPG[e] = 0.0;   // (the structural zeros of both buffers, once)
        K.sync();
#pragma unroll
        for (int r = 0; r < RT; r++) {
            const int e = tid + 64 * r;
            if (e < SP::NS) K.sPG[(kHi & 1) * NPG + doff[r]] = K.PGS[(size_t)kHi * SP::S + e];
        }
    } else {
#pragma unroll
    for (int r = 0; r < RT; r++) {
        const int e = tid + 64 * r;
        if (e < NPG) K.sPG[(kHi & 1) * NPG + e] = K.PGk(kHi)[e];
    }
    }
#pragma unroll
    for (int r = 0; r < RN; r++) {
        const int e = tid + 64 * r;
        if (!adj_rt)
        if (e < NN) { K.Paft[(size_t)kHi * R::SNN + e] = 0.0; K.Piaft[(size_t)kHi * R::SNN + e] = (SEG && isA && e / n == e % n) ? 1.0 : 0.0; }
    }
    double qc[KS + 2 * MS], qn[KS + 2 * MS], pgn[RT];
    auto gather = [&](int kk, double* q) {   // stage cost QQ_kk in the accumulator layout of the H tiles (clamped gathers)
        const double* rec = K.QQ + (size_t)kk * R::SQQ;
#pragma unroll
        for (int r = 0; r < KS; r++) q[r] = rec[qyy[r] < 0 ? 0 : qyy[r]];
#pragma unroll
        for (int s = 0; s < MS; s++) { q[KS + s] = rec[quy[s] < 0 ? 0 : quy[s]]; q[KS + MS + s] = rec[quu[s] < 0 ? 0 : quu[s]]; }
    };
    gather(kHi, qc);
#pragma unroll
    for (int r = 0; r < RT; r++) pgn[r] = 0.0;
    v4d Pt = {0, 0, 0, 0}, Pit = {0, 0, 0, 0}, Gdt = {0, 0, 0, 0};
    if constexpr (SEG) {
        if (isA) {
#pragma unroll
            for (int q = 0; q < KS; q++) Pit[q] = (mq + 4 * q == mi && mi < n) ? 1.0 : 0.0;   // Pi = I behind the chain's last knot
        }
    }
    K.sync();
    for (int k = kHi; k >= kLo; k--) {
        const double* PGs = pg_buf<MODEL>(K, k);
        gather((k > kLo) ? k - 1 : kLo, qn);
        if constexpr (SPR) {
            const auto pg = K.PGS + (size_t)((k > kLo) ? k - 1 : kLo) * SP::S;
#pragma unroll
            for (int r = 0; r < RT; r++) { const int e = tid + 64 * r; pgn[r] = pg[(e < SP::NS) ? e : SP::NS - 1]; }
        } else {
            const auto pg = K.PGk((k > kLo) ? k - 1 : kLo);
#pragma unroll
            for (int r = 0; r < RT; r++) { const int e = tid + 64 * r; pgn[r] = pg[(e < NPG) ? e : NPG - 1]; }
        }
        FT_(PF_FPRE);
        double F[KS], G[KS], GA[MS];
#pragma unroll
        for (int q = 0; q < KS; q++) {
            const int row = mq + 4 * q;
            const double* pa = (c15 && row < n) ? K.cv + k * n + row : PGs + oF[q];
            const double a = *pa, b = PGs[oG[q]];
            F[q] = vF[q] ? a : 0.0; G[q] = vG[q] ? b : 0.0;
        }
#pragma unroll
        for (int s = 0; s < MS; s++) { const double a = PGs[oGA[s]]; GA[s] = vGA[s] ? -a : 0.0; }
        // (every LDS operand of the stage requested before the first product: left alone hipcc pairs each ds_read with the
        // MFMA that consumes it and pays the LDS latency once per pair)
        __builtin_amdgcn_sched_barrier(0);
        FX_(0, F[0] + G[0] + GA[0]);   // operand gathers from LDS landed
        v4d tph = {0, 0, 0, 0}, tga = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < KS; q++) {               // T_Phi = P Phi (column 15: P c), T_Gam = P Gam
            tph = __builtin_amdgcn_mfma_f64_16x16x4f64(Pt[q], F[q], tph, 0, 0, 0);
            tga = __builtin_amdgcn_mfma_f64_16x16x4f64(Pt[q], G[q], tga, 0, 0, 0);
        }
        FX_(1, tph[0] + tga[0]);       // T = P [Phi Gam] done
        v4d hyy = {0, 0, 0, 0}, huy = {0, 0, 0, 0}, huu = {0, 0, 0, 0}, zy = {0, 0, 0, 0}, zu = {0, 0, 0, 0};
#pragma unroll
        for (int r = 0; r < KS; r++) hyy[r] = qyy[r] < 0 ? 0.0 : qc[r];
#pragma unroll
        for (int s = 0; s < MS; s++) { huy[s] = quy[s] < 0 ? 0.0 : qc[KS + s]; huu[s] = quu[s] < 0 ? 0.0 : qc[KS + MS + s]; }
#pragma unroll
        for (int q = 0; q < KS; q++) {
            huu = __builtin_amdgcn_mfma_f64_16x16x4f64(G[q], tga[q], huu, 0, 0, 0);     // Gam^T T_Gam
            huy = __builtin_amdgcn_mfma_f64_16x16x4f64(G[q], tph[q], huy, 0, 0, 0);     // Gam^T T_Phi
            zu = __builtin_amdgcn_mfma_f64_16x16x4f64(G[q], Pit[q], zu, 0, 0, 0);       // Gam^T Pi
            hyy = __builtin_amdgcn_mfma_f64_16x16x4f64(F[q], tph[q], hyy, 0, 0, 0);     // Phi^T T_Phi
            zy = __builtin_amdgcn_mfma_f64_16x16x4f64(F[q], Pit[q], zy, 0, 0, 0);       // Phi^T Pi (row 15: c^T Pi)
        }
        FX_(2, hyy[0] + huy[0] + huu[0] + zy[0] + zu[0]);   // H, Z done
        // the two matrix-vector products of the stage, for the stage-parallel blocks: r_k = P_k c_k, Pi_k^T c_k
#pragma unroll
        for (int q = 0; q < KS; q++) { const int row = mq + 4 * q; if (c15 && row < n) K.rv[k * n + row] = tph[q]; }
        if (mq == 3 && mi < n) K.nun[k * n + mi] = zy[3];
        if (k == N - 1) {   // + E = [M^T C^T; b^T M^T C^T], M = (Phi + I)/2, M b = Gam/2; column g only for goal coordinates
            const bool gg = mi < n && K.is_goal(mi < n ? mi : 0);
#pragma unroll
            for (int r = 0; r < 4; r++) {
                const int row = mq + 4 * r;
                if (gg && row < n) zy[r] += 0.5 * (PGs[mi * NZ + row] + ((row == mi) ? 1.0 : 0.0));
                if (gg && row < m) zu[r] += 0.5 * PGs[mi * NZ + n + row];
            }
        }
        FT_(PF_FAB);
        double S[m * m], Li[m * m];
#pragma unroll
        for (int i = 0; i < m; i++)
#pragma unroll
            for (int j = 0; j < m; j++) {   // H_uu[a][b], a <= b: lane (a & 3) << 4 | b, register a >> 2 of the tile
                const int a = i < j ? i : j, b = i < j ? j : i;
                S[i * m + j] = readlane_f64(huu[a >> 2], ((a & 3) << 4) | b);
            }
        if (!chol_inv<m>(S, Li)) *fail = 1.0;
        FT_(PF_F4);
        // L^-1 is wave-uniform: one lane parks it in LDS, every lane takes its entries of the two operand layouts
        if (tid == 0) {
#pragma unroll
            for (int a = 0; a < m; a++)
#pragma unroll
                for (int c = 0; c <= a; c++) Lw[a * m + c] = Li[a * m + c];
        }
        K.sync();
        double LiA[MS], LiT[MS];
#pragma unroll
        for (int s = 0; s < MS; s++) {
            const int kk = mq + 4 * s;
            const bool v = mi < m && kk < m;
            const double a = Lw[v ? oLA0 + kk : 0], b = Lw[v ? kk * m + mi : 0];
            LiA[s] = v ? a : 0.0;   // A[i = mi][k = kk] = Li[mi][kk]:   L^-1 X
            LiT[s] = v ? b : 0.0;   // A[i = mi][k = kk] = Li[kk][mi]:   L^-T X  (and, as a B operand, L^-1 itself)
        }
        FX_(3, LiA[0] + LiT[0]);       // L^-1 through LDS
        // take the prefetched QQ_{k-1} before this stage's stores are issued (see factor_sweep_1w)
#pragma unroll
        for (int e = 0; e < KS + 2 * MS; e++) qc[e] = qn[e];
        __builtin_amdgcn_sched_barrier(0);
        v4d W = {0, 0, 0, 0}, V = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < MS; s++) {
            W = __builtin_amdgcn_mfma_f64_16x16x4f64(LiA[s], huy[s], W, 0, 0, 0);      // W = L^-1 H_uy
            V = __builtin_amdgcn_mfma_f64_16x16x4f64(LiA[s], zu[s], V, 0, 0, 0);       // V = L^-1 Z_u
        }
        FX_(4, W[0] + V[0] + qc[0]);   // W, V done (and the QQ prefetch taken)
        v4d Kt = {0, 0, 0, 0}, Dt = {0, 0, 0, 0}, Si = {0, 0, 0, 0};
#pragma unroll
        for (int s = 0; s < MS; s++) {
            const double wn = -W[s];
            hyy = __builtin_amdgcn_mfma_f64_16x16x4f64(wn, W[s], hyy, 0, 0, 0);        // P'  = H_yy - W^T W
            zy = __builtin_amdgcn_mfma_f64_16x16x4f64(wn, V[s], zy, 0, 0, 0);          // Pi' = Z_y  - W^T V
            Gdt = __builtin_amdgcn_mfma_f64_16x16x4f64(V[s], V[s], Gdt, 0, 0, 0);      // Gd += V^T V
            Kt = __builtin_amdgcn_mfma_f64_16x16x4f64(LiT[s], W[s], Kt, 0, 0, 0);      // K = L^-T W
            Dt = __builtin_amdgcn_mfma_f64_16x16x4f64(LiT[s], V[s], Dt, 0, 0, 0);      // D = L^-T V
            Si = __builtin_amdgcn_mfma_f64_16x16x4f64(LiT[s], LiT[s], Si, 0, 0, 0);    // S^-1 = L^-T L^-1
        }
        FX_(5, hyy[0] + zy[0] + Gdt[0] + Kt[0] + Dt[0] + Si[0]);   // P', Pi', Gd, K, D, S^-1 done
        v4d Ph = {0, 0, 0, 0};
#pragma unroll
        for (int q = 0; q < KS; q++) Ph[q] = F[q];
#pragma unroll
        for (int s = 0; s < MS; s++) Ph = __builtin_amdgcn_mfma_f64_16x16x4f64(GA[s], Kt[s], Ph, 0, 0, 0);   // Phi - Gam K
        FX_(6, Ph[0]);                 // Phicl done
        FT_(PF_F6);
        Pt = hyy; Pit = zy;
        // records (unconditional stores: lanes outside a matrix aim at the padding slot of the record)
        {
            double* phr = K.Phicl + (size_t)k * R::SNN;
            // (record -1 exists for k == 0; chain B's record kLo - 1 belongs to chain A: its last stage aims at the junk record -1)
            const int krec = (SEG && kLo > 0 && k == kLo) ? -1 : k - 1;
            double* par = K.Paft + (size_t)krec * R::SNN;
            double* pir = K.Piaft + (size_t)krec * R::SNN;
            double* kdr = K.KD + (size_t)k * R::SKD;
#pragma unroll
            for (int q = 0; q < KS; q++) {
                phr[oN[q]] = Ph[q];
                if constexpr (!NOPP) { par[oNT[q]] = hyy[q]; pir[oNT[q]] = zy[q]; }
            }
#pragma unroll
            for (int s = 0; s < MS; s++) { kdr[oKr[s]] = Kt[s]; kdr[oDr[s]] = Dt[s]; kdr[oSr[s]] = Si[s]; }
        }
        if (k > kLo) {
#pragma unroll
            for (int r = 0; r < RT; r++) {
                const int e = tid + 64 * r;
                if constexpr (SPR) { if (e < SP::NS) K.sPG[((k - 1) & 1) * NPG + doff[r]] = pgn[r]; }
                else if (e < NPG) K.sPG[((k - 1) & 1) * NPG + e] = pgn[r];
            }
        }
        FT_(PF_F7);
        K.sync();
        FT_(PF_FCD);
    }
    // Gd = sum V^T V for the goal system of the mid phase
    if constexpr (SEG) {
        const LPtr<double> L = K.lds;
#pragma unroll
        for (int q = 0; q < KS; q++) {
            const int row = mq + 4 * q;
            if (row < n && mi < n) {
                if (oGd < 0) K.sGd[row * n + mi] = Gdt[q]; else L[oGd + row * n + mi] = Gdt[q];
                L[oP + row * n + mi] = Pt[q]; L[oPi + row * n + mi] = Pit[q];
            }
        }
        K.sync();
        return;
    }
#pragma unroll
    for (int q = 0; q < KS; q++) { const int row = mq + 4 * q; if (row < n && mi < n) K.sGd[row * n + mi] = Gdt[q]; }
    K.sync();
}

// Affine vector recurrences of the one-wave path.  The wave is split into C = 64/n groups of n lanes; group g
// holds the operands of knot (k0 -+ g) of the current chunk of C knots, so ONE batch of global loads feeds C knots
// of the dependency chain (the recurrences are memory-latency bound otherwise), and the next chunk is fetched
// while the current one computes.  The n-vector travels between groups with v_readlane (no LDS on the chain).
//
// backward: pt_{k-1} = Phicl_k^T pt_k + qq_k, k = N-1..1, pt := p + r (so qq_k = qt_k + r_{k-1}), pt_{N-1} = r_{N-1}.
//           pv[k] holds qq_k on entry and pt_k on exit.
#ifndef GUSTO_SWEEP_RING
#define GUSTO_SWEEP_RING 4
#endif

}  // namespace gusto
