// seg.hpp -- the KKT solve of one interior point iteration as TWO independent Riccati segments (round 6).
//
// The sequential phases of a KKT solve -- the factor sweep and the four vector sweeps -- are N-stage dependent chains run by
// one wave that issues ~40 % of the time.  Here the horizon is split at knot s = N / 2:
//   chain B = stages s .. N-1: the recursion as it was (P = Pi = 0 behind the last knot, the goal rows' E term there);
//   chain A = stages 0 .. s-1: the SAME recursion started from P = 0 with its end state dy_{s-1} = xi adjoined as a terminal
//             equality, multiplier lam = the costate of trapezoid row s -- the machinery the goal rows already use: Pi starts
//             as I, no E term, and "Gd" of the chain is the compliance Gd_A = d xi / d lam of the segment.
// The two chains are independent and run interleaved in the one instruction stream; what joins them is ONE coarse LQR stage:
// maximised over lam, chain A costs 1/2 w' Gd_A^-1 w with w = xi - th_A (th_A: the end state A reaches for lam = lam0), a free
// n-dim control in front of chain B's cost-to-go (P_B, p_B, Pi_B, Gd_B).  With Gd_A = G G' (Cholesky, floored pivots: a segment
// may have an uncontrollable direction) and M = I + G' P_B G (SPD, eigenvalues >= 1):
//       Sig = (Gd_A^-1 + P_B)^-1 = G M^-1 G',    Tt = (I + Gd_A P_B)^-1 = G M^-1 G^-1,    Pa = Tt' P_B,
//       Gdc = Gd_B + Pi_B' Sig Pi_B   (the goal Hessian of the WHOLE horizon, as the sequential recursion forms it),
//       w1 = Tt a - Sig ph,   mu_g = Gdc^-1 (th_B + Pi_B' w1),   xi = w1 - Sig Pi_B mu_g,   dlam = Pa a + Tt' ph + Tt' Pi_B mu_g
// with a = th_A, ph = p_B - lam0: products only -- Gd_A^-1 is never formed, and no output is a difference of large terms
// (a stiff cost-to-go behind the interface, barrier weights ~1/mu, against a soft segment in front of it is the normal case at
// the end of a solve).  lam0 = the CURRENT costate iterate nu_s: chain A's backward vector sweep starts from it, so that a, ph
// and every intermediate vanish with the Newton step.  Derivation and the CPU prototype that was run against the oracle's
// sequential recursion on all four models first: tools/proto/segriccati.c, profiles/r06_segmented_riccati_proto.txt.
// Reference path: the convex subproblem of scp_gusto.jl:104,178-314 (JuMP.optimize!).
#pragma once

namespace gusto {

#ifndef GUSTO_SEG_PIV
#define GUSTO_SEG_PIV 1e-13     // pivot floor of chol(Gd_A), relative to its largest diagonal entry (prototype: 1e-12 .. 1e-14 alike, 1e-16 fails)
#endif

// which kernels split their horizon: the one-wave double-integrator kernel (freeflyerSE2)
template <int MODEL> constexpr bool seg2_model() {
    return GUSTO_SEG2 && MT<MODEL>::PG2 && MT<MODEL>::LTI && MT<MODEL>::NDEF == 0 && LdsC<MODEL, true>::KD_LDS && !MT<MODEL>::SWEEP_CALL;
}

// LDS map of the segmented solve (offsets in doubles from the base of the dynamic LDS).  Chain A's working set of the factor
// sweep lives in the [Phi Gam] staging buffers the PG2 sweep never uses; the coarse stage's matrices take the place of the
// sweeps' working set, which is dead between two factor sweeps.
template <int MODEL> struct SegC {
    using T = MT<MODEL>;
    using C = LdsC<MODEL, true>;
    static constexpr int n = T::n, m = T::m, NZ = n + m, NN = n * n, NZN = NZ * n;
    // chain A, factor sweep: P | Pi | Z | rows 0..3 of H in the staging buffers, rows 4..5 of H behind Gd
    static constexpr int aP = C::sPG, aPi = aP + NN, aZ = aPi + NN, aH0 = aZ + NZN, aH1 = C::sGd + NN, HSPLIT = 4;
    static_assert(!seg2_model<MODEL>() || (aH0 + HSPLIT * NZ <= C::sT0 && (n - HSPLIT) * NZ <= NN && aP % 2 == 0 && aPi % 2 == 0), "chain A working set");
    // inputs of the coarse stage as the factor sweep leaves them
    static constexpr int PB = C::sP /* P_B, columns interleaved */, PIB = C::sPi /* Pi_B, transposed */, GDB = C::sGd, GDA = C::sGd + NN;
    // its outputs (until the next factor sweep) and scratch
    static constexpr int Tt = C::sPG, Sg = Tt + NN, Pa = Sg + NN, Gci = Pa + NN, A1 = Gci + NN, A2 = A1 + NN, A3 = A2 + NN, X1 = A3 + NN, X2 = X1 + NN;
    static_assert(!seg2_model<MODEL>() || X2 + NN <= C::sGd, "coarse stage scratch");
    // vectors in the misc block (Blk: [0..7] reductions, 8 fail, 16.. gxs, 32.. mug, 48.. mugn, 64.. goal values)
    static constexpr int XI = C::misc + 22 /* xi = dy_{s-1} */, PBV = C::misc + 38 /* p_B */, LAM = C::misc + 54 /* dlam */;
    static_assert(n <= 8, "misc slots");
};

// The coarse stage's matrices from what the factor sweep left (head of this file).  Whole wave, an entry per lane:
//   X = I + P_B Gd_A,  Ta' = X^-1 by Gauss-Jordan in LDS (no pivoting: X is similar to the SPD matrix I + G' P_B G; the CPU prototype
//   ran every BASELINE batch with and without partial pivoting, profiles/r06_segmented_riccati_proto.txt; a zero pivot
//   ends the solve as a failed factorisation),  Sig = Gd_A Ta',  Pa = Ta' P_B,  A3 = Ta' Pi_B,  A2 = Sig Pi_B,
//   Gdc = Gd_B + Pi_B' A2 (identity on the coordinates without a point goal),  Gdc^-1 by Cholesky,  A1 = Gdc^-1 Pi_B'.
// (Ta' of this comment is the matrix the head of the file calls Tt' = (I + P_B Gd_A)^-1; LDS slot Tt holds its TRANSPOSE.)
template <int MODEL, class BLK> GD void seg_coarse_factor(BLK& K, double* fail) {
    using S = SegC<MODEL>;
    constexpr int n = S::n, NN = S::NN, h3 = n / 2;
    static_assert(NN <= 64, "an entry per lane");
    const LPtr<double> L = K.lds;
    const int tid = K.tid;
    const bool on = tid < NN;
    const int i = on ? tid / n : 0, j = on ? tid % n : 0;
    auto pcol = [](int j_) { return 2 * (j_ % h3) + j_ / h3; };
    auto rm = [](int base) { return [base](int r_, int c_) { return base + r_ * n + c_; }; };     // row-major
    auto tr = [](int base) { return [base](int r_, int c_) { return base + c_ * n + r_; }; };     // the transpose of a row-major matrix
    auto pb = [&](int r_, int c_) { return S::PB + r_ * n + pcol(c_); };                          // P_B (columns interleaved)
    auto pib = [](int r_, int c_) { return S::PIB + c_ * n + r_; };                               // Pi_B (stored transposed)
    auto pibT = [](int r_, int c_) { return S::PIB + r_ * n + c_; };                              // Pi_B'
    // dst(i, j) = add + sum_l A(i, l) B(l, j), an entry per lane
    auto prod = [&](auto A, auto B, double add) {
        double acc = add;
#pragma unroll
        for (int l = 0; l < n; l++) acc += L[A(i, l)] * L[B(l, j)];
        return acc;
    };
    bool ok = true;
    // ---- W = X = I + P_B Gd_A, inverted in place (slot X1): lane (i, j) keeps its entry in a register ----
    double w = prod(pb, rm(S::GDA), (i == j) ? 1.0 : 0.0);
    if (on) L[S::X1 + tid] = w;
    K.sync();
    static_for<0, n>([&](auto Cc) {
        constexpr int c = decltype(Cc)::value;
        const double piv = L[S::X1 + c * n + c], wcj = L[S::X1 + c * n + j], wic = L[S::X1 + i * n + c];
        if (!(fabs(piv) > 0.0) || !isfinite(piv)) ok = false;   // (pivots of either sign occur: X is not symmetric)
        const double d = rcp_nr(piv);
        const double rowc = (j == c) ? d : wcj * d;                      // the new row c
        const double other = (j == c) ? -(wic * d) : w - wic * (wcj * d);
        w = (i == c) ? rowc : other;
        K.sync();
        if (on) L[S::X1 + tid] = w;
        K.sync();
    });
    // X1 = (I + P_B Gd_A)^-1 =: Y.  Tt = Y' (slot Tt: row-major transpose), Sig = Gd_A Y, Pa = Y P_B, A3 = Y Pi_B
    {
        const double sg = prod(rm(S::GDA), rm(S::X1), 0.0);
        const double pa = prod(rm(S::X1), pb, 0.0);
        const double a3 = prod(rm(S::X1), pib, 0.0);
        if (on) { L[S::Tt + j * n + i] = w; L[S::Sg + tid] = sg; L[S::Pa + tid] = pa; L[S::A3 + tid] = a3; }
    }
    K.sync();
    {   // Sig is symmetric in exact arithmetic: both triangles from one mean
        const double sgm = 0.5 * (L[S::Sg + i * n + j] + L[S::Sg + j * n + i]);
        K.sync();
        if (on) L[S::Sg + tid] = sgm;
        K.sync();
    }
    {
        const double a2 = prod(rm(S::Sg), pib, 0.0);                      // A2 = Sig Pi_B
        if (on) L[S::A2 + tid] = a2;
    }
    K.sync();
    {   // Gdc = Gd_B + Pi_B' A2 -> X2 (identity on the coordinates without a point goal)
        double g = L[S::GDB + (on ? tid : 0)] + prod(pibT, rm(S::A2), 0.0);
        if (!K.is_goal(i) || !K.is_goal(j)) g = (i == j) ? 1.0 : 0.0;
        if (on) L[S::X2 + tid] = g;
    }
    K.sync();
    {   // Gdc^-1: every lane factors the n x n block itself (broadcast reads; the flops of ONE lane), lane 0 publishes L^-1
        double G[n * n], Li[n * n];
#pragma unroll
        for (int e = 0; e < n * n; e++) G[e] = L[S::X2 + e];
        if (!chol_inv<n>(G, Li)) ok = false;
        K.sync();
        if (tid == 0) {
#pragma unroll
            for (int e = 0; e < n * n; e++) L[S::X2 + e] = Li[e];
        }
        K.sync();
        const double gi = prod(tr(S::X2), rm(S::X2), 0.0);               // Lc^-T Lc^-1
        if (on) L[S::Gci + tid] = gi;
    }
    K.sync();
    {
        const double a1 = prod(rm(S::Gci), pibT, 0.0);                    // A1 = Gdc^-1 Pi_B'
        if (on) L[S::A1 + tid] = a1;
    }
    if (!ok) *fail = 1.0;
    K.sync();
}

// ---- the factor sweep of the two chains ----------------------------------------------------------------------------------
// factor_sweep_pg2 (ipm.hpp: the software-pipelined, predication-free stage of the double integrator) for TWO chains at once:
// every phase of a stage is issued for chain A and chain B back to back -- operand loads of both, then the arithmetic of both,
// then the stores of both, one ordering point -- so that the LDS round trips, the v_readlane exchanges and the dependent flops of
// the 3 x 3 Cholesky of one chain run in the shadow of the other's.  Per chain the operations and their order are those of
// factor_sweep_pg2.
//   Chain B (stages N-1 .. s): the sweep as it was, but its last stage (k = s) and its last tail leave P_B, Pi_B in LDS only --
//     record s-1 belongs to chain A (their stores aim at the junk record -1).
//   Chain A (stages s-1 .. 0): P = 0, Pi = I behind its last knot (record s-1 = (0 | I)), no E term; its working set sits in the
//     [Phi Gam] staging buffers (SegC).
// On exit: P_B in sP, Pi_B in sPi, Gd_B in sGd, Gd_A behind it.
template <int MODEL> GD void factor_sweep_pg2s(SweepView<MODEL> K, double* fail, Prof& pf, int s) {
    using T = MT<MODEL>;
    using R = Rec<MODEL>;
    using C = LdsC<MODEL, true>;
    using S = SegC<MODEL>;
    constexpr int n = T::n, m = T::m, NZ = n + m, NQ = NZ * (NZ + 1) / 2, NN = n * n, NZN = NZ * n, h3 = n / 2, NH = n * (n + 1) / 2;
    static_assert(T::PG2 && T::LTI && C::KD_LDS && NQ <= 64 && NZN <= 64 && NN <= 64 && n >= 2 * m, "shape");
    static_assert(NH + NN < R::SNN && !C::BIG, "P | Pi record, dummy slot");
    static_assert(n * NZ >= 16 + 36, "dummy slot");
    const int tid = K.tid, N = K.N;
    // ---- lane roles (factor_sweep_pg2) ----
    const int ijh = K.lut[tid < NQ ? tid : 0], hc = ijh >> 8, hj = ijh & 255, i0 = T::pg_r0(hc), j0 = T::pg_r0(hj);   // H[hc][hj]
    const int zc = tid < NZN ? tid / n : 0, zg = tid < NZN ? tid % n : 0, z0 = T::pg_r0(zc);                       // Z[zc][zg]
    const int ri = tid < n ? tid : 0;                                                                             // r[ri], Pi^T c [ri]
    const bool on = tid < NN;
    const int i = on ? tid / n : 0, j = on ? tid % n : 0;                                                         // P[i][j], Pi[i][j]
    const LPtr<double> L = K.lds;
    typedef double v2d __attribute__((ext_vector_type(2)));
    auto ld2 = [&](int off) { return *(const __attribute__((address_space(3))) v2d*)(L.p + off); };   // ds_read_b128 (off even)
    const int dmy = C::sT0 + (tid & 15);             // dummy slot of this lane (+ immediates < 36 stay inside the T buffer)
    auto pcol = [](int j_) { return 2 * (j_ % h3) + j_ / h3; };
    // ---- LDS operands of the two chains ----
    struct Off { int oPP, oPZ, oPr, oPir, wH1, wZ, oHi, oHj, oPn, oZi, oZj, wP, wPi; };
    Off oB, oA;
    oB.oPP = C::sP + i0 * n + 2 * j0; oB.oPZ = C::sPi + zg * n + z0; oB.oPr = C::sP + ri * n; oB.oPir = C::sPi + ri * n;
    oB.wH1 = tid < NQ ? C::sHh + hc * NZ + hj : dmy;
    oB.wZ = tid < NZN ? C::sZ + tid : dmy;
    oB.oHi = C::sHh + i * NZ + n; oB.oHj = C::sHh + j * NZ + n; oB.oPn = C::sHh + (i < j ? i : j) * NZ + (i < j ? j : i);
    oB.oZi = C::sZ + n * n + i; oB.oZj = C::sZ + n * n + j;
    oB.wP = on ? C::sP + i * n + pcol(j) : dmy; oB.wPi = on ? C::sPi + j * n + i : dmy;
    auto hA = [](int r_, int c_) { return (r_ < S::HSPLIT ? S::aH0 + r_ * NZ : S::aH1 + (r_ - S::HSPLIT) * NZ) + c_; };   // (rows 0..n-1 of H only: S travels by v_readlane)
    oA.oPP = S::aP + i0 * n + 2 * j0; oA.oPZ = S::aPi + zg * n + z0; oA.oPr = S::aP + ri * n; oA.oPir = S::aPi + ri * n;
    oA.wH1 = (tid < NQ && hc < n) ? hA(hc, hj) : dmy;
    oA.wZ = tid < NZN ? S::aZ + tid : dmy;
    oA.oHi = hA(i, n); oA.oHj = hA(j, n); oA.oPn = hA(i < j ? i : j, i < j ? j : i);
    oA.oZi = S::aZ + n * n + i; oA.oZj = S::aZ + n * n + j;
    oA.wP = on ? S::aP + i * n + pcol(j) : dmy; oA.wPi = on ? S::aPi + j * n + i : dmy;
    const int vecs = C::vecs, oCv = vecs + 3 * N * n, oRv = vecs + 4 * N * n, oNun = vecs + 6 * N * n;   // (Blk::rebind_lds)
    const int wRv = tid < n ? oRv + tid : dmy, wNun = tid < n ? oNun + tid : dmy, sRv = tid < n ? n : 0;
    const int kdo = K.kd_off;
    const int wKD = tid < n ? kdo + tid : dmy, sKD = tid < n ? C::KDS : 0;             // K[a][tid] at + a n, D[a][tid] at + (m + a) n
    const int wSi = tid == 0 ? kdo + 2 * m * n : dmy, sSi = tid == 0 ? C::KDS : 0;     // S^-1, upper triangle
    const int ep = (on && i <= j) ? sidx(i, j, n) : R::SNN - 1, eq = on ? NH + tid : R::SNN - 1;   // packed P | Pi record
    const int qi = tid < NQ ? tid : 0;
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
    // E = [M^T C^T; b^T M^T C^T] of the last knot (chain B), column g only for goal coordinates
    const double eterm = pg_main(zg, zc) + ((zc == zg) ? 1.0 : 0.0);
    const bool egoal = K.is_goal(zg);
    bool okall = true;
    auto msync = [&]() { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); };

    // ---- what a chain keeps across stages: the prefetched stage cost and this lane's entry of Gd ----
    struct St { double qq, gdR; };
    struct Tmp {   // operands and intermediate results of one stage
        double p00, p01, p10, p11, ra[n], rb[n], pa[n], zb0, zb1, qqn, h, z;
        double hi[m], hjv[m], zi[m], zj[m], pn;
    };
    const int nB = N - s, nA = s;   // stages of the chains (nB - nA = 0 or 1)
    St stB, stA;
    // start: P = 0 behind the last knot of a chain; Pi = 0 (chain B) / I (chain A)
    L[oB.wP] = 0.0; L[oB.wPi] = 0.0;
    L[oA.wP] = 0.0; L[oA.wPi] = (on && i == j) ? 1.0 : 0.0;
    stB.gdR = 0.0; stA.gdR = 0.0;
    if (tid < R::SNN) {
        K.Paft[(size_t)(N - 1) * R::SNN + tid] = 0.0;
        K.Paft[(size_t)(s - 1) * R::SNN + tid] = (tid >= NH && tid < NH + NN && (tid - NH) / n == (tid - NH) % n) ? 1.0 : 0.0;
    }
    stB.qq = K.kdl[(N - 1) * C::KDS + qi];
    stA.qq = K.kdl[(s - 1) * C::KDS + qi];
    K.sync();

    // ---- the two halves of a stage.  (factor_sweep_pg2 runs the goal chain half a stage behind the value chain to fill the
    // latency of the Cholesky; here the other CHAIN fills it, and a stage is H, Z, r, Pi' c | ordering point | L, W, K, P', V, D,
    // Pi', Gd | ordering point -- the same operations per chain, one exchange through LDS less per stage.) ----
    auto h1_load = [&](const Off& o, Tmp& t, int k, int klo) {
        const v2d pA = ld2(o.oPP), pB = ld2(o.oPP + h3 * n);
        t.p00 = pA.x; t.p01 = pA.y; t.p10 = pB.x; t.p11 = pB.y;
#pragma unroll
        for (int l = 0; l < n; l += 2) {
            const v2d a2 = ld2(o.oPr + l), b2 = ld2(oCv + k * n + l), c2 = ld2(o.oPir + l);
            // row ri of P holds the columns in the order 0, n/2, 1, n/2 + 1, ...: ra[] back in natural order
            t.ra[l / 2] = a2.x; t.ra[l / 2 + h3] = a2.y; t.rb[l] = b2.x; t.rb[l + 1] = b2.y; t.pa[l] = c2.x; t.pa[l + 1] = c2.y;
        }
        t.zb0 = L[o.oPZ]; t.zb1 = L[o.oPZ + h3];
        t.qqn = K.kdl[((k > klo) ? k - 1 : klo) * C::KDS + qi];   // (slot k-1 still holds QQ_{k-1})
    };
    auto h1_comp = [&](const Off& o, const St& st, Tmp& t, int k, const PGC& c, bool last) {
        t.h = st.qq + c.a0 * (c.b0 * t.p00 + c.b1 * t.p01) + c.a1 * (c.b0 * t.p10 + c.b1 * t.p11);
        L[o.wH1] = t.h;
        double z = c.zv0 * t.zb0 + c.zv1 * t.zb1;
        const double zE = fma(0.5, eterm, z);
        z = (last && egoal) ? zE : z;
        L[o.wZ] = z;
        t.z = z;
        double rr = 0, rp = 0;
#pragma unroll
        for (int l = 0; l < n; l++) rr += t.ra[l] * t.rb[l];
#pragma unroll
        for (int l = 0; l < n; l++) rp += t.pa[l] * t.rb[l];
        L[wRv + k * sRv] = rr;
        L[wNun + k * sRv] = rp;
    };
    auto h2_load = [&](const Off& o, Tmp& t, double* Sm) {
#pragma unroll
        for (int a = 0; a < m; a++)
#pragma unroll
            for (int b = 0; b < m; b++) {
                const int e = sidx(n + (a < b ? a : b), n + (a < b ? b : a), NZ);
                Sm[a * m + b] = readlane_f64(t.h, e);
            }
#pragma unroll
        for (int l = 0; l < m; l++) { t.hi[l] = L[o.oHi + l]; t.hjv[l] = L[o.oHj + l]; t.zi[l] = L[o.oZi + l * n]; t.zj[l] = L[o.oZj + l * n]; }
        t.pn = L[o.oPn];
    };
    // L = chol(H_uu) is in Li; W = L^-1 H_uy, K = L^-T W, P' = H_yy - W^T W | V = L^-1 Z_u, D = L^-T V, Pi' = Z_y - W^T V, Gd += V^T V
    auto h2_comp = [&](const Off& o, St& st, Tmp& t, const double* Li, int k, int rec) {
        double wi[m], wj[m], kj[m], vi[m], vj[m], dj[m];
#pragma unroll
        for (int a = 0; a < m; a++) {
            double s1 = 0, s2 = 0, s3 = 0, s4 = 0;
#pragma unroll
            for (int l = 0; l <= a; l++) {
                s1 += Li[a * m + l] * t.hi[l]; s2 += Li[a * m + l] * t.hjv[l];
                s3 += Li[a * m + l] * t.zi[l]; s4 += Li[a * m + l] * t.zj[l];
            }
            wi[a] = s1; wj[a] = s2; vi[a] = s3; vj[a] = s4;
        }
#pragma unroll
        for (int a = 0; a < m; a++) {
            double s1 = 0, s2 = 0;
#pragma unroll
            for (int l = a; l < m; l++) { s1 += Li[l * m + a] * wj[l]; s2 += Li[l * m + a] * vj[l]; }
            kj[a] = s1; dj[a] = s2;
        }
        double pn = t.pn, pin = t.z, gd = st.gdR;
#pragma unroll
        for (int l = 0; l < m; l++) pn -= wi[l] * wj[l];
#pragma unroll
        for (int l = 0; l < m; l++) { pin -= wi[l] * vj[l]; gd += vi[l] * vj[l]; }
        L[o.wP] = pn; L[o.wPi] = pin; st.gdR = gd;
        K.Paft[(size_t)rec * R::SNN + ep] = pn;
        K.Paft[(size_t)rec * R::SNN + eq] = pin;
#pragma unroll
        for (int a = 0; a < m; a++) { L[wKD + k * sKD + a * n] = kj[a]; L[wKD + k * sKD + (m + a) * n] = dj[a]; }
#pragma unroll
        for (int a = 0; a < m; a++)   // S^-1 = L^-T L^-1, upper triangle (wave-uniform values)
#pragma unroll
            for (int b = 0; b <= a; b++) {
                double s1 = 0;
#pragma unroll
                for (int l = a; l < m; l++) s1 += Li[l * m + a] * Li[l * m + b];
                L[wSi + k * sSi + sidx(b, a, m)] = s1;
            }
        st.qq = t.qqn;
    };
    // chain B's stage k leaves record k - 1 (k = s: the junk record -1 -- record s-1 belongs to chain A), chain A's stage k record k - 1
    auto recB = [&](int k) { return k > s ? k - 1 : -1; };
    // one stage of chain B alone (odd N: chain B is one stage longer)
    auto stage1B = [&](int k, bool last) {
        Tmp t;
        double Sm[m * m], Li[m * m];
        h1_load(oB, t, k, s);
        h1_comp(oB, stB, t, k, cN, last);
        msync();
        h2_load(oB, t, Sm);
        okall = chol_inv<m>(Sm, Li) && okall;
        h2_comp(oB, stB, t, Li, k, recB(k));
        msync();
    };
    // one stage of both chains
    // GUSTO_STAGE_PROF (with GUSTO_PROFILE): time stamps INSIDE a stage, read asynchronously (s_memtime is issued where the wave
    // is, its result only waited for at the end of the stage)
#if defined(GUSTO_PROFILE) && defined(GUSTO_STAGE_PROF)
#define STAMP(v) const unsigned long long v = __builtin_amdgcn_s_memtime()
#define STAMPS_END() do { pf.acc[PF_FPRE] += (long long)(t1_ - t0_); pf.acc[PF_FAB] += (long long)(t2_ - t1_); \
                          pf.acc[PF_F4] += (long long)(t3_ - t2_); pf.acc[PF_F5] += (long long)(t4_ - t3_); pf.acc[PF_FCD] += (long long)(t5_ - t4_); } while (0)
#else
#define STAMP(v) do {} while (0)
#define STAMPS_END() do {} while (0)
#endif
    auto stage2 = [&](int kB, bool lastB, int kA, const PGC& cA) {
        Tmp tB, tA;
        double SmB[m * m], SmA[m * m], LiB[m * m], LiA[m * m];
        STAMP(t0_);
        h1_load(oB, tB, kB, s);
        h1_load(oA, tA, kA, 0);
        h1_comp(oB, stB, tB, kB, cN, lastB);
        h1_comp(oA, stA, tA, kA, cA, false);
        STAMP(t1_);
        msync();
        h2_load(oB, tB, SmB);
        h2_load(oA, tA, SmA);
        STAMP(t2_);
        okall = chol_inv<m>(SmB, LiB) && okall;
        okall = chol_inv<m>(SmA, LiA) && okall;
        STAMP(t3_);
        h2_comp(oB, stB, tB, LiB, kB, recB(kB));
        h2_comp(oA, stA, tA, LiA, kA, kA - 1);
        STAMP(t4_);
        msync();
        STAMP(t5_);
        STAMPS_END();
    };
#undef STAMP
#undef STAMPS_END
    int kB = N - 1;
    if (nB > nA) { stage1B(kB, true); kB--; }
    for (int t = 0; t + 1 < nA; t++) stage2(kB - t, kB - t == N - 1, s - 1 - t, cN);
    stage2(s, s == N - 1, 0, c0);
    if (on) { L[S::GDB + tid] = stB.gdR; L[S::GDA + tid] = stA.gdR; }
    if (!okall) *fail = 1.0;
    K.sync();
    (void)pf;
}

// ---- vector sweeps over a RANGE of knots (the one-wave sweeps of ipm.hpp for the double integrator, PHI_FROM_K) ----
// backward: pt_{k-1} = Phicl_k' pt_k + qq_k for k = khi .. klo, started from `start` (n values in LDS) as pt_khi; pv[k] holds qq_k
// on entry and pt_k on exit; the last output, pt_{klo-1}, goes to `last_out` (LDS offset) instead of pv[klo-1] when last_out >= 0.
template <int MODEL> GD void backward_sweep_rng(SweepView<MODEL> K, int khi, int klo, int start, int last_out) {
    using C = LdsC<MODEL, true>;
    constexpr int n = MT<MODEL>::n, m = MT<MODEL>::m, CG = 64 / n;
    static_assert(C::PHI_FROM_K, "the double integrator: Phicl rebuilt from K");
    const LPtr<double> L = K.lds;
    const int tid = K.tid;
    const int g = (tid < CG * n) ? tid / n : CG - 1, i = (tid < CG * n) ? tid % n : 0;
    double gl[n];
    {
        double Bd[n * m];
        Dyn<MODEL>::B(*K.mpp, Bd);
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
    for (int l = 0; l < n; l++) phc[l] = (l == i) ? 1.0 : ((i == l + n / 2) ? K.dt : 0.0);
    double col[n], coln[n], qv, qvn = 0, pval;
    auto fetch = [&](int k0, double* c, double& q) {
        const int kk = (k0 - g >= klo) ? k0 - g : klo;   // (clamped, unconditional loads)
        double kc[m];
#pragma unroll
        for (int a = 0; a < m; a++) kc[a] = K.kdl[kk * C::KDS + a * n + i];
#pragma unroll
        for (int l = 0; l < n; l++) c[l] = phc[l] - gl[l] * kc[l % m];
        q = K.pv[kk * n + i];
    };
    fetch(khi, col, qv);
    pval = L[start + i];
    K.sync();
    if (tid < n) K.pv[khi * n + tid] = pval;
    for (int k0 = khi; k0 >= klo; k0 -= CG) {
        if (k0 - CG >= klo) fetch(k0 - CG, coln, qvn);
#pragma unroll
        for (int gs = 0; gs < CG; gs++) {
            if (k0 - gs >= klo) {
                const int sg = (gs == 0) ? CG - 1 : gs - 1;
                double pbv[n];
#pragma unroll
                for (int l = 0; l < n; l++) pbv[l] = readlane_f64(pval, sg * n + l);
                __builtin_amdgcn_sched_barrier(0);
                double acc = qv;
#pragma unroll
                for (int l = 0; l < n; l++) acc += col[l] * pbv[l];
                pval = (g == gs) ? acc : pval;
            }
        }
        {   // group g produced pt_{kk-1}, kk = k0 - g
            const int kk = k0 - g;
            if (tid < CG * n && kk >= klo) {
                const int dst = (kk == klo && last_out >= 0) ? last_out + i : (int)(C::vecs + 2 * K.N * n) + (kk - 1) * n + i;
                L[dst] = pval;
            }
        }
#pragma unroll
        for (int l = 0; l < n; l++) col[l] = coln[l];
        qv = qvn;
    }
    K.sync();
}
// forward: dy_k = Phicl_k dy_{k-1} + ct_k for k = klo .. khi, dy_{klo-1} = `start` (n values in LDS; < 0: zero); dY[k] holds ct_k on
// entry and dy_k on exit
template <int MODEL> GD void forward_sweep_rng(SweepView<MODEL> K, int klo, int khi, int start) {
    using C = LdsC<MODEL, true>;
    constexpr int n = MT<MODEL>::n, m = MT<MODEL>::m, CG = 64 / n;
    const LPtr<double> L = K.lds;
    const int tid = K.tid;
    const int g = (tid < CG * n) ? tid / n : CG - 1, i = (tid < CG * n) ? tid % n : 0;
    double gi = 0.0;
    {
        double Bd[n * m];
        Dyn<MODEL>::B(*K.mpp, Bd);
        const double h = 0.5 * K.dt;
#pragma unroll
        for (int l = 0; l < n; l++) {
            const int c_ = l % m;
            const double hb = h * Bd[(c_ + n / 2) * m + c_];
            const double gv = (l < n / 2) ? 2.0 * (h * hb) : 2.0 * hb;
            gi = (i == l) ? gv : gi;
        }
    }
    double phr[n];   // row i of Phi
#pragma unroll
    for (int l = 0; l < n; l++) phr[l] = (i == l) ? 1.0 : ((l == i + n / 2) ? K.dt : 0.0);
    double row[n], rown[n], cv, cvn = 0;
    auto fetch = [&](int k0, double* r, double& c) {
        const int kk = (k0 + g <= khi) ? k0 + g : khi;
        const int ic = (i < m) ? i : i - m;
#pragma unroll
        for (int l = 0; l < n; l++) r[l] = phr[l] - gi * K.kdl[kk * C::KDS + ic * n + l];
        c = K.dY[kk * n + i];
    };
    fetch(klo, row, cv);
    // (every group starts from the start value; only group CG - 1 is read at step 0)
    double yval = (start >= 0) ? L[start + i] : 0.0;
    K.sync();
    for (int k0 = klo; k0 <= khi; k0 += CG) {
        if (k0 + CG <= khi) fetch(k0 + CG, rown, cvn);
#pragma unroll
        for (int gs = 0; gs < CG; gs++) {
            if (k0 + gs <= khi) {
                const int sg = (gs == 0) ? CG - 1 : gs - 1;
                double pbv[n];
#pragma unroll
                for (int l = 0; l < n; l++) pbv[l] = readlane_f64(yval, sg * n + l);
                __builtin_amdgcn_sched_barrier(0);
                double acc = cv;
#pragma unroll
                for (int l = 0; l < n; l++) acc += row[l] * pbv[l];
                yval = (g == gs) ? acc : yval;
            }
        }
        {
            const int kk = k0 + g;
            if (tid < CG * n && kk <= khi) K.dY[kk * n + i] = yval;
        }
#pragma unroll
        for (int l = 0; l < n; l++) row[l] = rown[l];
        cv = cvn;
    }
    K.sync();
}

// the two chains' sweeps (this version: one after the other)
template <int MODEL> GD void backward_sweep_seg(SweepView<MODEL> K, int s) {
    using C = LdsC<MODEL, true>;
    constexpr int n = MT<MODEL>::n;
    const int N = K.N;
    // chain B: from pt_{N-1} = r_{N-1}; its last output is p_B (the gradient of chain B's cost-to-go in dy_{s-1}; r_{s-1} belongs to chain A and is 0)
    backward_sweep_rng<MODEL>(K, N - 1, s, C::vecs + 4 * N * n + (N - 1) * n, SegC<MODEL>::PBV);
    // chain A: from pt_{s-1} = lam0 = the current costate iterate nu_s (r_{s-1} = 0: the chain starts from P = 0)
    backward_sweep_rng<MODEL>(K, s - 1, 1, C::vecs + 5 * N * n + s * n, -1);
}
template <int MODEL> GD void forward_sweep_seg(SweepView<MODEL> K, int s) {
    using C = LdsC<MODEL, true>;
    constexpr int n = MT<MODEL>::n;
    const int N = K.N;
    forward_sweep_rng<MODEL>(K, 0, s - 2, -1);                     // chain A: dy_0 .. dy_{s-2} (its end state is xi by construction)
    if (K.tid < n) K.dY[(s - 1) * n + K.tid] = K.lds[SegC<MODEL>::XI + K.tid];
    forward_sweep_rng<MODEL>(K, s, N - 1, SegC<MODEL>::XI);        // chain B from dy_{s-1} = xi
}

// mid_phase (ipm.hpp) of the segmented solve: per knot the feed-forward d0 = S^-1 lu and theta_k = Pi_k' c_k - D_k' lu_k with the
// records of the knot's OWN chain; theta summed per chain; the coarse stage (head of this file) for mu_g, xi, dlam; then
// d_k = d0 + D_k mult_k and ct_k = c_k - Gam_k d_k with mult = dlam for the knots of chain A, mu_g for those of chain B.
template <int MODEL, class BLK> GD void mid_phase_seg(BLK& K, int k, bool act, double hdt, int s, double* mugn, Prof* pf = nullptr) {
#define MT_(i) do { if (pf) pf->tick(i); } while (0)
    MT_(PF_MID);
    using T = MT<MODEL>;
    using R = Rec<MODEL>;
    using S = SegC<MODEL>;
    constexpr int n = T::n, m = T::m;
    static_assert(T::PG2 && T::LTI && BLK::C::KD_LDS && BLK::ONE, "the one-wave double-integrator kernel");
    const int N = K.N;
    const LPtr<double> L = K.lds;
    double th[n], d0[m];
#pragma unroll
    for (int i = 0; i < n; i++) th[i] = 0;
#pragma unroll
    for (int i = 0; i < m; i++) d0[i] = 0;
    if (act) {
        double tt[n], lu[m], Gamk[n * m], Mg[n * n];
        double gterm[n], gsub[n];
        if (k >= 1) {
            load_M_Gam(K, k, Mg, Gamk);
        } else {
            Dyn<MODEL>::B(K.P.mp, Gamk);
#pragma unroll
            for (int i = 0; i < n * m; i++) Gamk[i] *= hdt;
        }
#pragma unroll
        for (int i = 0; i < n; i++) tt[i] = K.pv[k * n + i];   // pt_k = p_k + r_k
#pragma unroll
        for (int i = 0; i < m; i++) {
            double sacc = K.qu_(k, i);
#pragma unroll
            for (int l = 0; l < n; l++) if (T::Gnz(l, i)) sacc += Gamk[l * m + i] * tt[l];
            lu[i] = sacc;
        }
#pragma unroll
        for (int i = 0; i < m; i++) {
            double sacc = 0;
#pragma unroll
            for (int l = 0; l < m; l++) sacc += K.kdS(k, i, l) * lu[l];
            d0[i] = sacc;
        }
        {   // the last knot's goal term, evaluated by every lane on its own knot's data and selected below (mid_phase)
            double rdl[n], Gg[n * m];
            load_M_Gam(K, k, Mg, Gg);
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
        double thd[n];
#pragma unroll
        for (int j = 0; j < n; j++) thd[j] = K.nun[k * n + j];   // Pi_k^T c_k from the factor sweep
#pragma unroll
        for (int i = 0; i < m; i++)
#pragma unroll
            for (int j = 0; j < n; j++) thd[j] -= K.kd(k, R::oD + i * n + j) * lu[i];
#pragma unroll
        for (int j = 0; j < n; j++) th[j] = (k == N - 1 && K.is_goal(j)) ? (thd[j] + gterm[j]) - gsub[j] : thd[j];
    }
    MT_(PF_M_TH);
    // theta of chain A (= a, the end state it reaches for lam0) and of chain B (with the goal terms), summed at once
    double r2[2 * n];
#pragma unroll
    for (int j = 0; j < n; j++) { r2[j] = (k < s) ? th[j] : 0.0; r2[n + j] = (k < s) ? 0.0 : th[j]; }
    wave_reduce_n<2 * n>(r2, OpSum());
    MT_(PF_M_RED);
    // the coarse stage: lane i < n forms row i (three levels of matrix-vector products, the vectors travel by v_readlane)
    double muv[n], lamv[n];
    {
        const int ri = K.tid < n ? K.tid : 0;
        double ph[n];
#pragma unroll
        for (int l = 0; l < n; l++) ph[l] = L[S::PBV + l] - K.nu[s * n + l];   // p_B - lam0
        double v1 = 0, v2 = 0, v3 = 0;
#pragma unroll
        for (int l = 0; l < n; l++) {
            const double tt_ = L[S::Tt + ri * n + l];
            v1 += tt_ * r2[l];
            v2 += L[S::Sg + ri * n + l] * ph[l];
            v3 += L[S::Pa + ri * n + l] * r2[l] + L[S::Tt + l * n + ri] * ph[l];
        }
        const double w1 = v1 - v2;
        double w1v[n];
#pragma unroll
        for (int l = 0; l < n; l++) w1v[l] = readlane_f64(w1, l);
        double mu = 0;
#pragma unroll
        for (int l = 0; l < n; l++) mu += L[S::Gci + ri * n + l] * r2[n + l] + L[S::A1 + ri * n + l] * w1v[l];
        mu = K.is_goal(ri) ? mu : 0.0;
#pragma unroll
        for (int l = 0; l < n; l++) muv[l] = readlane_f64(mu, l);
        double xi = w1, dl = v3;
#pragma unroll
        for (int l = 0; l < n; l++) { xi -= L[S::A2 + ri * n + l] * muv[l]; dl += L[S::A3 + ri * n + l] * muv[l]; }
#pragma unroll
        for (int l = 0; l < n; l++) lamv[l] = readlane_f64(dl, l);
        if (K.tid < n) { mugn[K.tid] = mu; L[S::XI + K.tid] = xi; L[S::LAM + K.tid] = dl; }
    }
    MT_(PF_M_MU);
    if (act) {  // d_k = d0 + D_k mult ; ct_k = c_k - Gam_k d_k
        double dk[m];
#pragma unroll
        for (int i = 0; i < m; i++) {
            double sacc = d0[i];
#pragma unroll
            for (int j = 0; j < n; j++) sacc += K.kd(k, R::oD + i * n + j) * ((k < s) ? lamv[j] : muv[j]);
            dk[i] = sacc;
            K.dv_(k, i) = sacc;
        }
        double Gamk[n * m];
        if (k >= 1) {
            double Mk_[n * n];
            load_M_Gam(K, k, Mk_, Gamk);
        } else {
            Dyn<MODEL>::B(K.P.mp, Gamk);
#pragma unroll
            for (int i = 0; i < n * m; i++) Gamk[i] *= hdt;
        }
#pragma unroll
        for (int i = 0; i < n; i++) {
            double sacc = K.cv[k * n + i];
#pragma unroll
            for (int l = 0; l < m; l++) if (T::Gnz(i, l)) sacc -= Gamk[i * m + l] * dk[l];
            K.dY[k * n + i] = sacc;
        }
    }
    MT_(PF_M_DK);
    K.sync();
    MT_(PF_M_SYNC);
#undef MT_
}

}  // namespace gusto
