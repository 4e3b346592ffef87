// segw.hpp -- a WAVE PER RICCATI CHAIN for the matrix-core kernels (astrobeeSE3, astrobeeSE3manifold; round 6).
//
// A batch smaller than the GPU leaves SIMDs without a wave, and a one-wave problem spends ~55 % of a KKT solve in N-stage dependent
// chains (the factor sweep, the vector sweeps).  scp_kernel_w2 gives a problem NCH = 2 or 4 waves and splits the horizon into NCH
// chains, chain c = the stages seg_lo(c) .. seg_lo(c + 1) - 1:
//   * the last chain is the recursion as it was (P = Pi = 0 behind the last knot, the goal rows' E term there);
//   * a chain in front of an interface runs the SAME recursion started from P = 0 with its end state adjoined as a terminal
//     equality, multiplier lam = the costate behind the interface -- the machinery the goal rows already use: Pi starts as I, and the
//     chain's "Gd" is its compliance d(end state) / d lam.
// A chain then is the affine map (y, lam) -> (front costate, end state) = (P y + Pi lam + p, Pi' y - Gd lam + th), and two neighbours
// MERGE into one chain of the same form (seg.hpp's coarse stage; cf. the associative LQR elements of parallel-in-time Riccati
// solvers).  With A in front of B, Ta = (I + P_B Gd_A)^-1, Sig = Gd_A Ta (symmetric), Pa = Ta P_B:
//       P = P_A + Pi_A Pa Pi_A',   Pi = Pi_A Ta Pi_B,   Gd = Gd_B + Pi_B' Sig Pi_B,
//       p = p_A + Pi_A (Pa th_A + Ta ph),   th = th_B + Pi_B' (Ta' th_A - Sig ph),          ph = p_B - lam0
//       interface:  xi = Ta' (Pi_A' y + th_A) - Sig (ph + Pi_B lam'),   dlam = Pa (Pi_A' y + th_A) + Ta (ph + Pi_B lam')
// -- products only, no inverse of a compliance (a chain may have an uncontrollable direction), and no output is a difference of large
// terms.  lam0 = the CURRENT costate iterate at the interface: the chain in front starts its backward vector sweep from it, so every
// coarse quantity vanishes with the Newton step.  Four chains are merged as a tree, (C0 | C1) and (C2 | C3) side by side and then
// the two pairs; the interfaces are resolved from the middle one outwards.  Prototype (its chains folded from the back one by one), run against the oracle's sequential recursion with 2 and 4 chains: tools/proto/segriccati.c,
// profiles/r06_segmented_riccati_proto.txt.  Reference path: the convex subproblem of scp_gusto.jl:104,178-314 (JuMP.optimize!).
//
// Who runs what (wave 0 = MAIN: the one-wave program, and the last chain; wave h >= 1 = helper, chain h - 1):
//   FACTOR   every wave its chain's factor sweep; join; helper 1 merges the chains' matrices (seg_merge) while the main wave builds
//            the predictor's right-hand side (four chains: helpers 1 and 3 the two pairs, a barrier, helper 1 the pairs); join
//   BACK     every wave its chain's backward vector sweep; join.  The main wave's mid phase then folds the vectors, gets mu_g and
//            every interface's (xi, dlam)
//   FWD      every wave its chain's forward sweep; join
//   ROWS_R   the residual pass: the helpers share each knot's obstacle rows (15 - 25 of a knot's ~30 rows in the ISS corner), the main
//            wave has the other rows and the stage cost; join; the main wave adds the helpers' partial sums (segw_rows_*)
//   STEP     the step pass of the predictor, likewise
//   STEP_CS  the step pass that ends with new costates: helper 1 computes them for all knots from the P | Pi records (record
//            seg_lo(c) - 1 = (0 | I): nu behind an interface = lam0 + dlam falls out of the same formula) with the first knot's
//            closing equation, the other helpers share the obstacle rows (two waves: helper and main wave take half of them each); join
// A command is a word in LDS and two workgroup barriers (post: the main wave has drained what the helpers read; join: everybody has
// drained its stores); the phases themselves stay barrier-free one-wave code on disjoint knots and LDS.
#pragma once

namespace gusto {

// ---- n x n products on the matrix cores: one 16 x 16 tile, K = 16 as four v_mfma_f64_16x16x4 (operand and accumulator layouts as in
// factor_sweep_mfma: lane (mi, mq), register q <-> A[mi][mq + 4 q], B[mq + 4 q][mi], C[mq + 4 q][mi]; entries beyond n are zeros)
template <int n> struct MMTile {
    LPtr<double> L;
    int mi, mq;
    GD MMTile(double* lds, int tid) : L(lds), mi(tid & 15), mq(tid >> 4) {}
    struct Op { double v[4]; };
    GD Op A(int o, bool tr = false) const {   // operand A = the matrix at offset o (row-major), or its transpose
        Op r;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = mq + 4 * q;
            const bool ok = mi < n && k < n;
            const double v = L[o + (ok ? (tr ? k * n + mi : mi * n + k) : 0)];
            r.v[q] = ok ? v : 0.0;
        }
        return r;
    }
    GD Op B(int o, bool tr = false) const {
        Op r;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int k = mq + 4 * q;
            const bool ok = mi < n && k < n;
            const double v = L[o + (ok ? (tr ? mi * n + k : k * n + mi) : 0)];
            r.v[q] = ok ? v : 0.0;
        }
        return r;
    }
    GD v4d C(int o) const {
        v4d c;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = mq + 4 * q;
            const bool ok = mi < n && row < n;
            const double v = L[o + (ok ? row * n + mi : 0)];
            c[q] = ok ? v : 0.0;
        }
        return c;
    }
    GD v4d eye() const {
        v4d c;
#pragma unroll
        for (int q = 0; q < 4; q++) c[q] = (mq + 4 * q == mi && mi < n) ? 1.0 : 0.0;
        return c;
    }
    GD void put(int o, v4d c, bool tr = false) const {
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int row = mq + 4 * q;
            if (mi < n && row < n) L[o + (tr ? mi * n + row : row * n + mi)] = c[q];
        }
    }
    GD static v4d mm(const Op& a, const Op& b, v4d acc) {
#pragma unroll
        for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a.v[q], b.v[q], acc, 0, 0, 0);
        return acc;
    }
};

// One merge of the matrices (one wave): the chain(s) in front of an interface, compliance Gd_A at LDS offset `Gdf`, with what lies
// behind it, (P, Pi) at `Pc`, `PIc` and Gd at `GdB`.  Leaves the interface's Ta', Sig, Pa, A2, A3 in the block `I` and the merged
// Gd = Gd_B + Pi_B' Sig Pi_B IN PLACE of Gd_B; X1 is n x n scratch.  Returns false on a zero / non-finite pivot.
template <int MODEL, int NCH, class BLK> GD bool seg_merge(BLK& K, int I, int Gdf, int Pc, int PIc, int GdB, int X1) {
    using SB = SegB<MODEL, NCH>;
    constexpr int n = SB::n, NN = n * n, RN = (NN + 63) / 64;
    static_assert(n <= 16, "one MFMA tile");
    const LPtr<double> L = K.lds;
    const int tid = K.tid;
    MMTile<n> T(K.lds, tid);
    const v4d Z = {0, 0, 0, 0};
    int ei[RN], ej[RN];
    bool on[RN];
#pragma unroll
    for (int r = 0; r < RN; r++) { const int e = tid + 64 * r; on[r] = e < NN; ei[r] = on[r] ? e / n : 0; ej[r] = on[r] ? e % n : 0; }
    bool ok = true;
    // X = I + P_B Gd_A -> X1, inverted in place by Gauss-Jordan without pivoting (X = I + (PSD)(PSD): eigenvalues >= 1)
    T.put(X1, T.mm(T.A(Pc), T.B(Gdf), T.eye()));
    K.sync();
    for (int c = 0; c < n; c++) {
        const double piv = L[X1 + c * n + c];
        if (!(fabs(piv) > 0.0) || !isfinite(piv)) ok = false;
        const double d = rcp_nr(piv);
        double wij[RN], wcj[RN], wic[RN], w[RN];
#pragma unroll
        for (int r = 0; r < RN; r++) { wij[r] = L[X1 + ei[r] * n + ej[r]]; wcj[r] = L[X1 + c * n + ej[r]]; wic[r] = L[X1 + ei[r] * n + c]; }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int r = 0; r < RN; r++) {
            const int i = ei[r], jj = ej[r];
            const double rowc = (jj == c) ? d : wcj[r] * d;
            const double other = (jj == c) ? -(wic[r] * d) : wij[r] - wic[r] * (wcj[r] * d);
            w[r] = (i == c) ? rowc : other;
        }
        K.sync();
#pragma unroll
        for (int r = 0; r < RN; r++) if (on[r]) L[X1 + ei[r] * n + ej[r]] = w[r];
        K.sync();
    }
    // X1 = Ta:  Tt = Ta', Sig = Gd_A Ta (both triangles from one mean: the transpose comes from the transposed product, the same
    // terms in the same order), Pa = Ta P_B, A3 = Ta Pi_B
    {
        const auto aG = T.A(Gdf), bX = T.B(X1), aXt = T.A(X1, true), bGt = T.B(Gdf, true), aX = T.A(X1), bP = T.B(Pc), bPi = T.B(PIc);
        const v4d ta = T.C(X1);
        const v4d s1 = T.mm(aG, bX, Z), s2 = T.mm(aXt, bGt, Z), pa = T.mm(aX, bP, Z), a3 = T.mm(aX, bPi, Z);
        v4d sg;
#pragma unroll
        for (int q = 0; q < 4; q++) sg[q] = 0.5 * (s1[q] + s2[q]);
        T.put(I + SB::Tt, ta, true); T.put(I + SB::Sg, sg); T.put(I + SB::Pa, pa); T.put(I + SB::A3, a3);
    }
    K.sync();
    T.put(I + SB::A2, T.mm(T.A(I + SB::Sg), T.B(PIc), Z));                 // A2 = Sig Pi_B
    K.sync();
    T.put(GdB, T.mm(T.A(PIc, true), T.B(I + SB::A2), T.C(GdB)));           // Gd_B += Pi_B' A2
    K.sync();
    return ok;
}
// ... and the merged chain's P = P_A + Pi_A Pa Pi_A', Pi = Pi_A Ta Pi_B (the interface block `I` holds Pa, A3 = Ta Pi_B): -> oP, oPi
template <int MODEL, int NCH, class BLK> GD void seg_merge_front(BLK& K, int I, int Pj, int PIj, int oP, int oPi, int X2) {
    using SB = SegB<MODEL, NCH>;
    MMTile<SB::n> T(K.lds, K.tid);
    const v4d Z = {0, 0, 0, 0};
    const auto aPi = T.A(PIj);
    T.put(X2, T.mm(aPi, T.B(I + SB::Pa), Z));                              // Pi_A Pa
    T.put(oPi, T.mm(aPi, T.B(I + SB::A3), Z));
    K.sync();
    T.put(oP, T.mm(T.A(X2), T.B(PIj, true), T.C(Pj)));
    K.sync();
}
// the goal Hessian of the whole horizon is in sGd: its inverse (inv_spd_block: sGd -> sP) -> Gci, and A1 = Gci Pi' with the Pi behind
// the interface that carries mu_g
template <int MODEL, int NCH, class BLK> GD void seg_fold_finish(BLK& K, double* fail, int PIc) {
    using SB = SegB<MODEL, NCH>;
    const int sb = K.P.ll.seg, oSP = LdsC<MODEL, true>::sP;
    MMTile<SB::n> T(K.lds, K.tid);
    const v4d Z = {0, 0, 0, 0};
    inv_spd_block<MODEL>(K, fail);
    K.sync();
    const v4d gi = T.C(oSP);
    T.put(sb + SB::Gci, gi);
    T.put(sb + SB::A1, T.mm(T.A(oSP), T.B(PIc, true), Z));
    K.sync();
}
// two chains: the one merge (helper 1)
template <int MODEL, class BLK> GD void seg_fold_factor2(BLK& K, double* fail) {
    using SB = SegB<MODEL, 2>;
    const int sb = K.P.ll.seg, I = sb + SB::IF(0);
    if (!seg_merge<MODEL, 2>(K, I, sb + SB::CH(0) + SB::Gdf, I + SB::Pc, I + SB::PIc, LdsC<MODEL, true>::sGd, sb + SB::X1)) *fail = 1.0;
    seg_fold_finish<MODEL, 2>(K, fail, I + SB::PIc);
}
// four chains, merged as a TREE: (C0 | C1) on helper 1 and (C2 | C3) on helper 3 side by side (level 1), a workgroup barrier, then
// (C0 C1 | C2 C3) on helper 1 (level 2) -- the multiplier behind interfaces 1 and 2 is mu_g, the one behind interface 0 is interface 1's.
template <int MODEL, class BLK> GD void seg_fold_tree_level1(BLK& K, double* fail, int h) {
    using SB = SegB<MODEL, 4>;
    const int sb = K.P.ll.seg;
    if (h == 3) {          // chain 2's wave: (C2 | C3) and the pair's (P, Pi), which interface 1 sees behind it
        const int I = sb + SB::IF(2), In = sb + SB::IF(1), scr = sb + SB::sPG2(2);   // (its own block is idle: scratch)
        if (!seg_merge<MODEL, 4>(K, I, sb + SB::CH(2) + SB::Gdf, I + SB::Pc, I + SB::PIc, LdsC<MODEL, true>::sGd, scr)) *fail = 1.0;
        seg_merge_front<MODEL, 4>(K, I, sb + SB::CH(2) + SB::Pf, sb + SB::CH(2) + SB::Pif, In + SB::Pc, In + SB::PIc, scr + SB::NNp);
    } else if (h == 1) {   // chain 0's wave: (C0 | C1); chain 1's Gd becomes the pair's
        const int I0 = sb + SB::IF(0), C1 = sb + SB::CH(1);
        if (!seg_merge<MODEL, 4>(K, I0, sb + SB::CH(0) + SB::Gdf, C1 + SB::Pf, C1 + SB::Pif, C1 + SB::Gdf, sb + SB::X1)) *fail = 1.0;
    }
}
template <int MODEL, class BLK> GD void seg_fold_tree_level2(BLK& K, double* fail) {      // helper 1
    using SB = SegB<MODEL, 4>;
    const int sb = K.P.ll.seg, I1 = sb + SB::IF(1);
    if (!seg_merge<MODEL, 4>(K, I1, sb + SB::CH(1) + SB::Gdf, I1 + SB::Pc, I1 + SB::PIc, LdsC<MODEL, true>::sGd, sb + SB::X1)) *fail = 1.0;
    seg_fold_finish<MODEL, 4>(K, fail, I1 + SB::PIc);
}

// ---- the ring-buffered one-wave vector sweeps of ipm.hpp over a RANGE of knots (operands from the global Phicl records) ----
template <int MODEL, class BLK> GD void backward_sweep_ring_rng(const BLK& K, int khi, int klo, int start, int last_out) {
    constexpr int n = BLK::n, C = 64 / n, PS = 4, RING = GUSTO_SWEEP_RING;
    const LPtr<double> L = K.lds;
    const int tid = K.tid;
    const int g = (tid < C * n) ? tid / n : C - 1, i = (tid < C * n) ? tid % n : 0;
    const int pvo = LdsC<MODEL, true>::vecs + 2 * K.N * n;
    auto fetch = [&](int k0, double* c, double& q) {
        const int kk = (k0 - g >= klo) ? k0 - g : klo;
#pragma unroll
        for (int l = 0; l < n; l++) c[l] = K.Phicl[(size_t)kk * BLK::SPH + l * n + i];
        q = K.pv[kk * n + i];
    };
    double cb[RING][n], qb[RING];
#pragma unroll
    for (int d = 0; d < RING - 1; d++) fetch(khi - d * C, cb[d], qb[d]);
    double pval = L[start + i];
    K.sync();
    if (tid < n) K.pv[khi * n + tid] = pval;
    for (int kb = khi; kb >= klo; kb -= RING * C) {
        static_for<0, RING>([&](auto DD) {
            constexpr int d = decltype(DD)::value;
            const int k0 = kb - d * C;
            fetch(k0 - (RING - 1) * C, cb[(d + RING - 1) % RING], qb[(d + RING - 1) % RING]);
            if (k0 >= klo) {
#pragma unroll
                for (int gs = 0; gs < C; gs++) {
                    if (k0 - gs >= klo) {
                        const int sg = (gs == 0) ? C - 1 : gs - 1;
                        double acc[PS];
#pragma unroll
                        for (int q = 0; q < PS; q++) acc[q] = (q == 0) ? qb[d] : 0.0;
                        double pb[n];
#pragma unroll
                        for (int l = 0; l < n; l++) pb[l] = readlane_f64(pval, sg * n + l);
                        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                        for (int l = 0; l < n; l++) acc[l % PS] += cb[d][l] * pb[l];
                        const double sacc = (acc[0] + acc[1]) + (acc[2] + acc[PS - 1]);
                        pval = (g == gs) ? sacc : pval;
                    }
                }
                const int kk = k0 - g;
                if (tid < C * n && kk >= klo) {
                    const int dst = (kk == klo && last_out >= 0) ? last_out + i : pvo + (kk - 1) * n + i;
                    L[dst] = pval;
                }
            }
        });
    }
    K.sync();
}
template <int MODEL, class BLK> GD void forward_sweep_ring_rng(const BLK& K, int klo, int khi, int start) {
    constexpr int n = BLK::n, C = 64 / n, PS = 4, RING = GUSTO_SWEEP_RING;
    const LPtr<double> L = K.lds;
    const int tid = K.tid;
    const int g = (tid < C * n) ? tid / n : C - 1, i = (tid < C * n) ? tid % n : 0;
    auto fetch = [&](int k0, double* r, double& c) {
        const int kk = (k0 + g <= khi) ? k0 + g : khi;
#pragma unroll
        for (int l = 0; l < n; l++) r[l] = K.Phicl[(size_t)kk * BLK::SPH + i * n + l];
        c = K.dY[kk * n + i];
    };
    double rb[RING][n], qb[RING];
#pragma unroll
    for (int d = 0; d < RING - 1; d++) fetch(klo + d * C, rb[d], qb[d]);
    double yval = (start >= 0) ? L[start + i] : 0.0;
    K.sync();
    for (int kb = klo; kb <= khi; kb += RING * C) {
        static_for<0, RING>([&](auto DD) {
            constexpr int d = decltype(DD)::value;
            const int k0 = kb + d * C;
            fetch(k0 + (RING - 1) * C, rb[(d + RING - 1) % RING], qb[(d + RING - 1) % RING]);
            if (k0 <= khi) {
#pragma unroll
                for (int gs = 0; gs < C; gs++) {
                    if (k0 + gs <= khi) {
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
                        const double sacc = (acc[0] + acc[1]) + (acc[2] + acc[PS - 1]);
                        yval = (g == gs) ? sacc : yval;
                    }
                }
                const int kk = k0 + g;
                if (tid < C * n && kk <= khi) K.dY[kk * n + i] = yval;
            }
        });
    }
    K.sync();
}

// costate_pass_1w of the segmented solve: nu_{k+1} = P_k dy_k + p_k + Pi_k mult_k
template <int MODEL, int NCH> GD void costate_pass_seg(SweepView<MODEL> K) {
    using T = MT<MODEL>;
    using R = Rec<MODEL>;
    constexpr int n = T::n, C = 64 / n;
    const LPtr<double> L = K.lds;
    const int tid = K.tid, N = K.N;
    const int g = (tid < C * n) ? tid / n : C - 1, i = (tid < C * n) ? tid % n : 0;
    constexpr int RING = GUSTO_SWEEP_RING;
    double pr[RING][n], pi[RING][n];
    auto fetch = [&](int k0, double* a, double* b) {
        const int k = (k0 + g + 1 < N) ? k0 + g : N - 2;
        const double* pa = K.Paft + (size_t)k * R::SNN + i;     // (records stored transposed by factor_sweep_mfma)
        const double* pb = K.Piaft + (size_t)k * R::SNN + i;
#pragma unroll
        for (int l = 0; l < n; l++) { a[l] = pa[l * n]; b[l] = pb[l * n]; }
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
                const int mo = seg_mult_off<MODEL, NCH>(k, N, K.seg_off);
                double dy[n], ml[n];
#pragma unroll
                for (int l = 0; l < n; l++) { dy[l] = K.dY[k * n + l]; ml[l] = L[mo + l]; }
                double sacc = K.pv[k * n + i] - K.rv[k * n + i];
#pragma unroll
                for (int l = 0; l < n; l++) sacc += pr[d][l] * dy[l] + pi[d][l] * ml[l];
                if (ok) K.nun[(k + 1) * n + i] = sacc;
            }
        });
    }
    K.sync();
}

// ---- commands ---------------------------------------------------------------------------------------------------------------
// (the command words: common.hpp, SEGW_*)
GD void segw_barrier() { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory"); }
GD void segw_join() { segw_barrier(); }
// mailbox: [0] the command, [2..6] what a helper needs to rebuild the problem's view (written once per interior point solve),
// [8..13] the scalars of a row pass: alpha_prev | pass, kappa, omega, Delta, mu_t, tau
template <int MODEL, int NCH, class BLK> GD void segw_open(BLK& K) {
    const LPtr<double> L = K.lds;
    const int mb = K.P.ll.seg + SegB<MODEL, NCH>::MBX;
    if (K.tid == 0) {
        const typename BLK::Args a = K.args();
        L[mb + 2] = (double)a.b; L[mb + 3] = (double)a.slot; L[mb + 4] = (double)a.goalmask; L[mb + 5] = (double)a.boxmask; L[mb + 6] = a.dt;
    }
}
template <int MODEL, int NCH, class BLK> GD void segw_post(BLK& K, int cmd) {
    const LPtr<double> L = K.lds;
    if (K.tid == 0) L[K.P.ll.seg + SegB<MODEL, NCH>::MBX] = (double)cmd;
    segw_barrier();
}
template <int MODEL, int NCH, class BLK>
GD void segw_post_rows(BLK& K, int cmd, double a0, double kappa, double omega, double Delta, double mu_t, double tau) {
    const LPtr<double> L = K.lds;
    const int mb = K.P.ll.seg + SegB<MODEL, NCH>::MBX;
    if (K.tid == 0) {
        L[mb] = (double)cmd;
        L[mb + 8] = a0; L[mb + 9] = kappa; L[mb + 10] = omega; L[mb + 11] = Delta; L[mb + 12] = mu_t; L[mb + 13] = tau;
    }
    segw_barrier();
}
// kernel exit of the main wave: release the helpers for good (no join: a wave that has ended is not waited for)
GD void segw_exit(double* lds, int mbx) {
    if ((threadIdx.x & 63) == 0) lds[mbx] = (double)SEGW_EXIT;
    segw_barrier();
}

// ---- a knot's obstacle rows shared between the waves ------------------------------------------------------------------------------
// An obstacle row touches the position coordinates only (window 0 .. WS - 1): what a wave's share contributes to the knot's sums is
// SEG_RP numbers, left in the wave's LDS block as [value][lane] and added by the main wave in the order of the waves.
//   residual pass: comp, max |r_p|, r_dx[WS], gx0[WS], the upper triangle of H_x on the window
//   step pass:     the step-length fraction (an, ad), c0 c1 c2, gA_x[WS], gB_x[WS]
template <int MODEL, int NCH> GD void segw_rows_resid_helper(Blk<MODEL, true>& B, int hi, int rank, int nshare) {
    using T = MT<MODEL>;
    using SB = SegB<MODEL, NCH>;
    constexpr int n = T::n, m = T::m, WS = T::WS, NHX = n * (n + 1) / 2;
    static_assert(2 + 2 * WS + WS * (WS + 1) / 2 <= SEG_RP && 5 + 2 * WS <= SEG_RP, "row partials");
    const LPtr<double> L = B.lds;
    const int sb = B.P.ll.seg, mb = sb + SB::MBX, k = B.tid;
    const bool act = k < B.N;
    const double alpha_prev = L[mb + 8], kappa = L[mb + 9], omega = L[mb + 10], Delta = L[mb + 11];
    RowCtx<MODEL> ctx;
    RowState rs;
    make_row_ctx<MODEL>(B, k, act, kappa, omega, Delta, ctx, rs);
    double xs[n], Hx[NHX], Hu[1] = {0}, rdx[n], rdu[1] = {0}, gx0[n], gu0[1] = {0};
#pragma unroll
    for (int i = 0; i < n; i++) { xs[i] = act ? B.Xw[k * n + i] : 0.0; rdx[i] = 0; gx0[i] = 0; }
#pragma unroll
    for (int i = 0; i < NHX; i++) Hx[i] = 0;
    RowPre<0> pre;
    OpResidHess<n, m, 0, false> op{rs, Hx, Hu, rdx, rdu, gx0, gu0, alpha_prev, &pre};
    visit_obs_rows<MODEL>(ctx, xs, op, ctx.mask & seg_obs_share(rank, nshare));
    const int o = sb + SB::sPG2(hi) + k;
    L[o] = op.comp; L[o + 64] = op.maxrp;
    int v = 2;
#pragma unroll
    for (int a = 0; a < WS; a++) { L[o + 64 * v] = rdx[a]; L[o + 64 * (v + WS)] = gx0[a]; v++; }
    v += WS;
#pragma unroll
    for (int a = 0; a < WS; a++)
#pragma unroll
        for (int b = a; b < WS; b++) { L[o + 64 * v] = Hx[sidx(a, b, n)]; v++; }
}
template <int MODEL, int NCH, class Op> GD void segw_rows_resid_add(const LPtr<double> L, int sb, int k, int nhelp, Op& op, double* Hx, double* rdx, double* gx0) {
    using T = MT<MODEL>;
    using SB = SegB<MODEL, NCH>;
    constexpr int n = T::n, WS = T::WS;
#pragma unroll
    for (int hi = 0; hi < NCH - 1; hi++) {
        if (hi < nhelp) {
            const int o = sb + SB::sPG2(hi) + k;
            op.comp += L[o]; op.maxrp = nanmax(op.maxrp, L[o + 64]);
            int v = 2;
#pragma unroll
            for (int a = 0; a < WS; a++) { rdx[a] += L[o + 64 * v]; gx0[a] += L[o + 64 * (v + WS)]; v++; }
            v += WS;
#pragma unroll
            for (int a = 0; a < WS; a++)
#pragma unroll
                for (int b = a; b < WS; b++) { Hx[sidx(a, b, n)] += L[o + 64 * v]; v++; }
        }
    }
}
// (cs: helper 1 is busy with the costates -- the sharers are the main wave and the helpers behind it)
template <int MODEL, int NCH> GD void segw_rows_step_helper(Blk<MODEL, true>& B, int hi, int rank, int nshare) {
    using T = MT<MODEL>;
    using SB = SegB<MODEL, NCH>;
    constexpr int n = T::n, WS = T::WS;
    const LPtr<double> L = B.lds;
    const int sb = B.P.ll.seg, mb = sb + SB::MBX, k = B.tid;
    const bool act = k < B.N;
    const int pass = (int)L[mb + 8];
    const double kappa = L[mb + 9], omega = L[mb + 10], Delta = L[mb + 11], mu_t = L[mb + 12], tau = L[mb + 13];
    RowCtx<MODEL> ctx;
    RowState rs;
    make_row_ctx<MODEL>(B, k, act, kappa, omega, Delta, ctx, rs);
    double xs[n], dxs[n], dus[1] = {0}, gAx[n], gBx[n], gAu[1] = {0}, gBu[1] = {0};
#pragma unroll
    for (int i = 0; i < n; i++) { xs[i] = act ? B.Xw[k * n + i] : 0.0; gAx[i] = 0; gBx[i] = 0; dxs[i] = 0; }
#pragma unroll
    for (int i = 0; i < WS; i++) dxs[i] = act ? B.dXs[k * n + i] : 0.0;   // (the main wave's step phase has stored the primal step)
    RowPre<0> pre;
    OpStep<0, RowState, false> op{rs, dxs, dus, pass, mu_t, tau, gAx, gAu, gBx, gBu, &pre, nullptr};
    visit_obs_rows<MODEL>(ctx, xs, op, ctx.mask & seg_obs_share(rank, nshare));
    const int o = sb + SB::sPG2(hi) + k;
    L[o] = op.amax.an; L[o + 64] = op.amax.ad; L[o + 128] = op.c0; L[o + 192] = op.c1; L[o + 256] = op.c2;
#pragma unroll
    for (int a = 0; a < WS; a++) { L[o + 64 * (5 + a)] = gAx[a]; L[o + 64 * (5 + WS + a)] = gBx[a]; }
}
template <int MODEL, int NCH, class Op> GD void segw_rows_step_add(const LPtr<double> L, int sb, int k, int h0, Op& op, double* gAx, double* gBx) {
    using T = MT<MODEL>;
    using SB = SegB<MODEL, NCH>;
    constexpr int WS = T::WS;
#pragma unroll
    for (int hi = 0; hi < NCH - 1; hi++) {
        if (hi >= h0) {
            const int o = sb + SB::sPG2(hi) + k;
            const double an = L[o], ad = L[o + 64];
            const bool take = an * op.amax.ad < op.amax.an * ad;   // (StepFrac::test: both denominators are positive)
            op.amax.an = take ? an : op.amax.an; op.amax.ad = take ? ad : op.amax.ad;
            op.c0 += L[o + 128]; op.c1 += L[o + 192]; op.c2 += L[o + 256];
#pragma unroll
            for (int a = 0; a < WS; a++) { gAx[a] += L[o + 64 * (5 + a)]; gBx[a] += L[o + 64 * (5 + WS + a)]; }
        }
    }
}

// one chain's share of the three sequential phases (c = NCH - 1: the main wave's, as called phases below)
template <int MODEL, int NCH> GD void seg_chain_factor(SweepView<MODEL> K, int c, double* fail, Prof& pf) {
    using SB = SegB<MODEL, NCH>;
    const int N = K.N, sb = K.seg_off;
    const bool last = c == NCH - 1;
    const int oP = last ? sb + SB::IF(NCH - 2) + SB::Pc : sb + SB::CH(c < NCH - 1 ? c : 0) + SB::Pf;
    const int oPi = last ? sb + SB::IF(NCH - 2) + SB::PIc : sb + SB::CH(c < NCH - 1 ? c : 0) + SB::Pif;
    const int oGd = last ? -1 : sb + SB::CH(c < NCH - 1 ? c : 0) + SB::Gdf;
    factor_sweep_mfma<MODEL, false, true>(K, fail, pf, seg_lo(c + 1, N, NCH) - 1, seg_lo(c, N, NCH), !last, oP, oPi, oGd);
}
template <int MODEL, int NCH, class BLK> GD void seg_chain_backward(const BLK& B, int c) {
    using SB = SegB<MODEL, NCH>;
    using C = LdsC<MODEL, true>;
    constexpr int n = MT<MODEL>::n;
    const int N = B.N, lo = seg_lo(c, N, NCH), hi = seg_lo(c + 1, N, NCH) - 1;
    // the last chain starts from r_{N-1}, a chain in front of an interface from the costate iterate there (lam0 = nu behind its last
    // knot); the first chain ends at knot 1, the others leave the costate offset in front of their first stage in SegB::PBV
    const int start = (c == NCH - 1) ? C::vecs + 4 * N * n + (N - 1) * n : C::vecs + 5 * N * n + (hi + 1) * n;
    backward_sweep_ring_rng<MODEL>(B, hi, c == 0 ? 1 : lo, start, c == 0 ? -1 : B.P.ll.seg + SB::PBV(c > 0 ? c : 1));
}
template <int MODEL, int NCH, class BLK> GD void seg_chain_forward(const BLK& B, int c) {
    using SB = SegB<MODEL, NCH>;
    const int N = B.N, lo = seg_lo(c, N, NCH), hi = seg_lo(c + 1, N, NCH) - 1;
    // (a chain in front of an interface stops one knot early: its end state is the interface's xi by construction, written by the main wave)
    forward_sweep_ring_rng<MODEL>(B, lo, c == NCH - 1 ? hi : hi - 1, c == 0 ? -1 : B.P.ll.seg + SB::XI(c > 0 ? c - 1 : 0));
}

// The helper wave h = 1 .. NCH - 1 (chain h - 1).  What it runs are real calls, like the main wave's phases: inlined into the kernel
// they shared its register allocation (1300 spilled SGPRs, 320 VGPRs) and every piece added there slowed the sweeps of all helpers.
// (the chain is a template parameter, like the main wave's: one called sweep per chain)
template <int MODEL, int NCH, int CH> __device__ __noinline__ void segw_h_factor_chain(typename Blk<MODEL, true>::Args a) {
    using SB = SegB<MODEL, NCH>;
    Blk<MODEL, true> B(a, gusto_dyn_lds);
    SweepView<MODEL> K = SweepView<MODEL>::make(B);
    const int sb = B.P.ll.seg;
    K.sPG = gusto_dyn_lds + sb + SB::sPG2(CH); K.sHh = gusto_dyn_lds + sb + SB::Lw2(CH);   // its own operand buffers
    Prof pfd;
    seg_chain_factor<MODEL, NCH>(K, CH, gusto_dyn_lds + LdsC<MODEL, true>::misc + 8, pfd);
}
template <int MODEL, int NCH> GD void segw_h_factor_call(typename Blk<MODEL, true>::Args a, int c) {
    if (c == 0) segw_h_factor_chain<MODEL, NCH, 0>(a);
    else if (c == 1) segw_h_factor_chain<MODEL, NCH, (NCH > 2 ? 1 : 0)>(a);
    else segw_h_factor_chain<MODEL, NCH, (NCH > 2 ? 2 : 0)>(a);
}
template <int MODEL, int NCH> __device__ __noinline__ void segw_h_fold_call(typename Blk<MODEL, true>::Args a, int h, int level) {
    Blk<MODEL, true> B(a, gusto_dyn_lds);
    double* fail = gusto_dyn_lds + LdsC<MODEL, true>::misc + 8;
    if constexpr (NCH == 2) seg_fold_factor2<MODEL>(B, fail);
    else if (level == 1) seg_fold_tree_level1<MODEL>(B, fail, h);
    else seg_fold_tree_level2<MODEL>(B, fail);
}
template <int MODEL, int NCH> __device__ __noinline__ void segw_h_back_call(typename Blk<MODEL, true>::Args a, int c) {
    Blk<MODEL, true> B(a, gusto_dyn_lds);
    seg_chain_backward<MODEL, NCH>(B, c);
}
template <int MODEL, int NCH> __device__ __noinline__ void segw_h_fwd_call(typename Blk<MODEL, true>::Args a, int c) {
    Blk<MODEL, true> B(a, gusto_dyn_lds);
    seg_chain_forward<MODEL, NCH>(B, c);
}
template <int MODEL, int NCH> __device__ __noinline__ void segw_h_rows_call(typename Blk<MODEL, true>::Args a, bool step, int hi, int rank, int nshare) {
    Blk<MODEL, true> B(a, gusto_dyn_lds);
    if (step) segw_rows_step_helper<MODEL, NCH>(B, hi, rank, nshare);
    else segw_rows_resid_helper<MODEL, NCH>(B, hi, rank, nshare);
}
template <int MODEL, int NCH> __device__ __noinline__ void segw_h_costate_call(typename Blk<MODEL, true>::Args a) {
    Blk<MODEL, true> B(a, gusto_dyn_lds);
    costate_pass_seg<MODEL, NCH>(SweepView<MODEL>::make(B));
    if (B.tid == 0) costate_close_x1<MODEL>(B, 0.5 * B.dt, gusto_dyn_lds + LdsC<MODEL, true>::misc + 16);
}
// (two waves per problem: the one helper's small phases stay inlined in the kernel, with its view of the problem -- some 45 base
// pointers, scalar loads from the kernel arguments -- rebuilt when the problem changes and not per command: ten commands per interior
// point iteration, 3.5 % of the kernel's time if all were calls.  Its factor sweep and the merge ARE calls: 2 - 4 % faster with a
// register allocation of their own)
template <int MODEL> GD void segw_helper2(const KParams& P, double* lds) {
    constexpr int NCH = 2;
    using SB = SegB<MODEL, NCH>;
    using BLK = Blk<MODEL, true>;
    using C = LdsC<MODEL, true>;
    const LPtr<double> L = lds;
    const int sb = P.ll.seg, mb = sb + SB::MBX;
    asm volatile("s_barrier" ::: "memory");
    int cmd = (int)L[mb];
    while (cmd != SEGW_EXIT) {
        typename BLK::Args a;
        a.Pk = (const KParams*)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();   // (inlined into the kernel; KParams is its first argument)
        const double pb = L[mb + 2];
        a.b = (int)pb; a.slot = (int)L[mb + 3]; a.goalmask = (unsigned)L[mb + 4]; a.boxmask = (unsigned)L[mb + 5]; a.dt = L[mb + 6];
        BLK B(a, lds);
        do {
            if (cmd == SEGW_FACTOR) {   // (the two big pieces are calls here too: a register allocation of their own)
#ifdef GUSTO_PROFILE   // (slots 29 .. 31: its factor sweep, its backward sweeps, the merge)
                const long long t0 = clock64();
#endif
                segw_h_factor_chain<MODEL, NCH, 0>(a);
#ifdef GUSTO_PROFILE
                const long long t1 = clock64();
#endif
                segw_barrier();
#ifdef GUSTO_PROFILE
                const long long t2 = clock64();
#endif
                segw_h_fold_call<MODEL, NCH>(a, 1, 0);
#ifdef GUSTO_PROFILE
                if (B.tid == 0 && P.prof) {
                    const long long t3 = clock64();
                    long long* o = P.prof + (size_t)B.b * PROF_N;
                    o[29] += t1 - t0; o[31] += t3 - t2;
                }
#endif
            } else if (cmd == SEGW_BACK) {
#ifdef GUSTO_PROFILE
                const long long t0 = clock64();
#endif
                seg_chain_backward<MODEL, NCH>(B, 0);
#ifdef GUSTO_PROFILE
                if (B.tid == 0 && P.prof) P.prof[(size_t)B.b * PROF_N + 30] += clock64() - t0;
#endif
            } else if (cmd == SEGW_FWD) {
                seg_chain_forward<MODEL, NCH>(B, 0);
            } else if (cmd == SEGW_ROWS_R) {         // (the helper takes ALL obstacle rows: the main wave has the other rows and the stage cost)
                segw_rows_resid_helper<MODEL, NCH>(B, 0, 0, 1);
            } else if (cmd == SEGW_STEP) {
                segw_rows_step_helper<MODEL, NCH>(B, 0, 0, 1);
            } else if (cmd == SEGW_STEP_CS) {   // (these as calls too: measured slower, 46.5 against 45.4 ms -- the view rebuilt per call)
                costate_pass_seg<MODEL, NCH>(SweepView<MODEL>::make(B));
                if (B.tid == 0) costate_close_x1<MODEL>(B, 0.5 * B.dt, lds + C::misc + 16);
                segw_rows_step_helper<MODEL, NCH>(B, 0, 1, 2);   // (then half of the obstacle rows)
            }
            segw_barrier();
            asm volatile("s_barrier" ::: "memory");
            cmd = (int)L[mb];
        } while (cmd != SEGW_EXIT && L[mb + 2] == pb);
    }
}
template <int MODEL, int NCH> GD void segw_helper(const KParams& P, double* lds) {
    if constexpr (NCH == 2) { segw_helper2<MODEL>(P, lds); return; }
    using SB = SegB<MODEL, NCH>;
    using BLK = Blk<MODEL, true>;
    const LPtr<double> L = lds;
    const int sb = P.ll.seg, mb = sb + SB::MBX;
    const int h = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)), c = h - 1;
    for (;;) {
        asm volatile("s_barrier" ::: "memory");
        const int cmd = (int)L[mb];
        if (cmd == SEGW_EXIT) return;
        typename BLK::Args a;
        a.Pk = (const KParams*)(uintptr_t)__builtin_amdgcn_kernarg_segment_ptr();   // (this function is inlined into the kernel; KParams is its first argument)
        a.b = (int)L[mb + 2]; a.slot = (int)L[mb + 3]; a.goalmask = (unsigned)L[mb + 4]; a.boxmask = (unsigned)L[mb + 5]; a.dt = L[mb + 6];
        if (cmd == SEGW_FACTOR) {
#ifdef GUSTO_PROFILE   // (slots 29 .. 31, helper 1: its factor sweep, its backward sweeps, its share of the merges)
            const long long t0 = clock64();
#endif
            segw_h_factor_call<MODEL, NCH>(a, c);
#ifdef GUSTO_PROFILE
            const long long t1 = clock64();
#endif
            segw_barrier();
#ifdef GUSTO_PROFILE
            const long long t2 = clock64();
#endif
            if (h == 1 || h == 3) segw_h_fold_call<MODEL, NCH>(a, h, 1);   // the tree's first level: two merges side by side
            segw_barrier();
            if (h == 1) segw_h_fold_call<MODEL, NCH>(a, h, 2);
#ifdef GUSTO_PROFILE
            if (h == 1 && (threadIdx.x & 63) == 0 && P.prof) {
                const long long t3 = clock64();
                long long* o = P.prof + (size_t)a.b * PROF_N;
                o[29] += t1 - t0; o[31] += t3 - t2;
            }
#endif
        } else if (cmd == SEGW_BACK) {
#ifdef GUSTO_PROFILE
            const long long t0 = clock64();
#endif
            segw_h_back_call<MODEL, NCH>(a, c);
#ifdef GUSTO_PROFILE
            if (h == 1 && (threadIdx.x & 63) == 0 && P.prof) P.prof[(size_t)a.b * PROF_N + 30] += clock64() - t0;
#endif
        } else if (cmd == SEGW_FWD) {
            segw_h_fwd_call<MODEL, NCH>(a, c);
        } else if (cmd == SEGW_ROWS_R || cmd == SEGW_STEP) {   // (the helpers take ALL obstacle rows: the main wave has the other rows and the stage cost)
            segw_h_rows_call<MODEL, NCH>(a, cmd == SEGW_STEP, c, h - 1, NCH - 1);
        } else if (cmd == SEGW_STEP_CS) {
            if (h == 1) {
                segw_h_costate_call<MODEL, NCH>(a);
                if constexpr (NCH == 2) segw_h_rows_call<MODEL, NCH>(a, true, c, 1, 2);   // (the only helper: then half of the obstacle rows)
            } else segw_h_rows_call<MODEL, NCH>(a, true, c, h - 2, NCH - 2);
        }
        segw_barrier();
    }
}

// the main wave's share of the three sequential phases (the last chain), as called phases
template <int MODEL, int NCH> __device__ __noinline__ void factor_sweep_seg_call(typename Blk<MODEL, true>::Args a, Prof* pf) {
    Blk<MODEL, true> B(a, gusto_dyn_lds);
    seg_chain_factor<MODEL, NCH>(SweepView<MODEL>::make(B), NCH - 1, gusto_dyn_lds + LdsC<MODEL, true>::misc + 8, *pf);
}
template <int MODEL, int NCH> __device__ __noinline__ void backward_sweep_seg_call(typename Blk<MODEL, true>::Args a) {
    Blk<MODEL, true> B(a, gusto_dyn_lds);
    seg_chain_backward<MODEL, NCH>(B, NCH - 1);
}
template <int MODEL, int NCH> __device__ __noinline__ void forward_sweep_seg_call(typename Blk<MODEL, true>::Args a) {
    Blk<MODEL, true> B(a, gusto_dyn_lds);
    using SB = SegB<MODEL, NCH>;
    constexpr int n = MT<MODEL>::n;
    const int N = B.N;
    // the end states of the chains in front of an interface: the interface's xi (their own sweeps stop one knot early)
    if (B.tid < n * (NCH - 1)) {
        const int j = B.tid / n, i = B.tid % n;
        B.dY[(seg_lo(j + 1, N, NCH) - 1) * n + i] = B.lds[B.P.ll.seg + SB::XI(0) + 16 * j + i];
    }
    seg_chain_forward<MODEL, NCH>(B, NCH - 1);
}

}  // namespace gusto
