// common.hpp -- shared definitions for the batched GuSTO SCP kernels (gfx950 / CDNA4, wave64, fp64).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

#include "gusto_hip.h"

namespace gusto {

#define GD __device__ __forceinline__
typedef double v4d __attribute__((ext_vector_type(4)));   // accumulator tile of v_mfma_f64_16x16x4_f64

// row kinds of the convex subproblem (scp_gusto.jl:192-314)
constexpr int ROW_HARD = 0;     // hard inequality (convex_control_ineq, BoxGoal rows)          :213-221,236-245
constexpr int ROW_PEN = 1;      // L1-penalised state inequality, part of the post-check         :281-295
constexpr int ROW_PEN_TR = 2;   // L1-penalised trust region, not part of the post-check         :265-279
constexpr int ROW_PEN_EQ = 3;   // j=2 half of a penalised equality, |h| < eps post-check        :297-311
constexpr int ROW_HARD_EQ = 4;  // j=1 half of a penalised equality: 0 <= s1 <= w*h + eps
constexpr int ROW_EQ = 5;       // hard equality h = 0 (TrajOpt's convex_state_eq rows)                  scp_trajopt.jl:200-208
GD bool row_is_hard(int kind) { return kind == ROW_HARD || kind == ROW_HARD_EQ; }

// per-row interior point state, stored [var][slot][k] so that lane k's accesses coalesce
constexpr int RS_T = 0, RS_LAM = 1, RS_LAMB = 2, RS_S = 3, RS_DT = 4, RS_DL = 5, RS_DS = 6, RS_KA = 7, RS_KB = 8,
              RS_NVAR = 9;

#ifndef GUSTO_DUBINS_WAVES
#define GUSTO_DUBINS_WAVES 2
#endif
#ifndef GUSTO_WAVES_PER_EU
#define GUSTO_WAVES_PER_EU 1
#endif

#ifndef GUSTO_TO_SWEEP_CALL
// the phases and sweeps of the astrobee TrajOpt kernels as real calls (register allocations of their own): inlined, trajopt_kernel<5>
// spilled 3436 registers (6.6 KB of scratch per lane), <6> 4046 (7.6 KB); called, 491 / 554 -- B = 256 astrobeeSE3 831 -> 542 ms,
// B = 128 manifold 1128 -> 893 ms, bit-identical
#define GUSTO_TO_SWEEP_CALL true
#endif
#ifndef GUSTO_TO4_SWEEP_CALL
#define GUSTO_TO4_SWEEP_CALL true   // ... of the freeflyer TrajOpt kernel (B = 1024: 58.3 -> 53.4 ms; the same schedule, results within 1e-9: contraction differs across the call)
#endif
#ifndef GUSTO_USE_MFMA
#define GUSTO_USE_MFMA true   // -DGUSTO_USE_MFMA=false: the VALU two-step contraction instead (A/B measurements)
#endif
// TrajOpt variants of two models (src/scp/scp_trajopt.jl; internal ids, not part of gusto_model_id): a knot carries the
// controls (u_k, d_k), d_k = the n defect variables of the interval (k, k+1) -- the L1-penalised dynamics.  In the LQR form
// of the Newton system a defect moves y_k = F_k x_k + b_k u_k + d_k directly (Gam_d = I) and not x_k (b_d = 0); MT::NDEF
// marks the few places that differ (ipm.hpp).  They run the generic multi-wave phases (one wave of them for N <= 64).
constexpr int GUSTO_TO_FREEFLYER_SE2 = 4, GUSTO_TO_ASTROBEE_SE3 = 5, GUSTO_TO_ASTROBEE_SE3_MANIFOLD = 6;
// vanishing quadratic cost on the defects next to their L1 penalty (the oracle's GO_TRAJOPT_DEFECT_REG)
constexpr double TRAJOPT_DEFECT_REG = 1e-4;
template <int MODEL> struct MT;
template <> struct MT<GUSTO_FREEFLYER_SE2> {
    static constexpr int NDEF = 0;   // (no defect controls: the dynamics are hard rows)
    static constexpr int n = 6, m = 3, WS = 2, NFIX = 3, NHU = 2;
    static constexpr int WAVES_PER_EU = GUSTO_WAVES_PER_EU;   // register budget of the one-wave kernel: 512 / this
    static constexpr bool SWEEP_CALL = false;   // factor sweep as a function call (ipm.hpp:factor_sweep)
    static constexpr bool MFMA = false;   // dense per-knot products of the factor sweep on v_mfma_f64_16x16x4_f64
    static constexpr int SCHED_PROBE = 2;   // default number of one-trip probing slices of the longest-first scheduler
    static constexpr int SCHED_SLICE = 4;   // then slices of 4 trips for problems of penalty level 0 (34.1 vs 34.65 ms; the 12/13-state models lose with any)
    static constexpr bool LTI = true, HAS_OBS = true;
    // Double integrator (freeflyer_se2.jl:121,178-179: A = [0 I; 0 0], B = [0; diag]): Phi = I + dt A and
    // Gam = 2 (I + dt/2 A) b have at most TWO nonzeros per column of [Phi Gam], at rows pg_r0(c), pg_r1(c)
    // (a row holding a structural zero when the column has a single entry).  The factor sweep contracts over those
    // two rows only.
    static constexpr bool PG2 = true;
    static constexpr int pg_r0(int c) { return c < 3 ? c : (c < 6 ? c - 3 : c - 6); }
    static constexpr int pg_r1(int c) { return c < 3 ? c + 3 : (c < 6 ? c : c - 3); }
    // the same structure for the stage-parallel products: M = I + dt/2 A, B = [0; diag], Gam = 2 M (dt/2 B).
    // Unrolled loops test these with compile-time indices, so the structural zeros cost neither a load nor an FMA.
    static constexpr bool Anz(int i, int j) { return j == i + 3; }
    static constexpr bool Mnz(int i, int j) { return i == j || j == i + 3; }
    static constexpr bool Bnz(int i, int j) { return i == j + 3; }
    static constexpr bool Gnz(int i, int j) { return i == j || i == j + 3; }
    static constexpr bool Hnz(int, int) { return true; }   // (the trust region row couples every pair of states)
};
template <> struct MT<GUSTO_DUBINS_CAR> {
    static constexpr int NDEF = 0;   // (no defect controls: the dynamics are hard rows)
    static constexpr int n = 3, m = 1, WS = 2, NFIX = 6, NHU = 2;
    static constexpr int WAVES_PER_EU = GUSTO_DUBINS_WAVES;   // register budget of the one-wave kernel: 512 / this
    static constexpr bool SWEEP_CALL = false;   // factor sweep as a function call (ipm.hpp:factor_sweep)
    static constexpr bool MFMA = false;   // dense per-knot products of the factor sweep on v_mfma_f64_16x16x4_f64
    static constexpr int SCHED_PROBE = 1; static constexpr int SCHED_SLICE = 0;   // (short problems: 2 slices cost more than they order -- 316 vs 211 ms at B = 65 536)
    static constexpr bool LTI = false, HAS_OBS = false;
    static constexpr bool PG2 = false;
    static constexpr int pg_r0(int) { return 0; }
    static constexpr int pg_r1(int) { return 0; }
    static constexpr bool Anz(int, int) { return true; }
    static constexpr bool Mnz(int, int) { return true; }
    static constexpr bool Bnz(int, int) { return true; }
    static constexpr bool Gnz(int, int) { return true; }
    static constexpr bool Hnz(int, int) { return true; }
};
template <> struct MT<GUSTO_ASTROBEE_SE3> {
    static constexpr int NDEF = 0;   // (no defect controls: the dynamics are hard rows)
    static constexpr int n = 12, m = 6, WS = 3, NFIX = 3, NHU = 2;
    static constexpr int WAVES_PER_EU = 1;   // register budget of the one-wave kernel: 512 / this
    static constexpr bool SWEEP_CALL = true;   // factor sweep as a function call (ipm.hpp:factor_sweep)
    static constexpr bool MFMA = GUSTO_USE_MFMA;   // dense per-knot products of the factor sweep on v_mfma_f64_16x16x4_f64
    static constexpr int SCHED_PROBE = 1; static constexpr int SCHED_SLICE = 0;   // (measured with raised-penalty problems ahead of fresh ones: 123.0 / 127.1 / 132.7 ms for 1 / 2 / 3 slices)
    static constexpr bool LTI = false, HAS_OBS = true;
    static constexpr bool PG2 = false;
    static constexpr int pg_r0(int) { return 0; }
    static constexpr int pg_r1(int) { return 0; }
    // x = (r, v, p MRP, w): A = [0 I 0 0; 0 0 0 0; 0 0 App Apw; 0 0 0 Aww] in 3x3 blocks, so M = (I - dt/2 A)^-1 is
    // block upper triangular with the same pattern plus the diagonal; B = [0; I/m; 0; J^-1]; Gam = 2 M (dt/2 B)
    static constexpr bool Anz(int i, int j) { return i < 3 ? j == i + 3 : (i < 6 ? false : (i < 9 ? j >= 6 : j >= 9)); }
    static constexpr bool Mnz(int i, int j) { return i < 3 ? (j == i || j == i + 3) : (i < 6 ? j == i : (i < 9 ? j >= 6 : j >= 9)); }
    static constexpr bool Bnz(int i, int j) { return j < 3 ? i == j + 3 : i == j + 6; }
    static constexpr bool Gnz(int i, int j) { return j < 3 ? (i == j || i == j + 3) : i >= 6; }
    // Hessian of the rows in x: every row but the trust region touches ONE of the blocks r, v, p, w (obstacles r, speed v,
    // rate w, goal rows single coordinates), so it is block diagonal; the trust region row is a diagonal plus ONE dyad
    // over all states, which resid_phase carries separately (rank one) instead of filling the 78 entries with it
    static constexpr bool Hnz(int i, int j) { return i / 3 == j / 3; }
};
template <> struct MT<GUSTO_ASTROBEE_SE3_MANIFOLD> {
    static constexpr int NDEF = 0;   // (no defect controls: the dynamics are hard rows)
    static constexpr int n = 13, m = 6, WS = 3, NFIX = 5, NHU = 2;
    static constexpr int WAVES_PER_EU = 1;   // register budget of the one-wave kernel: 512 / this
    static constexpr bool SWEEP_CALL = true ;   // factor sweep as a function call (ipm.hpp:factor_sweep)
    static constexpr bool MFMA = GUSTO_USE_MFMA;   // dense per-knot products of the factor sweep on v_mfma_f64_16x16x4_f64
    static constexpr int SCHED_PROBE = 1; static constexpr int SCHED_SLICE = 0;   // (130.1 / 133.7 / 130.9 ms for 1 / 2 / 3 slices)
    static constexpr bool LTI = false, HAS_OBS = true;
    static constexpr bool PG2 = false;
    static constexpr int pg_r0(int) { return 0; }
    static constexpr int pg_r1(int) { return 0; }
    // x = (r, v, q, w): the block pattern of astrobeeSE3 with a 4-row quaternion block
    static constexpr bool Anz(int i, int j) { return i < 3 ? j == i + 3 : (i < 6 ? false : (i < 10 ? j >= 6 : j >= 10)); }
    static constexpr bool Mnz(int i, int j) { return i < 3 ? (j == i || j == i + 3) : (i < 6 ? j == i : (i < 10 ? j >= 6 : j >= 10)); }
    static constexpr bool Bnz(int i, int j) { return j < 3 ? i == j + 3 : i == j + 7; }
    static constexpr bool Gnz(int i, int j) { return j < 3 ? (i == j || i == j + 3) : i >= 6; }
    // (no trust region row on the manifold: the Hessian of the rows is block diagonal in r, v, q (4 rows), w)
    static constexpr int hblk(int i) { return i < 3 ? 0 : (i < 6 ? 1 : (i < 10 ? 2 : 3)); }
    static constexpr bool Hnz(int i, int j) { return hblk(i) == hblk(j); }
};

template <> struct MT<GUSTO_TO_FREEFLYER_SE2> {
    using G = MT<GUSTO_FREEFLYER_SE2>;
    static constexpr int NDEF = 6, n = 6, m = 3 + NDEF, WS = 2, NFIX = 3, NHU = 2 + 2 * NDEF;
    static constexpr int WAVES_PER_EU = 1, SCHED_PROBE = 0, SCHED_SLICE = 0;
    static constexpr bool SWEEP_CALL = GUSTO_TO4_SWEEP_CALL, MFMA = false, LTI = false, HAS_OBS = true, PG2 = false;
    static constexpr int pg_r0(int) { return 0; }
    static constexpr int pg_r1(int) { return 0; }
    static constexpr bool Anz(int i, int j) { return G::Anz(i, j); }
    static constexpr bool Mnz(int i, int j) { return G::Mnz(i, j); }
    static constexpr bool Bnz(int i, int j) { return j < G::m && G::Bnz(i, j); }
    static constexpr bool Gnz(int i, int j) { return j < G::m ? G::Gnz(i, j) : i == j - G::m; }
    static constexpr bool Hnz(int, int) { return true; }
};
template <> struct MT<GUSTO_TO_ASTROBEE_SE3> {
    using G = MT<GUSTO_ASTROBEE_SE3>;
    static constexpr int NDEF = 12, n = 12, m = 6 + NDEF, WS = 3, NFIX = 3, NHU = 2 + 2 * NDEF;
    static constexpr int WAVES_PER_EU = 1, SCHED_PROBE = 0, SCHED_SLICE = 0;
    static constexpr bool SWEEP_CALL = GUSTO_TO_SWEEP_CALL, MFMA = false, LTI = false, HAS_OBS = true, PG2 = false;
    static constexpr int pg_r0(int) { return 0; }
    static constexpr int pg_r1(int) { return 0; }
    static constexpr bool Anz(int i, int j) { return G::Anz(i, j); }
    static constexpr bool Mnz(int i, int j) { return G::Mnz(i, j); }
    static constexpr bool Bnz(int i, int j) { return j < G::m && G::Bnz(i, j); }
    static constexpr bool Gnz(int i, int j) { return j < G::m ? G::Gnz(i, j) : i == j - G::m; }
    static constexpr bool Hnz(int, int) { return true; }   // (the hard trust region row couples every pair of states)
};

template <> struct MT<GUSTO_TO_ASTROBEE_SE3_MANIFOLD> {
    using G = MT<GUSTO_ASTROBEE_SE3_MANIFOLD>;
    // fixed state rows of a knot: the +- band of the (hard) quaternion norm row, orientation sign, speed, rate
    static constexpr int NDEF = 13, n = 13, m = 6 + NDEF, WS = 3, NFIX = 5, NHU = 2 + 2 * NDEF;
    static constexpr int WAVES_PER_EU = 1, SCHED_PROBE = 0, SCHED_SLICE = 0;
    static constexpr bool SWEEP_CALL = GUSTO_TO_SWEEP_CALL, MFMA = false, LTI = false, HAS_OBS = true, PG2 = false;
    static constexpr int pg_r0(int) { return 0; }
    static constexpr int pg_r1(int) { return 0; }
    static constexpr bool Anz(int i, int j) { return G::Anz(i, j); }
    static constexpr bool Mnz(int i, int j) { return G::Mnz(i, j); }
    static constexpr bool Bnz(int i, int j) { return j < G::m && G::Bnz(i, j); }
    static constexpr bool Gnz(int i, int j) { return j < G::m ? G::Gnz(i, j) : i == j - G::m; }
    static constexpr bool Hnz(int, int) { return true; }
};
// TrajOpt keeps convex_state_eq rows hard (`== 0`, scp_trajopt.jl:200-208): the manifold model's linearised quaternion norm is an
// EQUALITY ROW of the interior point method (round 6; rounds 3-5 carried it as the band |h_k| <= 1e-4, which moved the optimum of a
// subproblem by 7 % of its objective and decided 3 % of whole runs at its edge) -- h(x_k) = 0 with a multiplier eta of either sign
// and the constraint regularisation delta of a primal-dual method (Ipopt's delta_c):  grad h' dx - delta d_eta = -h, eliminated per
// row like every other row: H += grad grad' / delta, coefficient eta + h / delta, d_eta = (h + grad' dx) / delta.  No slack, no
// barrier, no step-length limit, not part of the complementarity measure; |h| is part of the primal residual, so a solve that
// stops OPTIMAL holds the equality to the 1e-8 stopping tolerance (the oracle's GO_EQ_DELTA, ROW_EQ).
constexpr double TRAJOPT_EQ_DELTA = 1e-8;   // (1e-6: 39 interior point iterations per subproblem instead of 15; 1e-10: solves at mu = 125 break down)

// Warm start of the interior point method (gusto_ipm_opts: mu_warm, mu_warm_gain, mu_warm_max; gusto_hip.h).  The model
// defaults, resolved on the host when a launch is prepared (mu_warm < 0 = "the model's"): measured on the BASELINE batches
// (tools/ipm_opts_scan.py).  dubins_car -- bound rows only, most of them inactive -- is best started almost ON the boundary
// at every trip (3.06 M interior point iterations for the config-3 batch at 1e-10, 3.08 M at 1e-9, 4.15 M at 1e-4); the
// models with obstacle rows pay for a start that close to the boundary whenever the linearisation point has moved (their
// first warm trip took 17 iterations against 11 for the cold one), hence the level that follows the last trajectory change.
inline void warm_defaults(int model, gusto_ipm_opts& io) {
    // (complementarity floor: a tenth of the stopping level for the manifold model -- at 1e-11 the dual residual of its converged
    // solves sits at its noise floor, 5e-8, for ten iterations before it passes the 3e-8 test: -14 % KKT solves on config 5 --,
    // a hundredth for the others, whose hard problems agree with the oracle 30 x better that way)
    if (io.mu_floor < 0) io.mu_floor = (model == GUSTO_ASTROBEE_SE3_MANIFOLD) ? 1e-10 : 1e-11;
    // (Mehrotra's centring parameter is bounded for GuSTO's subproblems, gusto_hip.h; TrajOpt's -- close to linear programs in
    // their defect variables -- keep the unbounded rule they were validated with)
    if (io.sigma_max < 0) io.sigma_max = (model == GUSTO_TO_FREEFLYER_SE2 || model == GUSTO_TO_ASTROBEE_SE3 || model == GUSTO_TO_ASTROBEE_SE3_MANIFOLD) ? 0.0 : 0.1;
    if (!(io.mu_warm < 0)) { if (io.mu_warm_gain < 0) io.mu_warm_gain = 0.0; return; }
    switch (model) {
    case GUSTO_DUBINS_CAR: io.mu_warm = 1e-9; io.mu_warm_gain = 0.0; io.mu_warm_max = 1e-9; break;
    case GUSTO_FREEFLYER_SE2: io.mu_warm = 1e-4; io.mu_warm_gain = 0.1; io.mu_warm_max = 1e-2; break;
    case GUSTO_ASTROBEE_SE3: io.mu_warm = 1e-6; io.mu_warm_gain = 1.0; io.mu_warm_max = 1e-2; break;
    default: io.mu_warm = 1e-4; io.mu_warm_gain = 1.0; io.mu_warm_max = 1e-2; break;   // (astrobeeSE3manifold)
    }
}
// An interior point solve whose mean complementarity has grown to this multiple of max(1, its value at the start point) is
// diverging -- the subproblem is infeasible (dubins_car: 7 % of the config-3 batch at trip 0, certified by an LP in
// tests/test_oracle_scp.py; mu then runs to 1e15 and the multipliers to 1e20 before the iteration cap ends the solve with the
// same verdict, GUSTO_SOLVER_FAILED): stop there instead of at the cap.  Convergent solves never come near it (their mu
// falls from the first iteration on; checked on every BASELINE batch: identical statuses with and without the test).
constexpr double IPM_DIVERGED = 1e3;
// start level of a warm subproblem: c = the convergence measure of the previous SCP iteration
GD double warm_mu(const gusto_ipm_opts& io, double c) {
    if (io.mu_warm == 0.0) return 0.0;
    return fmin(fmax(io.mu_warm, io.mu_warm_max), fmax(io.mu_warm, io.mu_warm_gain * c * c));
}

// symmetric packed index (upper triangle, row-major)
GD constexpr int sidx(int i, int j, int n) {
    return i <= j ? i * n - i * (i - 1) / 2 + (j - i) : j * n - j * (j - 1) / 2 + (i - j);
}

// Stage records of the per-problem workspace: every per-knot factor array is padded to a multiple of 64 doubles per
// knot, so a sweep stores/loads a record with ONE unconditional, fully coalesced access per 64 entries at
// `uniform base + (k * stride + lane)` (SGPR base + VGPR offset; no per-lane pointers, no predication).
// K (m x n), D (m x n) and S^-1 (m x m) of a knot share one record.
template <int MODEL> struct Rec {
    using T = MT<MODEL>;
    static constexpr int n = T::n, m = T::m, NZ = n + m;
    static constexpr int r64(int c) { return (c + 63) / 64 * 64; }
    // (+1: every record keeps at least one padding slot, the target of the lanes that hold no entry of a tile store)
    // (S^-1 of the TrajOpt variants as its upper triangle was built and measured in round 5 -- astrobeeSE3 B = 256 427 -> 397 ms -- and
    // dropped: the changed rounding moves the subproblems at the edge of break-down, profiles/r05_trajopt_stage.txt)
    static constexpr bool S_TRI = false;
    static constexpr int NS = S_TRI ? m * (m + 1) / 2 : m * m;
    static constexpr int SQQ = r64(NZ * (NZ + 1) / 2), SNN = r64(n * n + 1), SKD = r64(2 * m * n + NS + 1);
    static constexpr int oK = 0, oD = m * n, oS = 2 * m * n;
    static constexpr int KD_DUMMY = 2 * m * n + NS;   // the padding slot behind the entries
    // record position of S^-1[i][l]
    static constexpr int sS(int i, int l) {
        if (!S_TRI) return oS + i * m + l;
        const int a = i < l ? i : l, b = i < l ? l : i;
        return oS + a * m - a * (a - 1) / 2 + (b - a);
    }
};

// Compact [Phi Gam] record of the matrix-core models (MT::MFMA): only the structural nonzeros of Phi = 2 M - I (MT::Mnz) and
// Gam (MT::Gnz) -- 60 of 216 doubles for astrobeeSE3, 73 of 247 for the manifold model -- Phi first, column by column (the
// adjoint costate recursion reads a column per lane), then Gam row by row.  Written by linearize() next to the dense block,
// read by the factor sweep (scattered into its dense LDS operand buffer) and by adjoint_sweep_1w.
template <int MODEL> struct SpPG {
    using T = MT<MODEL>;
    static constexpr int n = T::n, m = T::m;
    static constexpr int pos_phi(int l, int i) {   // rank of Phi[l][i] among the nonzeros, column-major
        int c = 0;
        for (int i2 = 0; i2 < n; i2++)
            for (int l2 = 0; l2 < n; l2++)
                if (T::Mnz(l2, i2) && (i2 < i || (i2 == i && l2 < l))) c++;
        return c;
    }
    static constexpr int NPHI = pos_phi(0, n);
    static constexpr int pos_gam(int l, int j) {   // ... of Gam[l][j], row-major, behind Phi
        int c = NPHI;
        for (int l2 = 0; l2 < n; l2++)
            for (int j2 = 0; j2 < m; j2++)
                if (T::Gnz(l2, j2) && (l2 < l || (l2 == l && j2 < j))) c++;
        return c;
    }
    static constexpr int NS = pos_gam(n, 0);
    static constexpr int S = (NS + 15) / 16 * 16;   // stride per knot: whole 128-byte lines
    static constexpr bool USE = T::MFMA && T::NDEF == 0;
};

// per-problem global workspace, offsets in doubles
struct WsLayout {
    int nslot;
    size_t rowstate, obs_nh, obs_c0, obs_mask, PG, PGS, QQ, Paft, Piaft, KD, Phicl, pvt, to_traj, total;
};
template <int MODEL> inline WsLayout make_ws_layout(int N, int n_obs) {
    using T = MT<MODEL>;
    constexpr int n = T::n, m = T::m, NZ = n + m;
    WsLayout L;
    L.nslot = T::NFIX + n_obs + 2 * n + T::NHU;
    size_t o = 0;
    auto take = [&](size_t c) { size_t r = o; o += (c + 1) & ~size_t(1); return r; };
    L.rowstate = take((size_t)RS_NVAR * L.nslot * N);
    L.obs_nh = take((size_t)n_obs * T::WS * N);
    L.obs_c0 = take((size_t)n_obs * N);
    L.obs_mask = take((size_t)N);
    L.PG = take((size_t)(T::LTI ? 1 : N) * n * NZ);
    L.PGS = take(SpPG<MODEL>::USE ? (size_t)N * SpPG<MODEL>::S : 0);
    using R = Rec<MODEL>;
    L.QQ = take((size_t)N * R::SQQ);
    L.Paft = take((size_t)(N + 1) * R::SNN) + R::SNN;    // record -1 exists: the sweep stores P_{k-1} unconditionally
    L.Piaft = take((size_t)(N + 1) * R::SNN) + R::SNN;
    L.KD = take((size_t)N * R::SKD);
    L.Phicl = take((size_t)N * R::SNN);
    L.pvt = take((size_t)N * (5 * n + 5 * m));   // rd qrd dXs | dUs qu dv | gAx gBx gAu gBu (corrector row sums)
    L.to_traj = take(T::NDEF > 0 ? (size_t)2 * N * (n + m) : 0);   // TrajOpt: old_penalty_traj and old_convex_traj (X | U each)
    L.total = o;
    return L;
}

// LDS layout, offsets in doubles.  The cooperative working set of the sweeps sits first at compile-time offsets
// (so its addresses are instruction immediates, not SGPRs); the per-knot vectors follow, `vec(i)` = i-th vector.
// ONE = one wave per problem (N <= 64): no multi-wave sweep buffers (sK sD sW sV); time-varying models keep two
// [Phi Gam] buffers (the knot-0 operand comes from block 0 of the global PG array, linearize()); the large models, whose
// sweep forms T, then H, then Z, let T share the memory of Z, and their goal-system inverse borrows sHh as scratch.
// That is what fits 3 problems of the 12/13-state models into the 160 KB of a CU.
template <int MODEL, bool ONE> struct LdsC {
    using T = MT<MODEL>;
    static constexpr int n = T::n, m = T::m, NZ = n + m;
    static constexpr bool BIG = n > 8;
    static constexpr int NPGB = (ONE && !T::LTI) ? 2 : 3;
    static constexpr int sP = 0, sPi = sP + n * n, sPG = sPi + n * n, sT0 = sPG + NPGB * n * NZ,
                         sHh = sT0 + ((ONE && BIG) ? 0 : n * NZ), sZ = sHh + NZ * NZ, sT = (ONE && BIG) ? sZ : sT0,
                         sGd = sZ + NZ * n, misc = sGd + ((ONE && BIG) ? 1 : 2) * n * n, sgoal = misc + 64 /* goal_lo of the problem */, lut = misc + 80,
                         vecs1w = (lut + (NZ * (NZ + 1) / 2 + 1) / 2 + 1 + 1) & ~1,   // (even: 16-byte aligned rows of the n-vectors, ds_read_b128)
                         sK = vecs1w, sD = sK + m * n, sW = sD + m * n, sV = sW + m * n, vecsmw = sV + m * n,
                         vecs = ONE ? vecs1w : vecsmw;
    // per-knot vectors shared between lanes: n-vectors (Xw dY pv cv rv nu nun) then the m-vector Uw; vectors only
    // their own knot touches (rd qrd dXs | dUs qu dv) live in the per-problem global workspace, the linearisation
    // point (Xp, Up) is read from the problem's trajectory in HBM/L2
    static constexpr int NVN = 7, NVM = 1;
    // One-wave problems of the small models also keep the closed-loop matrices Phicl_k (written once by the factor
    // sweep, read by the four vector sweeps of an iteration) in LDS, behind the vectors: [N][n*n].  4 problems per
    // CU are register-limited anyway, so up to 40 KB of LDS per problem are free.
    // The double integrator (MT::PG2) keeps K_k | D_k | S_k^-1 there instead (2 m n + m (m + 1) / 2 doubles per knot):
    // Phicl = Phi - Gam K is ONE fma per entry from K and the model constants, so the vector sweeps rebuild their
    // operands from K, and the stage-parallel phases -- which walked these records in global memory, one cache line per
    // lane per load, ~160 loads per interior point iteration -- read them from LDS.
    // (the 3-state model too: it keeps BOTH K | D | S^-1 -- 10 doubles per knot, for the stage-parallel phases -- and Phicl,
    // for the vector sweeps, in LDS; no select chains for the K | D | S^-1 record, no QQ record in global memory)
    static constexpr bool KD_LDS = ONE && n <= 8;
    static constexpr bool PHI_FROM_K = KD_LDS && T::PG2;   // the vector sweeps rebuild Phicl from K (double integrator)
    // the 3-state time-varying model keeps [Phi Gam] of every knot in LDS as well (12 doubles per knot): linearize() writes
    // it there, the factor sweep reads its stage operands in place (no prefetch, no staging buffer) and the stage-parallel
    // phases read M and Gam of their knot from LDS instead of walking a global record
    static constexpr bool PG_LDS = ONE && !T::LTI && n <= 4;
    // ... and two numbers per knot from which f and A of the linearisation point follow without a sin / cos (Dyn::lin_cache):
    // the phases of an interior point iteration asked for them six times, ~140 instructions apiece in double precision
    static constexpr bool LC_LDS = ONE && MODEL == GUSTO_DUBINS_CAR;
    static constexpr int KDW = 2 * m * n + m * (m + 1) / 2;
    // ... and the slot of knot k first holds the stage cost QQ_k (NZ (NZ + 1) / 2 doubles): the residual phase writes it
    // there, factor stage k reads it and then overwrites the slot with K_k | D_k | S_k^-1 -- the factors of the previous
    // interior point iteration are dead by the time the next residual phase runs.  No QQ record in global memory at all.
    static constexpr int KDS = (KDW > NZ * (NZ + 1) / 2) ? KDW : NZ * (NZ + 1) / 2;
    static constexpr bool PHICL_LDS = n <= 8 && !PHI_FROM_K;
};
// The KKT solve as two Riccati segments joined by a coarse LQR stage (round 6; seg.hpp).
//   GUSTO_SEG2   (off): both chains interleaved in ONE wave, freeflyerSE2 -- parity-green and slower (profiles/r06_two_chains.txt)
//   GUSTO_SEG_W2 (on):  a WAVE PER CHAIN for the matrix-core kernels (astrobeeSE3, astrobeeSE3manifold): scp_kernel_w2, launched for
//                       batches that leave half of the SIMDs idle (launch.hpp: seg_w2_wanted)
#ifndef GUSTO_SEG2
#define GUSTO_SEG2 0
#endif
#ifndef GUSTO_SEG_W2
#ifdef GUSTO_STRICT_SYNC   // (the check build's ordering points are workgroup barriers: one wave per workgroup only)
#define GUSTO_SEG_W2 0
#else
#define GUSTO_SEG_W2 1
#endif
#endif
#define GUSTO_SEG_ANY (GUSTO_SEG2 || GUSTO_SEG_W2)
#ifndef GUSTO_SEG_MIN_N
#define GUSTO_SEG_MIN_N 4       // stages per chain at least (a chain of two or three stages is barely controllable)
#endif
__host__ __device__ constexpr int seg_split(int N) { return N >> 1; }   // two chains: A = stages 0 .. s-1, B = s .. N-1 (one stage longer for odd N)
template <int MODEL> constexpr bool seg2_big() { return GUSTO_SEG_W2 && MT<MODEL>::MFMA && MT<MODEL>::SWEEP_CALL && MT<MODEL>::NDEF == 0; }
// LDS block of the segmented solve of these kernels, behind everything else (offsets relative to LdsLayout::seg).  NCH chains
// (= waves per problem, 2 or 4): chain c covers the stages seg_lo(c) .. seg_lo(c + 1) - 1, interface j sits between chain j and
// what lies behind it as the merge tree pairs them (segw.hpp: seg_merge; four chains: interfaces 0 and 2 inside the pairs (C0 | C1),
// (C2 | C3), interface 1 between the pairs).
__host__ __device__ constexpr int seg_lo(int c, int N, int NCH) { return (int)((long)N * c / NCH); }
constexpr int SEGW_FACTOR = 1, SEGW_BACK = 2, SEGW_FWD = 3, SEGW_ROWS_R = 5, SEGW_STEP = 6, SEGW_STEP_CS = 7, SEGW_EXIT = 9;   // commands to the helper waves (segw.hpp)
constexpr int SEG_RP = 15;   // values per knot a wave leaves of its share of a row pass (segw.hpp: segw_rows_*)
template <int MODEL, int NCH> struct SegB {
    static constexpr int n = MT<MODEL>::n, m = MT<MODEL>::m, NNp = (n * n + 1) & ~1, NPG = n * (n + m), NI = NCH - 1;
    // per interface: Ta', Sig, Pa = Ta Pc, A2 = Sig Pic, A3 = Ta Pic and the folded rear part's (Pc, Pic) -- the last chain's own sweep
    // leaves its P, Pi in the last interface's pair, a fold writes the pair of the interface in front
    static constexpr int IFB = 7 * NNp, Tt = 0, Sg = NNp, Pa = 2 * NNp, A2 = 3 * NNp, A3 = 4 * NNp, Pc = 5 * NNp, PIc = 6 * NNp;
    static constexpr int IF(int j) { return j * IFB; }
    // per chain in front of the last: what its factor sweep leaves (P, Pi in front of its first stage, its Gd)
    static constexpr int CHB = 3 * NNp, Pf = 0, Pif = NNp, Gdf = 2 * NNp;
    static constexpr int CH(int c) { return NI * IFB + c * CHB; }
    static constexpr int COM = NI * (IFB + CHB), Gci = COM, A1 = COM + NNp, X1 = COM + 2 * NNp, X2 = COM + 3 * NNp;
    // n-vectors (16-double slots): interface state and costate increment per interface, the front costate offset of the chains c >= 1
    static constexpr int vec = COM + 4 * NNp;
    static constexpr int XI(int j) { return vec + 16 * j; }
    static constexpr int LAM(int j) { return vec + 16 * NI + 16 * j; }
    static constexpr int PBV(int c) { return vec + 32 * NI + 16 * (c - 1); }
    // a helper wave's block: its own [Phi Gam] double buffer and L^-1 scratch in the factor sweep (which runs beside the others'),
    // the partial sums of its share of the obstacle rows in the row passes (SEG_RP values per knot, [value][lane]); the mailbox
    static constexpr int hlp = vec + 48 * NI, HLB = (2 * NPG + 64 > SEG_RP * 64) ? 2 * NPG + 64 : SEG_RP * 64;
    static constexpr int sPG2(int h) { return hlp + h * HLB; }
    static constexpr int Lw2(int h) { return hlp + h * HLB + 2 * NPG; }
    static constexpr int MBX = hlp + NI * HLB, total = MBX + 24;
};
// the obstacles of wave `rank` when `nshare` waves share a knot's obstacle rows (by obstacle index: bit i of the knot's active mask)
__host__ __device__ constexpr unsigned long long seg_obs_share(int rank, int nshare) {
    unsigned long long p = 0;
    for (int i = rank; i < 64; i += (nshare > 0 ? nshare : 1)) p |= 1ull << i;
    return p;
}
// where a knot's costate (and its control's feed-forward) takes its multiplier from: the costate increment of the interface behind its
// chain, mu_g for the last chain (offset from the base of the dynamic LDS)
template <int MODEL, int NCH> GD int seg_mult_off(int k, int N, int seg_base) {
    using SB = SegB<MODEL, NCH>;
    int c = 0;
#pragma unroll
    for (int j = 1; j < NCH; j++) c += (k >= seg_lo(j, N, NCH)) ? 1 : 0;
    return (c < NCH - 1) ? seg_base + SB::LAM(0) + 16 * c : LdsC<MODEL, true>::misc + 48;
}

struct LdsLayout {
    int total;
    int phicl;  // offset of the LDS copy of Phicl (doubles), -1 if in the global workspace or rebuilt from K
    int kd;     // offset of K | D | S^-1 per knot in LDS (LdsC::KD_LDS), -1 if in the global workspace
    int pg;     // offset of [Phi Gam] per knot in LDS (LdsC::PG_LDS), -1 if in the global workspace
    int lc;     // offset of the linearisation cache (LdsC::LC_LDS: 2 doubles per knot), -1 if none
    int seg;    // offset of the segmented solve's block (SegB), -1 if none
};
template <int MODEL> inline LdsLayout make_lds_layout(int N, bool multi_wave = false, int seg_nch = 0) {
    using C1 = LdsC<MODEL, true>;
    using CM = LdsC<MODEL, false>;
    LdsLayout L;
    // (the TrajOpt variants run the multi-wave phases whatever N: their layout is the multi-wave one)
    const bool one = N <= 64 && MT<MODEL>::NDEF == 0 && !multi_wave;
    L.total = (one ? C1::vecs : CM::vecs) + N * (C1::NVN * C1::n + C1::NVM * C1::m);
    L.phicl = -1; L.kd = -1;
    if (C1::KD_LDS && one) { L.kd = L.total; L.total += N * C1::KDS; }
    if (C1::PHICL_LDS && one) { L.phicl = L.total; L.total += N * C1::n * C1::n; }
    L.pg = -1;
    if (C1::PG_LDS && one) { L.pg = L.total; L.total += N * C1::n * C1::NZ; }
    L.lc = -1;
    if (C1::LC_LDS && one) { L.lc = L.total; L.total += 2 * N; }
    L.seg = -1;
    if (seg2_big<MODEL>() && one && seg_nch > 0) { L.seg = L.total; L.total += (seg_nch == 4) ? SegB<MODEL, 4>::total : SegB<MODEL, 2>::total; }
    return L;
}

// kernel arguments
struct KParams {
    int N, B, n_obs, n_box, n_sph, hist_cap, max_iter, force, mode;  // mode 0: SCP solve, 1: one subproblem
    int probe_visits;   // longest-first schedule: a problem's first `probe_visits` time slices are ONE trip each (0: off)
    int slice_q;        // ... after which a problem of penalty level 0 goes on in slices of `slice_q` trips (0: runs to its end)
    int* queue;         // scheduler state of this launch (Sched below): persistent workgroups pull work with atomics
    int* lists;         // [SCHED_LEVELS][list_cap] problems waiting for their next slice, by penalty level; entries start
                        // at -1; an entry is (slices so far << 24) | problem
    int list_cap;       // entries per list: a problem is pushed at most once per finite slice
    const int* order;   // fresh problems are handed out in this order (hardest first, scp.hpp: sched_key_kernel); null = 0, 1, 2, ...
    int n_fresh;        // how many problems the launch hands out: B, or the number of active ones (gusto_set_active: `order` lists them)
    gusto_scp_params sp;
    gusto_model_params mp;
    gusto_ipm_opts io;
    const double* box;  // [n_box][6]   (gusto_set_env_batch: the tables of all problems, concatenated)
    const double* sph;  // [n_sph][4]
    const int* env;     // null: one keep-out set for the batch; else [B][4] = box offset, n_box, sphere offset, n_sph of problem b
                        // (n_obs is then the LARGEST count of the batch: it sizes the row slots and the obstacle arrays)
    double* X;          // [B][N][n]  SCPS.traj.X
    double* U;          // [B][N][m]
    const double *x_init, *goal_lo, *goal_hi, *tf;
    // subproblem mode
    const double *sub_Delta, *sub_omega, *sub_toggle;
    double *sub_X, *sub_U, *sub_obj;
    int *sub_status, *sub_iters;
    // SCP state and histories
    int* st_i;     // [B][8]: iterations, converged, successful, stop_reason, total_ipm, n_hist, nJ, n_rho
    double* st_d;  // [B][2+MAXN]: toggle, spare, dual[n]
    double *J_true, *J_full, *conv, *Delta, *omega, *rho;                       // [B][hist_cap]
    int *accept, *scp_status, *solver_status, *tr_sat, *cvx_sat, *ipm_it;      // [B][hist_cap]
    // TrajOpt (scp_trajopt.jl): parameters and the vectors GuSTO has no counterpart of; rho / Delta hold rho_vec / s_vec
    gusto_trajopt_params tp;
    double *to_mu, *to_xtol, *to_ftol, *to_ctol;                                // [B][hist_cap]
    double* ws;
    long long* prof;  // [B][PROF_N] phase cycle counters (GUSTO_PROFILE builds only)
    WsLayout wl;
    LdsLayout ll;
};
constexpr int ST_ITER = 0, ST_CONV = 1, ST_SUCC = 2, ST_STOP = 3, ST_IPM = 4, ST_NHIST = 5, ST_NJ = 6, ST_NRHO = 7, ST_WARM = 8,
              ST_CAP = 9 /* iter_cap of the running gusto_solve call */, ST_VISITS = 10 /* time slices so far */,
              ST_NMU = 11, ST_NXTOL = 12, ST_NFTOL = 13, ST_NCTOL = 14 /* TrajOpt: entries of mu_vec / xtol_vec / ftol_vec / ctol_vec */, ST_NI = 16;
// scheduler words in KParams::queue.  Every counter sits in its own 128-byte line (SQ_STRIDE ints apart): thousands of
// workgroups poll and bump them, and two counters in one line serialise each other's atomics in the L2.
constexpr int SCHED_LEVELS = 16, SQ_STRIDE = 32;
constexpr int SQ_HEAD_A = 0, SQ_PROBING = SQ_STRIDE /* problems that may still be pushed */, SQ_TAIL = 2 * SQ_STRIDE /* [level] */,
              SQ_HEAD = (2 + SCHED_LEVELS) * SQ_STRIDE /* [level] */, SQ_ERR = (2 + 2 * SCHED_LEVELS) * SQ_STRIDE /* scheduler gave up */,
              SQ_HI = (3 + 2 * SCHED_LEVELS) * SQ_STRIDE /* entries waiting in the lists of level >= 1 (a hint) */,
              SQ_WORDS = (4 + 2 * SCHED_LEVELS) * SQ_STRIDE;
constexpr int SD_TOGGLE = 0, SD_DUAL = 2, SD_ND = 2 + GUSTO_MAXN;

// ---- optional phase timers (compile with -DGUSTO_PROFILE) ---------------------------------------------
constexpr int PROF_N = 48;
enum { PF_RESID = 0, PF_BUILD, PF_FACTOR, PF_POSTF, PF_RHS, PF_BACK, PF_MID, PF_FWD, PF_STEP, PF_UPDATE, PF_LIN, PF_SCP, PF_INIT,
       PF_FPRE, PF_FAB, PF_FCD, PF_F1, PF_F2, PF_F3, PF_F4, PF_F5, PF_F6, PF_F7, PF_F8,
       PF_M_TH /* mid phase: theta */, PF_M_RED, PF_M_MU, PF_M_DK, PF_M_SYNC,
       PF_R0 = 32 /* residual pass: prologue | fixed rows | obstacle rows | control rows (the rest of the pass: PF_RESID) */,
       PF_S0 = 36 /* step pass, likewise (the rest: PF_STEP, PF_F1) */ };
struct Prof {
#ifdef GUSTO_PROFILE
    long long t0, acc[PROF_N];
    GD Prof() { for (int i = 0; i < PROF_N; i++) acc[i] = 0; t0 = clock64(); }
#ifdef GUSTO_PROFILE_COARSE   // no stamp inside a factor stage or a row pass: a stamp waits for what is in flight (~150 cycles each)
    GD void tick(int id) { if ((id >= PF_FPRE && id <= PF_F8) || id >= PF_R0) return; const long long t = clock64(); acc[id] += t - t0; t0 = t; }
#else
    GD void tick(int id) { const long long t = clock64(); acc[id] += t - t0; t0 = t; }
#endif
    // (a problem runs in several time slices under the scheduler: the first one overwrites, the others accumulate)
    GD void flush(long long* out, int b, bool cont = false) {
        if (out && threadIdx.x == 0)
            for (int i = 0; i < PROF_N; i++)
                if (i < 29 || i > 31) out[(size_t)b * PROF_N + i] = (cont ? out[(size_t)b * PROF_N + i] : 0) + acc[i];   // (29 .. 31: a helper wave's, segw.hpp)
    }
#else
    GD void tick(int) {}
    GD void flush(long long*, int, bool = false) {}
#endif
};

// ---- block-wide reductions (every thread of the block must call) -----------------------------------
struct OpMax { GD double operator()(double a, double b) const { return fmax(a, b); } };
struct OpMin { GD double operator()(double a, double b) const { return fmin(a, b); } };
struct OpSum { GD double operator()(double a, double b) const { return a + b; } };

GD double readlane_f64(double v, int lane) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffu), lane);
    const unsigned hi = __builtin_amdgcn_readlane((int)(u >> 32), lane);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// lane i <- lane (i - N) mod 16 within its row of 16 lanes (DPP row_ror): a VALU move, no trip through the LDS crossbar
template <int N> GD double row_ror_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = (int)(u & 0xffffffffu), hi = (int)(u >> 32);
    const unsigned rl = (unsigned)__builtin_amdgcn_update_dpp(lo, lo, 0x120 + N, 0xf, 0xf, false);
    const unsigned rh = (unsigned)__builtin_amdgcn_update_dpp(hi, hi, 0x120 + N, 0xf, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)rh << 32) | rl);
}
// wave-wide reduction, the same value in every lane: four DPP rotate steps inside each row of 16 lanes, then the four
// row results through v_readlane (ds_bpermute based __shfl_xor steps cost an LDS round trip each: ~600 cycles per
// reduction, and an interior point iteration makes two dozen of them)
template <class Op> GD double wave_reduce(double v, Op op) {
    v = op(v, row_ror_f64<1>(v));
    v = op(v, row_ror_f64<2>(v));
    v = op(v, row_ror_f64<4>(v));
    v = op(v, row_ror_f64<8>(v));
    const double r0 = readlane_f64(v, 0), r1 = readlane_f64(v, 16), r2 = readlane_f64(v, 32), r3 = readlane_f64(v, 48);
    return op(op(r0, r1), op(r2, r3));
}
// n reductions at once, step by step across all of them: each DPP step of one value waits ~3 dependent instructions for the
// one before it, and written one reduction after the other hipcc keeps that order (6-13 reductions back to back in the mid
// phase); interleaved, a step of one value issues in the shadow of the others'.  Same operations per value.
template <int NV, class Op> GD void wave_reduce_n(double* v, Op op) {
#pragma unroll
    for (int j = 0; j < NV; j++) v[j] = op(v[j], row_ror_f64<1>(v[j]));
#pragma unroll
    for (int j = 0; j < NV; j++) v[j] = op(v[j], row_ror_f64<2>(v[j]));
#pragma unroll
    for (int j = 0; j < NV; j++) v[j] = op(v[j], row_ror_f64<4>(v[j]));
#pragma unroll
    for (int j = 0; j < NV; j++) v[j] = op(v[j], row_ror_f64<8>(v[j]));
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const double r0 = readlane_f64(v[j], 0), r1 = readlane_f64(v[j], 16), r2 = readlane_f64(v[j], 32), r3 = readlane_f64(v[j], 48);
        v[j] = op(op(r0, r1), op(r2, r3));
    }
}
// lane i <- lane i + N of its row of 16 lanes, 0 past the row's end (DPP row_shl, bound_ctrl): the step of a suffix scan
template <int N> GD double row_shl0_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = (int)(u & 0xffffffffu), hi = (int)(u >> 32);
    const unsigned rl = (unsigned)__builtin_amdgcn_update_dpp(0, lo, 0x100 + N, 0xf, 0xf, true);
    const unsigned rh = (unsigned)__builtin_amdgcn_update_dpp(0, hi, 0x100 + N, 0xf, 0xf, true);
    return __builtin_bit_cast(double, ((unsigned long long)rh << 32) | rl);
}
// NV inclusive SUFFIX sums over the 64 lanes at once, v[j] of lane i <- sum of v[j] over the lanes >= i: four DPP steps inside each
// row of 16 lanes (interleaved across the values, see wave_reduce_n), then the totals of the rows behind through v_readlane
template <int NV> GD void wave_suffix_sum_n(double* v) {
#pragma unroll
    for (int j = 0; j < NV; j++) v[j] += row_shl0_f64<1>(v[j]);
#pragma unroll
    for (int j = 0; j < NV; j++) v[j] += row_shl0_f64<2>(v[j]);
#pragma unroll
    for (int j = 0; j < NV; j++) v[j] += row_shl0_f64<4>(v[j]);
#pragma unroll
    for (int j = 0; j < NV; j++) v[j] += row_shl0_f64<8>(v[j]);
    const int row = (threadIdx.x & 63) >> 4;
#pragma unroll
    for (int j = 0; j < NV; j++) {
        const double t1 = readlane_f64(v[j], 16), t2 = readlane_f64(v[j], 32), t3 = readlane_f64(v[j], 48);
        const double behind = (row == 0) ? (t1 + t2) + t3 : ((row == 1) ? t2 + t3 : ((row == 2) ? t3 : 0.0));
        v[j] += behind;
    }
}
// the value of the next lane (lane 63: 0): inside a row by DPP, across a row boundary through v_readlane
GD double wave_next_f64(double v) {
    const double in_row = row_shl0_f64<1>(v);
    const double t1 = readlane_f64(v, 16), t2 = readlane_f64(v, 32), t3 = readlane_f64(v, 48);
    const int l = threadIdx.x & 63;
    return (l == 15) ? t1 : ((l == 31) ? t2 : ((l == 47) ? t3 : in_row));
}
// the value of the previous lane (lane 0: 0)
GD double wave_prev_f64(double v) {
    const unsigned long long u = __builtin_bit_cast(unsigned long long, v);
    const int lo = (int)(u & 0xffffffffu), hi = (int)(u >> 32);
    const unsigned rl = (unsigned)__builtin_amdgcn_update_dpp(0, lo, 0x111, 0xf, 0xf, true);   // row_shr:1
    const unsigned rh = (unsigned)__builtin_amdgcn_update_dpp(0, hi, 0x111, 0xf, 0xf, true);
    const double in_row = __builtin_bit_cast(double, ((unsigned long long)rh << 32) | rl);
    const double t0 = readlane_f64(v, 15), t1 = readlane_f64(v, 31), t2 = readlane_f64(v, 47);
    const int l = threadIdx.x & 63;
    return (l == 16) ? t0 : ((l == 32) ? t1 : ((l == 48) ? t2 : in_row));
}
// NaN-propagating max: used for residuals so that a NaN iterate is detected
GD double nanmax(double a, double b) { return (a != a || b != b) ? NAN : fmax(a, b); }
struct OpNanMax { GD double operator()(double a, double b) const { return nanmax(a, b); } };

// ONE: the caller's kernel is a one-wave kernel (Blk::ONE).  Without it the test below reads blockDim from the dispatch
// packet -- inside the phases that are real calls (MT::SWEEP_CALL) on every call: an s_load, a global_load_ushort and an
// s_waitcnt vmcnt(0) that drains every load in flight, ~830 cycles per reduction, two dozen reductions per KKT solve.
template <bool ONE = false, class Op> GD double block_reduce(double v, Op op, double* sred) {
    v = wave_reduce(v, op);
    if constexpr (ONE) return v;
    if (blockDim.x <= 64) return v;
    const int w = threadIdx.x >> 6, nw = blockDim.x >> 6;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) sred[w] = v;
    __syncthreads();
    double r = sred[0];
    for (int i = 1; i < nw; i++) r = op(r, sred[i]);
    return r;
}

// ---- small dense helpers on register arrays (fully unrolled) ---------------------------------------
template <int n> GD bool inv_gauss_jordan(const double* A, double* Ainv) {
    double W[n][2 * n];
#pragma unroll
    for (int i = 0; i < n; i++)
#pragma unroll
        for (int j = 0; j < n; j++) { W[i][j] = A[i * n + j]; W[i][n + j] = (i == j) ? 1.0 : 0.0; }
    bool ok = true;
#pragma unroll
    for (int c = 0; c < n; c++) {
        // partial pivoting by conditional row swap (branch-free on register arrays)
        int piv = c;
        double best = fabs(W[c][c]);
#pragma unroll
        for (int r = c + 1; r < n; r++) {
            const double v = fabs(W[r][c]);
            if (v > best) { best = v; piv = r; }
        }
        if (best == 0.0) ok = false;
#pragma unroll
        for (int r = c + 1; r < n; r++) {
            if (r == piv) {
#pragma unroll
                for (int j = 0; j < 2 * n; j++) { const double t = W[c][j]; W[c][j] = W[r][j]; W[r][j] = t; }
            }
        }
        const double d = 1.0 / W[c][c];
#pragma unroll
        for (int j = 0; j < 2 * n; j++) W[c][j] *= d;
#pragma unroll
        for (int r = 0; r < n; r++) {
            if (r != c) {
                const double f = W[r][c];
#pragma unroll
                for (int j = 0; j < 2 * n; j++) W[r][j] -= f * W[c][j];
            }
        }
    }
#pragma unroll
    for (int i = 0; i < n; i++)
#pragma unroll
        for (int j = 0; j < n; j++) Ainv[i * n + j] = W[i][n + j];
    return ok;
}

// 1/sqrt(d) to double precision: v_rsq_f64 seed (~2^-26) + two Newton-Raphson steps, ~14 dependent flops
// instead of the ~40 of sqrt() followed by a division -- the Cholesky of the m x m block sits on the critical
// path of every knot of the factor sweep.
// Marks a pointer that crossed a function call as pointing to global memory (generic -> addrspace(1) -> generic): the
// accesses through it compile to global_load/global_store instead of flat_load/flat_store.
template <class Tp> GD Tp* as_global(Tp* p) {
    typedef __attribute__((address_space(1))) Tp G;
    return (Tp*)(G*)p;
}
// Pointer members that remember their address space.  A struct that is passed to a (non-inlined) phase function, or whose
// address escapes, lives in memory; a plain pointer reloaded from it is GENERIC to the compiler and every access through
// it becomes flat_load / flat_store -- which also counts against lgkmcnt, so each wait for an LDS read waits for the
// global loads in flight as well.  The 12/13-state kernels had 4 200 flat and 146 global memory instructions that way.
// These wrappers keep the pointer TYPED with its address space (a cast pair generic -> global -> generic at the point
// of use is folded away before the address space inference runs).
#ifdef GUSTO_DEBUG_LDS
// Debug builds (-DGUSTO_DEBUG_LDS, tools/debug_lds.sh): every indexed LDS access through an LPtr is checked against the
// workgroup's LDS allocation (static + dynamic bytes, published by the kernel in gusto_dbg_lds_limit); an access past it
// names itself and traps instead of silently reading a neighbour's memory -- the sanitizer this single-source kernel
// family can have ("debug-build LDS index asserts").
__device__ inline unsigned& gusto_dbg_lds_limit() { static __shared__ unsigned lim; return lim; }
#endif
template <class Tp, int AS> struct ASPtr {
    typedef __attribute__((address_space(AS))) Tp A;
    A* p;
    ASPtr() = default;
    GD ASPtr(Tp* q) : p((A*)q) {}
    template <class U> GD ASPtr(const ASPtr<U, AS>& o) : p(o.p) {}
    template <class I> GD A& operator[](I i) const {
#ifdef GUSTO_DEBUG_LDS
        if constexpr (AS == 3) {
            const unsigned at = (unsigned)(uintptr_t)(p + i);
            if (at + sizeof(Tp) > gusto_dbg_lds_limit()) {
                printf("gusto: LDS access out of bounds: byte %u + %u of %u (block %d thread %d)\n", at, (unsigned)sizeof(Tp),
                       gusto_dbg_lds_limit(), (int)blockIdx.x, (int)threadIdx.x);
                __builtin_trap();
            }
        }
#endif
        return p[i];
    }
    template <class I> GD ASPtr operator+(I i) const { ASPtr r; r.p = p + i; return r; }
    GD A& operator*() const { return *p; }
    GD operator Tp*() const { return (Tp*)p; }   // (generic again: for the few callers that take a plain pointer)
};
template <class Tp> using GPtr = ASPtr<Tp, 1>;   // global memory
template <class Tp> using LPtr = ASPtr<Tp, 3>;   // LDS
// The per-knot records of the workspace start on 16-byte boundaries (WsLayout: even offsets, 64-double strides): told
// so, the compiler merges the loads of two adjacent entries into one global_load_dwordx4 -- half the instructions for
// the record walks of the stage-parallel phases, each of which touches one cache line per lane.
template <class Tp> GD Tp* al16(Tp* p) { return static_cast<Tp*>(__builtin_assume_aligned(p, 16)); }

// 1/d from the hardware seed + two Newton steps (the inner part of the IEEE division sequence, without its scaling
// and final correction: <= 2 ulp for the normal-range, finite operands of the row algebra): ~6 instructions against
// ~14 for `a / b`.  The row algebra of one knot holds ~250 divisions per interior point iteration.
GD double rcp_nr(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    return r;
}

GD double rsqrt_nr(double d) {
    double r = __builtin_amdgcn_rsq(d);
    const double h = 0.5 * d;
    r = r * (1.5 - h * r * r);
    r = r * (1.5 - h * r * r);
    return r;
}

// Cholesky S = L L^T on register arrays: returns Li = L^-1 (lower, row-major m x m); false if not PD.
template <int m> GD bool chol_inv(const double* S, double* Li) {
    double L[m][m], r[m];
    bool ok = true;
#pragma unroll
    for (int i = 0; i < m; i++)
#pragma unroll
        for (int j = 0; j < m; j++) { L[i][j] = 0; Li[i * m + j] = 0; }
#pragma unroll
    for (int j = 0; j < m; j++) {
        double d = S[j * m + j];
#pragma unroll
        for (int l = 0; l < j; l++) d -= L[j][l] * L[j][l];
        if (!(d > 0.0)) ok = false;
        r[j] = rsqrt_nr(d);
        L[j][j] = d * r[j];
#pragma unroll
        for (int i = j + 1; i < m; i++) {
            double s = S[i * m + j];
#pragma unroll
            for (int l = 0; l < j; l++) s -= L[i][l] * L[j][l];
            L[i][j] = s * r[j];
        }
    }
#pragma unroll
    for (int j = 0; j < m; j++) {
        Li[j * m + j] = r[j];
#pragma unroll
        for (int i = j + 1; i < m; i++) {
            double s = 0;
#pragma unroll
            for (int l = j; l < i; l++) s -= L[i][l] * Li[l * m + j];
            Li[i * m + j] = s * r[i];
        }
    }
    return ok;
}

// runtime-size SPD inverse in memory (LDS), single thread; used for the ng x ng goal system
GD bool inv_spd_rt(const double* S, double* Sinv, double* Lw, int n) {
    bool ok = true;
    for (int i = 0; i < n * n; i++) { Lw[i] = 0; Sinv[i] = 0; }
    for (int j = 0; j < n; j++) {
        double d = S[j * n + j];
        for (int l = 0; l < j; l++) d -= Lw[j * n + l] * Lw[j * n + l];
        if (!(d > 0.0)) ok = false;
        d = sqrt(d);
        Lw[j * n + j] = d;
        for (int i = j + 1; i < n; i++) {
            double s = S[i * n + j];
            for (int l = 0; l < j; l++) s -= Lw[i * n + l] * Lw[j * n + l];
            Lw[i * n + j] = s / d;
        }
    }
    // Sinv <- L^{-1} (lower), then Lw <- L^{-T} L^{-1}
    for (int j = 0; j < n; j++) {
        Sinv[j * n + j] = 1.0 / Lw[j * n + j];
        for (int i = j + 1; i < n; i++) {
            double s = 0;
            for (int l = j; l < i; l++) s -= Lw[i * n + l] * Sinv[l * n + j];
            Sinv[i * n + j] = s / Lw[i * n + i];
        }
    }
    for (int i = 0; i < n; i++)
        for (int j = 0; j < n; j++) {
            double s = 0;
            for (int l = (i > j ? i : j); l < n; l++) s += Sinv[l * n + i] * Sinv[l * n + j];
            Lw[i * n + j] = s;
        }
    for (int i = 0; i < n * n; i++) Sinv[i] = Lw[i];
    return ok;
}

}  // namespace gusto
