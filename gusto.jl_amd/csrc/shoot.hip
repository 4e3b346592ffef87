// shoot.hip -- batched indirect shooting seeded by the SCP dual (src/shooting.jl:4-66, src/traj_opt.jl:4-45).
// One THREAD per problem: the state + costate ODE of a problem is 2n-dimensional (6 for DubinsCar, 26 for
// AstrobeeSE3Manifold) and its Newton system n x n, so a problem is register / scratch work for one lane and a batch of
// thousands fills the GPU with independent lanes.
//   DubinsCar:            shooting_ode! / get_control   src/dynamics/dubins_car.jl:259-280
//   AstrobeeSE3Manifold:  dynamics_shooting! / shooting_ode! / get_control   src/dynamics/astrobee_se3_manifold.jl:831-895
// The reference integrates with DifferentialEquations' default adaptive method and solves F(p0) = x_goal - x(tf; p0) = 0
// with NLsolve (trust region, finite-difference Jacobian, ftol = 1e-3, <= 100 iterations).  Neither exists here; the
// scheme is stated instead (the test suite's CPU restatement follows the same one): classical RK4 with `substeps`
// steps per knot interval, Newton with a forward-difference Jacobian (h_j = 1e-6 max(1, |p_j|)), halving line search
// on |F|_inf; the Newton step by Cramer's rule (n = 3) or rank-revealing Gaussian elimination with COMPLETE pivoting
// (n = 13: dF/dp0 is rank deficient by one, newton_step below).  The two models above are the ones that have a shooting
// ODE in the reference.
// Two passes (gusto_shoot): a lane per problem for the first Newton iterations, then the problems still running -- the ones
// that do not converge and would hold their wave for a hundred iterations -- with a group of lanes each, one lane per
// Jacobian column / line-search candidate (shoot_group_kernel).
// Stores: a lane owns a problem, so writing its knots straight into X[b][k][i] would touch one cache line per lane per
// store (stride N n between lanes).  The recovered trajectory is written knot-major, Xt[k][i][b] -- consecutive lanes,
// consecutive addresses -- and a tiled transpose (shoot_transpose_kernel) produces the [b][k][i] layout of the C ABI.
#include <hip/hip_runtime.h>

#include "handle.hpp"

using namespace gusto;

namespace {

struct ShootParams {
    int B, N, substeps, max_newton;
    double ftol, v, k, mass, J[3];
    const double *x_init, *goal_lo, *goal_hi, *tf, *p0;   // p0 [B][n]
    double *X, *U, *p_out, *resid;                        // knot-major staging of the trajectories: X [N][n][B], U [N][m][B]
    int *status, *iters;
    int cap;            // Newton iterations of the lane-per-problem pass; a problem still running then goes to `list`
    int *list, *count;  // problems handed to the group pass (shoot_group_kernel), their number
    const int* active;  // gusto_set_active: null = every problem, else the mask [B] (an inactive problem is reported :Diverged, untouched)
};

template <int MODEL> struct ShootModel;
// DubinsCar: shooting_ode! / get_control, dubins_car.jl:259-280
template <> struct ShootModel<GUSTO_DUBINS_CAR> {
    static constexpr int n = 3, m = 1;
    __device__ __forceinline__ static void ctrl(const ShootParams& S, const double* z, double* u) { u[0] = 0.5 * S.k * z[5]; }
    __device__ __forceinline__ static void rhs(const ShootParams& S, const double* z, double* dz) {
        const double u = 0.5 * S.k * z[5];
        double sn, cs;
        sincos(z[2], &sn, &cs);
        dz[0] = S.v * cs; dz[1] = S.v * sn; dz[2] = S.k * u;
        dz[3] = 0.0; dz[4] = 0.0; dz[5] = z[3] * S.v * sn - z[4] * S.v * cs;
    }
};
// AstrobeeSE3Manifold: dynamics_shooting! / shooting_ode! / get_control, astrobee_se3_manifold.jl:831-895 (the row
// contributions of :897-1006 are commented out in shooting_ode! at HEAD).  z = (r v q w | pr pv pq pw).
template <> struct ShootModel<GUSTO_ASTROBEE_SE3_MANIFOLD> {
    static constexpr int n = 13, m = 6;
    __device__ __forceinline__ static void ctrl(const ShootParams& S, const double* z, double* u) {
#pragma unroll
        for (int i = 0; i < 3; i++) { u[i] = z[13 + 3 + i] / (2.0 * S.mass); u[3 + i] = z[13 + 10 + i] / S.J[i] / 2.0; }
    }
    __device__ __forceinline__ static void rhs(const ShootParams& S, const double* z, double* dz) {
        const double qw = z[6], qx = z[7], qy = z[8], qz = z[9], wx = z[10], wy = z[11], wz = z[12];
        const double pqw = z[19], pqx = z[20], pqy = z[21], pqz = z[22];
        double u[6];
        ctrl(S, z, u);
#pragma unroll
        for (int i = 0; i < 3; i++) { dz[i] = z[3 + i]; dz[3 + i] = u[i] / S.mass; }
        dz[6] = 0.5 * (-wx * qx - wy * qy - wz * qz);
        dz[7] = 0.5 * (wx * qw - wz * qy + wy * qz);
        dz[8] = 0.5 * (wy * qw + wz * qx - wx * qz);
        dz[9] = 0.5 * (wz * qw - wy * qx + wx * qy);
        const double Jw[3] = {S.J[0] * wx, S.J[1] * wy, S.J[2] * wz};
        const double c[3] = {wy * Jw[2] - wz * Jw[1], wz * Jw[0] - wx * Jw[2], wx * Jw[1] - wy * Jw[0]};
        dz[10] = (u[3] - c[0]) / S.J[0]; dz[11] = (u[4] - c[1]) / S.J[1]; dz[12] = (u[5] - c[2]) / S.J[2];
        dz[13] = 0; dz[14] = 0; dz[15] = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) dz[16 + i] = -z[13 + i];
        dz[19] = -0.5 * (pqx * wx + pqy * wy + pqz * wz);
        dz[20] = -0.5 * (-pqw * wx + pqy * wz - pqz * wy);
        dz[21] = -0.5 * (-pqw * wy - pqx * wz + pqz * wx);
        dz[22] = -0.5 * (-pqw * wz + pqx * wy - pqy * wx);
        dz[23] = -0.5 * (-pqw * qx + pqx * qw - pqy * qz + pqz * qy);
        dz[24] = -0.5 * (-pqw * qy + pqx * qz + pqy * qw - pqz * qx);
        dz[25] = -0.5 * (-pqw * qz - pqx * qy + pqy * qx + pqz * qw);
    }
};

// integrates from (x_init, p0); writes the knots of problem b (knot-major, coalesced over the lanes) when X != nullptr;
// returns x(tf) in xT
template <int MODEL>
__device__ void integrate(const ShootParams& S, const double* x0, const double* p0, double tf, double* xT, double* X, double* U, int b = 0) {
    using M = ShootModel<MODEL>;
    constexpr int n = M::n, m = M::m, nz = 2 * n;
    double z[nz], k1[nz], k2[nz], k3[nz], k4[nz], w[nz];
    const double h = tf / ((S.N - 1) * (double)S.substeps);
#pragma unroll
    for (int i = 0; i < n; i++) { z[i] = x0[i]; z[n + i] = p0[i]; }
    for (int k = 0; k < S.N; k++) {
        if (X) {
#pragma unroll
            for (int i = 0; i < n; i++) X[(size_t)(k * n + i) * S.B + b] = z[i];
            double u[m];
            M::ctrl(S, z, u);                               // get_control
#pragma unroll
            for (int i = 0; i < m; i++) U[(size_t)(k * m + i) * S.B + b] = u[i];
        }
        if (k == S.N - 1) break;
        for (int s = 0; s < S.substeps; s++) {
            M::rhs(S, z, k1);
#pragma unroll
            for (int i = 0; i < nz; i++) w[i] = z[i] + 0.5 * h * k1[i];
            M::rhs(S, w, k2);
#pragma unroll
            for (int i = 0; i < nz; i++) w[i] = z[i] + 0.5 * h * k2[i];
            M::rhs(S, w, k3);
#pragma unroll
            for (int i = 0; i < nz; i++) w[i] = z[i] + h * k3[i];
            M::rhs(S, w, k4);
#pragma unroll
            for (int i = 0; i < nz; i++) z[i] += h / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < n; i++) xT[i] = z[i];
}

// dp = -J^-1 F: Cramer's rule for n = 3, Gaussian elimination with partial pivoting otherwise (J is overwritten)
template <int n> __device__ bool newton_step(double* J, const double* F, double* dp) {
    if constexpr (n == 3) {
        const double det = J[0] * (J[4] * J[8] - J[5] * J[7]) - J[1] * (J[3] * J[8] - J[5] * J[6]) + J[2] * (J[3] * J[7] - J[4] * J[6]);
        if (!(fabs(det) > 1e-300) || !isfinite(det)) return false;
        dp[0] = -(F[0] * (J[4] * J[8] - J[5] * J[7]) - J[1] * (F[1] * J[8] - J[5] * F[2]) + J[2] * (F[1] * J[7] - J[4] * F[2])) / det;
        dp[1] = -(J[0] * (F[1] * J[8] - J[5] * F[2]) - F[0] * (J[3] * J[8] - J[5] * J[6]) + J[2] * (J[3] * F[2] - F[1] * J[6])) / det;
        dp[2] = -(J[0] * (J[4] * F[2] - F[1] * J[7]) - J[1] * (J[3] * F[2] - F[1] * J[6]) + F[0] * (J[3] * J[7] - J[4] * J[6])) / det;
        return true;
    } else {
        // Gaussian elimination with COMPLETE pivoting, stopped at the numerical rank (pivots below 1e-6 max|J|, the
        // accuracy of the forward-difference Jacobian, are noise): the costate of the quaternion along q itself does not
        // move the state, dF/dp0 is rank deficient by one, and that component of the step stays 0 (see the oracle)
        double r[n], y[n], jmax = 0;
        int perm[n], rank = 0;
        for (int i = 0; i < n; i++) { r[i] = -F[i]; perm[i] = i; dp[i] = 0.0; }
        for (int i = 0; i < n * n; i++) jmax = fmax(jmax, fabs(J[i]));
        if (!(jmax > 0) || !isfinite(jmax)) return false;
        for (int c = 0; c < n; c++) {
            int pi = c, pj = c;
            double best = -1.0;
            for (int i = c; i < n; i++)
                for (int j = c; j < n; j++) {
                    const double a = fabs(J[i * n + j]);
                    if (!(a == a)) return false;
                    if (a > best) { best = a; pi = i; pj = j; }
                }
            if (!(best > 1e-6 * jmax)) break;
            if (pi != c) {
                for (int j = 0; j < n; j++) { const double t = J[c * n + j]; J[c * n + j] = J[pi * n + j]; J[pi * n + j] = t; }
                const double t = r[c]; r[c] = r[pi]; r[pi] = t;
            }
            if (pj != c) {
                for (int i = 0; i < n; i++) { const double t = J[i * n + c]; J[i * n + c] = J[i * n + pj]; J[i * n + pj] = t; }
                const int t = perm[c]; perm[c] = perm[pj]; perm[pj] = t;
            }
            for (int i = c + 1; i < n; i++) {
                const double f = J[i * n + c] / J[c * n + c];
                for (int j = c; j < n; j++) J[i * n + j] -= f * J[c * n + j];
                r[i] -= f * r[c];
            }
            rank = c + 1;
        }
        if (rank == 0) return false;
        for (int c = rank - 1; c >= 0; c--) {
            double sacc = r[c];
            for (int j = c + 1; j < rank; j++) sacc -= J[c * n + j] * y[j];
            y[c] = sacc / J[c * n + c];
        }
        for (int c = 0; c < rank; c++) dp[perm[c]] = y[c];
        return true;
    }
}

template <int MODEL> __global__ void __launch_bounds__(64) shoot_kernel(const ShootParams S) {
    using M = ShootModel<MODEL>;
    constexpr int n = M::n, m = M::m;
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.B) return;
    if (S.active && !S.active[b]) {
        S.status[b] = 0; S.iters[b] = 0; S.resid[b] = NAN;
        for (int i = 0; i < n; i++) S.p_out[(size_t)b * n + i] = S.p0[(size_t)b * n + i];
        return;
    }
    double x0[n], pv[n], xg[n], F[n], xT[n], nf = 0.0;
    const double tf = S.tf[b];
#pragma unroll
    for (int i = 0; i < n; i++) {
        x0[i] = S.x_init[(size_t)b * n + i];
        pv[i] = S.p0[(size_t)b * n + i];
        const double lo = S.goal_lo[(size_t)b * n + i], hi = S.goal_hi[(size_t)b * n + i];
        xg[i] = (isfinite(lo) && isfinite(hi)) ? 0.5 * (lo + hi) : 0.0;     // ShootingProblem ctor, types.jl:219-226
    }
    integrate<MODEL>(S, x0, pv, tf, xT, nullptr, nullptr);
#pragma unroll
    for (int i = 0; i < n; i++) { F[i] = xg[i] - xT[i]; nf = (F[i] != F[i] || nf != nf) ? NAN : fmax(nf, fabs(F[i])); }   // (fmax drops a NaN)
    int it = 0, ok = 0;
    bool later = false;
    for (;; it++) {
        if (!(nf == nf) || !isfinite(nf)) break;
        if (nf <= S.ftol) { ok = 1; break; }
        if (it >= S.max_newton) break;
        if (it >= S.cap) { later = true; break; }
        double J[n * n], pj[n];
        for (int j = 0; j < n; j++) {
            const double h = 1e-6 * fmax(1.0, fabs(pv[j]));
            for (int i = 0; i < n; i++) pj[i] = (i == j) ? pv[i] + h : pv[i];
            integrate<MODEL>(S, x0, pj, tf, xT, nullptr, nullptr);
            for (int i = 0; i < n; i++) J[i * n + j] = ((xg[i] - xT[i]) - F[i]) / h;
        }
        double dp[n];
        if (!newton_step<n>(J, F, dp)) break;
        double a = 1.0, nn = 0.0, Fn[n], pn[n];
        bool dec = false;
        while (a > 1e-4) {
            for (int i = 0; i < n; i++) pn[i] = pv[i] + a * dp[i];
            integrate<MODEL>(S, x0, pn, tf, xT, nullptr, nullptr);
            nn = 0.0;
            for (int i = 0; i < n; i++) { Fn[i] = xg[i] - xT[i]; nn = (Fn[i] != Fn[i] || nn != nn) ? NAN : fmax(nn, fabs(Fn[i])); }
            if (nn < nf) { dec = true; break; }
            a *= 0.5;
        }
        if (!dec) break;
        for (int i = 0; i < n; i++) { pv[i] = pn[i]; F[i] = Fn[i]; }
        nf = nn;
    }
    S.status[b] = ok; S.iters[b] = it; S.resid[b] = nf;
    for (int i = 0; i < n; i++) S.p_out[(size_t)b * n + i] = pv[i];
    if (later) S.list[atomicAdd(S.count, 1)] = b;   // (its state: p_out, iters)
    if (ok) integrate<MODEL>(S, x0, pv, tf, xT, S.X, S.U, b);
}

// The problems the lane-per-problem pass did not finish within S.cap Newton iterations -- for dubins_car the ones that do
// not converge at all and run their 100 iterations, each a Jacobian of n integrations and a halving line search of up to
// 14 -- continue here with G lanes each: a lane of the group integrates ONE perturbed start (a column of the Jacobian) or
// ONE candidate of the line search, so an iteration is two to three integrations long instead of up to n + 14.  In the
// first pass such a problem held its whole wave for ~1 700 integrations (measured: 370 ms for B = 65 536, 10 % of the lanes
// active); here the few thousand of them fill the GPU by themselves.  The same arithmetic per integration, the same
// acceptance rule (the first candidate of the halving sequence that decreases |F|): the same results.
template <int MODEL, int G> __global__ void __launch_bounds__(64) shoot_group_kernel(const ShootParams S, int count) {
    using M = ShootModel<MODEL>;
    constexpr int n = M::n, PG = 64 / G, NC = 14;   // candidates a = 2^-c > 1e-4: c = 0 .. 13
    static_assert(G >= n + 1 && 64 % G == 0, "a lane per Jacobian column");
    __shared__ double ex[64 * (2 * n + 1)];
    const int lane = threadIdx.x, g = lane / G, r = lane % G;
    const int idx = blockIdx.x * PG + g;
    const bool have = idx < count;
    const int b = S.list[have ? idx : 0];
    double x0[n], pv[n], xg[n], F[n], xT[n], nf = 0.0;
    const double tf = S.tf[b];
#pragma unroll
    for (int i = 0; i < n; i++) {
        x0[i] = S.x_init[(size_t)b * n + i];
        pv[i] = S.p_out[(size_t)b * n + i];
        const double lo = S.goal_lo[(size_t)b * n + i], hi = S.goal_hi[(size_t)b * n + i];
        xg[i] = (isfinite(lo) && isfinite(hi)) ? 0.5 * (lo + hi) : 0.0;
    }
    integrate<MODEL>(S, x0, pv, tf, xT, nullptr, nullptr);   // (F at the saved iterate: the values the first pass had)
#pragma unroll
    for (int i = 0; i < n; i++) { F[i] = xg[i] - xT[i]; nf = (F[i] != F[i] || nf != nf) ? NAN : fmax(nf, fabs(F[i])); }
    int it = S.iters[b], ok = 0;
    bool done = !have;
    for (;;) {   // (every lane of the wave runs every trip: the barriers below are wave-wide; a finished group idles)
        if (!done) {
            if (!(nf == nf) || !isfinite(nf)) done = true;
            else if (nf <= S.ftol) { ok = 1; done = true; }
            else if (it >= S.max_newton) done = true;
        }
        if (!__any(!done)) break;
        // Jacobian: lane r = 1 .. n integrates the start perturbed in component r - 1
        {
            const int j = (r >= 1 && r <= n) ? r - 1 : 0;
            double pj[n];
            double h = 0;
#pragma unroll
            for (int i = 0; i < n; i++) { const double hh = 1e-6 * fmax(1.0, fabs(pv[i])); pj[i] = (i == j) ? pv[i] + hh : pv[i]; h = (i == j) ? hh : h; }
            if (!done && r >= 1 && r <= n) integrate<MODEL>(S, x0, pj, tf, xT, nullptr, nullptr);
#pragma unroll
            for (int i = 0; i < n; i++) ex[lane * (2 * n + 1) + i] = ((xg[i] - xT[i]) - F[i]) / h;
        }
        __syncthreads();
        double J[n * n], dp[n];
#pragma unroll
        for (int j = 0; j < n; j++)
#pragma unroll
            for (int i = 0; i < n; i++) J[i * n + j] = ex[(g * G + 1 + j) * (2 * n + 1) + i];
        __syncthreads();
        bool stepok = newton_step<n>(J, F, dp);
        if (!done && !stepok) done = true;
        // line search: lane r of round q takes candidate c = q G + r of a = 1, 1/2, 1/4, ...; the first that decreases |F| wins
        bool dec = false;
        for (int q = 0; q * G < NC; q++) {
            const int c = q * G + r;
            double a = 1.0;
            for (int e = 0; e < c; e++) a *= 0.5;
            double pn[n], Fn[n], nn = 0.0;
#pragma unroll
            for (int i = 0; i < n; i++) pn[i] = pv[i] + a * dp[i];
            const bool mine = !done && !dec && c < NC;
            if (mine) integrate<MODEL>(S, x0, pn, tf, xT, nullptr, nullptr);
#pragma unroll
            for (int i = 0; i < n; i++) { Fn[i] = xg[i] - xT[i]; nn = (Fn[i] != Fn[i] || nn != nn) ? NAN : fmax(nn, fabs(Fn[i])); }
            const unsigned long long good = __ballot(mine && nn < nf);
            const unsigned grp = (unsigned)((good >> (g * G)) & ((1ull << G) - 1ull));
#pragma unroll
            for (int i = 0; i < n; i++) { ex[lane * (2 * n + 1) + i] = pn[i]; ex[lane * (2 * n + 1) + n + i] = Fn[i]; }
            ex[lane * (2 * n + 1) + 2 * n] = nn;
            __syncthreads();
            if (!done && !dec && grp != 0) {
                const int w = g * G + (__ffs(grp) - 1);
#pragma unroll
                for (int i = 0; i < n; i++) { pv[i] = ex[w * (2 * n + 1) + i]; F[i] = ex[w * (2 * n + 1) + n + i]; }
                nf = ex[w * (2 * n + 1) + 2 * n];
                dec = true;
            }
            __syncthreads();
        }
        if (!done && !dec) done = true;
        else if (!done) it++;
    }
    if (have && r == 0) {
        S.status[b] = ok; S.iters[b] = it; S.resid[b] = nf;
        for (int i = 0; i < n; i++) S.p_out[(size_t)b * n + i] = pv[i];
        if (ok) integrate<MODEL>(S, x0, pv, tf, xT, S.X, S.U, b);
    }
}

// out[b][r] = in[r][b] for the problems whose shooting converged (R = N n or N m rows, B columns): 32 x 32 tiles through
// LDS, reads and writes both coalesced; the rows of the other problems are left as they are
__global__ void __launch_bounds__(256) shoot_transpose_kernel(const double* in, double* out, const int* status, int R, int B) {
    __shared__ double tile[32][33];
    const int b0 = blockIdx.x * 32, r0 = blockIdx.y * 32, tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    for (int j = ty; j < 32; j += 8) {
        const int r = r0 + j, b = b0 + tx;
        tile[j][tx] = (r < R && b < B) ? in[(size_t)r * B + b] : 0.0;
    }
    __syncthreads();
    for (int j = ty; j < 32; j += 8) {
        const int b = b0 + j, r = r0 + tx;
        if (b < B && r < R && status[b]) out[(size_t)b * R + r] = tile[tx][j];
    }
}

}  // namespace

extern "C" {

int gusto_default_shoot_opts(gusto_shoot_opts* o) {
    if (!o) return GUSTO_ERR_ARG;
    o->substeps = 4; o->max_newton = 100; o->ftol = 1e-3;      // shooting.jl:14: iterations = 100, ftol = 1e-3
    o->no_group_pass = 0;
    return GUSTO_OK;
}

int gusto_shoot(gusto_handle h, const double* p0, const gusto_shoot_opts* opts) {
    if (!h) return GUSTO_ERR_ARG;
    if (h->model != GUSTO_DUBINS_CAR && h->model != GUSTO_ASTROBEE_SE3_MANIFOLD) {
        h->err = "gusto_shoot: only DubinsCar and AstrobeeSE3Manifold have a shooting ODE (as in the reference)";
        return GUSTO_ERR_ARG;
    }
    const size_t n = h->n, m = h->m;
    if (!h->have_problems) { h->err = "gusto_shoot: call gusto_set_problems first"; return GUSTO_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    { int rc = gusto_finish(h); if (rc) return rc; }
    gusto_shoot_opts o;
    gusto_default_shoot_opts(&o);
    if (opts) o = *opts;
    if (o.substeps < 1 || o.max_newton < 0 || !(o.ftol > 0) || (o.no_group_pass != 0 && o.no_group_pass != 1)) { h->err = "gusto_shoot: bad options"; return GUSTO_ERR_ARG; }
    const size_t B = h->batch_cap, N = h->N;
    if (!h->d_shX) {
        HIPCHK(h, dalloc(&h->d_shX, B * N * n)); HIPCHK(h, dalloc(&h->d_shU, B * N * m)); HIPCHK(h, dalloc(&h->d_shP, B * n));
        HIPCHK(h, dalloc(&h->d_shXt, B * N * n)); HIPCHK(h, dalloc(&h->d_shUt, B * N * m));
        HIPCHK(h, dalloc(&h->d_shP0, B * n)); HIPCHK(h, dalloc(&h->d_shRes, B)); HIPCHK(h, dalloc(&h->d_shSt, B)); HIPCHK(h, dalloc(&h->d_shIt, B));
        HIPCHK(h, dalloc(&h->d_shList, B + 1));   // [B] problems for the group pass, [1] their number
    }
    if (p0) {
        HIPCHK(h, hipMemcpyAsync(h->d_shP0, p0, sizeof(double) * h->B * n, hipMemcpyHostToDevice, h->stream));
    } else {   // SCPS.dual of every problem (st_d rows: [toggle, spare, dual[n]])
        HIPCHK(h, hipMemcpy2DAsync(h->d_shP0, sizeof(double) * n, h->d_std + SD_DUAL, sizeof(double) * SD_ND, sizeof(double) * n, h->B,
                                   hipMemcpyDeviceToDevice, h->stream));
    }
    ShootParams S{};
    S.B = h->B; S.N = h->N; S.substeps = o.substeps; S.max_newton = o.max_newton; S.ftol = o.ftol;
    S.v = h->mp.dubins_v; S.k = h->mp.dubins_k; S.mass = h->mp.mass;
    for (int i = 0; i < 3; i++) S.J[i] = h->mp.Jdiag[i];
    S.x_init = h->d_xinit; S.goal_lo = h->d_glo; S.goal_hi = h->d_ghi; S.tf = h->d_tf; S.p0 = h->d_shP0;
    S.X = h->d_shXt; S.U = h->d_shUt; S.p_out = h->d_shP; S.resid = h->d_shRes; S.status = h->d_shSt; S.iters = h->d_shIt;
    // two passes: a lane per problem for the first SHOOT_CAP Newton iterations (the problems that converge need 0-4), then
    // the stragglers with a group of lanes each (shoot_group_kernel)
    constexpr int SHOOT_CAP = 8;
    S.cap = (o.no_group_pass || o.max_newton <= SHOOT_CAP) ? o.max_newton : SHOOT_CAP;
    S.list = h->d_shList; S.count = h->d_shList + h->batch_cap;
    S.active = h->n_active >= 0 ? h->d_active : nullptr;
    HIPCHK(h, hipMemsetAsync(S.count, 0, sizeof(int), h->stream));
    if (h->model == GUSTO_DUBINS_CAR) hipLaunchKernelGGL(shoot_kernel<GUSTO_DUBINS_CAR>, dim3((h->B + 63) / 64), dim3(64), 0, h->stream, S);
    else hipLaunchKernelGGL(shoot_kernel<GUSTO_ASTROBEE_SE3_MANIFOLD>, dim3((h->B + 63) / 64), dim3(64), 0, h->stream, S);
    HIPCHK(h, hipGetLastError());
    int n_later = 0;
    HIPCHK(h, hipMemcpyAsync(&n_later, S.count, sizeof(int), hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    if (n_later > 0) {
        // (16 lanes per problem: the 14 candidates of a line search in one round; measured 83 ms against 104 with 8 lanes and
        // 112 with 4 for dubins_car B = 65 536, 368 ms with the lane-per-problem pass alone)
        if (h->model == GUSTO_DUBINS_CAR) hipLaunchKernelGGL((shoot_group_kernel<GUSTO_DUBINS_CAR, 16>), dim3((n_later + 3) / 4), dim3(64), 0, h->stream, S, n_later);
        else hipLaunchKernelGGL((shoot_group_kernel<GUSTO_ASTROBEE_SE3_MANIFOLD, 16>), dim3((n_later + 3) / 4), dim3(64), 0, h->stream, S, n_later);
        HIPCHK(h, hipGetLastError());
    }
    {   // knot-major staging -> X[b][k][i], U[b][k][i]
        const int RX = (int)(N * n), RU = (int)(N * m);
        hipLaunchKernelGGL(shoot_transpose_kernel, dim3((h->B + 31) / 32, (RX + 31) / 32), dim3(256), 0, h->stream, h->d_shXt, h->d_shX, h->d_shSt, RX, h->B);
        hipLaunchKernelGGL(shoot_transpose_kernel, dim3((h->B + 31) / 32, (RU + 31) / 32), dim3(256), 0, h->stream, h->d_shUt, h->d_shU, h->d_shSt, RU, h->B);
        HIPCHK(h, hipGetLastError());
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_shoot = true;
    return GUSTO_OK;
}

int gusto_get_shoot(gusto_handle h, int* status, int* newton_iters, double* resid, double* p0, double* X, double* U) {
    if (!h) return GUSTO_ERR_ARG;
    if (!h->have_shoot) { h->err = "gusto_get_shoot: call gusto_shoot first"; return GUSTO_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    const size_t B = h->B, N = h->N, n = h->n, m = h->m;
    if (status) HIPCHK(h, hipMemcpy(status, h->d_shSt, sizeof(int) * B, hipMemcpyDeviceToHost));
    if (newton_iters) HIPCHK(h, hipMemcpy(newton_iters, h->d_shIt, sizeof(int) * B, hipMemcpyDeviceToHost));
    if (resid) HIPCHK(h, hipMemcpy(resid, h->d_shRes, sizeof(double) * B, hipMemcpyDeviceToHost));
    if (p0) HIPCHK(h, hipMemcpy(p0, h->d_shP, sizeof(double) * B * n, hipMemcpyDeviceToHost));
    if (X) HIPCHK(h, hipMemcpy(X, h->d_shX, sizeof(double) * B * N * n, hipMemcpyDeviceToHost));
    if (U) HIPCHK(h, hipMemcpy(U, h->d_shU, sizeof(double) * B * N * m, hipMemcpyDeviceToHost));
    return GUSTO_OK;
}

}  // extern "C"
