// shoot.hip -- batched indirect shooting seeded by the SCP dual (src/shooting.jl:4-66, src/traj_opt.jl:4-45).
// One THREAD per problem: the state + costate ODE of a problem is 2n-dimensional (6 for DubinsCar) and its Newton system
// n x n, so a problem is register work for one lane and a batch of thousands fills the GPU with independent lanes.
//   DubinsCar: shooting_ode! / get_control   src/dynamics/dubins_car.jl:259-280
// The reference integrates with DifferentialEquations' default adaptive method and solves F(p0) = x_goal - x(tf; p0) = 0
// with NLsolve (trust region, finite-difference Jacobian, ftol = 1e-3, <= 100 iterations).  Neither exists here; the
// scheme is stated instead (the test suite's CPU restatement follows the same one): classical RK4 with `substeps`
// steps per knot interval, Newton with a forward-difference Jacobian (h_j = 1e-6 max(1, |p_j|)), halving line search
// on |F|_inf.  (AstrobeeSE3Manifold's 26-dimensional shooting ODE, astrobee_se3_manifold.jl:831-1006, is not built.)
#include <hip/hip_runtime.h>

#include "handle.hpp"

using namespace gusto;

namespace {

struct ShootParams {
    int B, N, substeps, max_newton;
    double ftol, v, k;
    const double *x_init, *goal_lo, *goal_hi, *tf, *p0;   // p0 [B][3]
    double *X, *U, *p_out, *resid;                        // X [B][N][3], U [B][N]
    int *status, *iters;
};

__device__ __forceinline__ void rhs(const ShootParams& S, const double* z, double* dz) {
    const double u = 0.5 * S.k * z[5];
    double sn, cs;
    sincos(z[2], &sn, &cs);
    dz[0] = S.v * cs; dz[1] = S.v * sn; dz[2] = S.k * u;
    dz[3] = 0.0; dz[4] = 0.0; dz[5] = z[3] * S.v * sn - z[4] * S.v * cs;
}

// integrates from (x_init, p0); writes the knots when X != nullptr; returns x(tf) in xT
__device__ void integrate(const ShootParams& S, const double* x0, const double* p0, double tf, double* xT, double* X, double* U) {
    double z[6], k1[6], k2[6], k3[6], k4[6], w[6];
    const double h = tf / ((S.N - 1) * (double)S.substeps);
#pragma unroll
    for (int i = 0; i < 3; i++) { z[i] = x0[i]; z[3 + i] = p0[i]; }
    for (int k = 0; k < S.N; k++) {
        if (X) {
#pragma unroll
            for (int i = 0; i < 3; i++) X[k * 3 + i] = z[i];
            U[k] = 0.5 * S.k * z[5];                       // get_control: U = k/2 p_theta
        }
        if (k == S.N - 1) break;
        for (int s = 0; s < S.substeps; s++) {
            rhs(S, z, k1);
#pragma unroll
            for (int i = 0; i < 6; i++) w[i] = z[i] + 0.5 * h * k1[i];
            rhs(S, w, k2);
#pragma unroll
            for (int i = 0; i < 6; i++) w[i] = z[i] + 0.5 * h * k2[i];
            rhs(S, w, k3);
#pragma unroll
            for (int i = 0; i < 6; i++) w[i] = z[i] + h * k3[i];
            rhs(S, w, k4);
#pragma unroll
            for (int i = 0; i < 6; i++) z[i] += h / 6.0 * (k1[i] + 2 * k2[i] + 2 * k3[i] + k4[i]);
        }
    }
#pragma unroll
    for (int i = 0; i < 3; i++) xT[i] = z[i];
}

__global__ void __launch_bounds__(64) shoot_dubins_kernel(const ShootParams S) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= S.B) return;
    double x0[3], pv[3], xg[3], F[3], xT[3], nf = 0.0;
    const double tf = S.tf[b];
#pragma unroll
    for (int i = 0; i < 3; i++) {
        x0[i] = S.x_init[(size_t)b * 3 + i];
        pv[i] = S.p0[(size_t)b * 3 + i];
        const double lo = S.goal_lo[(size_t)b * 3 + i], hi = S.goal_hi[(size_t)b * 3 + i];
        xg[i] = (isfinite(lo) && isfinite(hi)) ? 0.5 * (lo + hi) : 0.0;     // ShootingProblem ctor, types.jl:219-226
    }
    integrate(S, x0, pv, tf, xT, nullptr, nullptr);
#pragma unroll
    for (int i = 0; i < 3; i++) { F[i] = xg[i] - xT[i]; nf = fmax(nf, fabs(F[i])); }
    int it = 0, ok = 0;
    for (;; it++) {
        if (!(nf == nf) || !isfinite(nf)) break;
        if (nf <= S.ftol) { ok = 1; break; }
        if (it >= S.max_newton) break;
        double J[9], pj[3];
        for (int j = 0; j < 3; j++) {
            const double h = 1e-6 * fmax(1.0, fabs(pv[j]));
#pragma unroll
            for (int i = 0; i < 3; i++) pj[i] = (i == j) ? pv[i] + h : pv[i];
            integrate(S, x0, pj, tf, xT, nullptr, nullptr);
#pragma unroll
            for (int i = 0; i < 3; i++) J[i * 3 + j] = ((xg[i] - xT[i]) - F[i]) / h;
        }
        const double det = J[0] * (J[4] * J[8] - J[5] * J[7]) - J[1] * (J[3] * J[8] - J[5] * J[6]) + J[2] * (J[3] * J[7] - J[4] * J[6]);
        if (!(fabs(det) > 1e-300) || !isfinite(det)) break;
        double dp[3];
        dp[0] = -(F[0] * (J[4] * J[8] - J[5] * J[7]) - J[1] * (F[1] * J[8] - J[5] * F[2]) + J[2] * (F[1] * J[7] - J[4] * F[2])) / det;
        dp[1] = -(J[0] * (F[1] * J[8] - J[5] * F[2]) - F[0] * (J[3] * J[8] - J[5] * J[6]) + J[2] * (J[3] * F[2] - F[1] * J[6])) / det;
        dp[2] = -(J[0] * (J[4] * F[2] - F[1] * J[7]) - J[1] * (J[3] * F[2] - F[1] * J[6]) + F[0] * (J[3] * J[7] - J[4] * J[6])) / det;
        double a = 1.0, nn = 0.0, Fn[3], pn[3];
        bool dec = false;
        while (a > 1e-4) {
#pragma unroll
            for (int i = 0; i < 3; i++) pn[i] = pv[i] + a * dp[i];
            integrate(S, x0, pn, tf, xT, nullptr, nullptr);
            nn = 0.0;
#pragma unroll
            for (int i = 0; i < 3; i++) { Fn[i] = xg[i] - xT[i]; nn = fmax(nn, fabs(Fn[i])); }
            if (nn < nf) { dec = true; break; }
            a *= 0.5;
        }
        if (!dec) break;
#pragma unroll
        for (int i = 0; i < 3; i++) { pv[i] = pn[i]; F[i] = Fn[i]; }
        nf = nn;
    }
    S.status[b] = ok; S.iters[b] = it; S.resid[b] = nf;
#pragma unroll
    for (int i = 0; i < 3; i++) S.p_out[(size_t)b * 3 + i] = pv[i];
    if (ok) integrate(S, x0, pv, tf, xT, S.X + (size_t)b * S.N * 3, S.U + (size_t)b * S.N);
}

}  // namespace

extern "C" {

int gusto_default_shoot_opts(gusto_shoot_opts* o) {
    if (!o) return GUSTO_ERR_ARG;
    o->substeps = 4; o->max_newton = 100; o->ftol = 1e-3;      // shooting.jl:14: iterations = 100, ftol = 1e-3
    return GUSTO_OK;
}

int gusto_shoot(gusto_handle h, const double* p0, const gusto_shoot_opts* opts) {
    if (!h) return GUSTO_ERR_ARG;
    if (h->model != GUSTO_DUBINS_CAR) { h->err = "gusto_shoot: only DubinsCar has a shooting ODE in this library"; return GUSTO_ERR_ARG; }
    if (!h->have_problems) { h->err = "gusto_shoot: call gusto_set_problems first"; return GUSTO_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    { int rc = gusto_finish(h); if (rc) return rc; }
    gusto_shoot_opts o;
    gusto_default_shoot_opts(&o);
    if (opts) o = *opts;
    if (o.substeps < 1 || o.max_newton < 0 || !(o.ftol > 0)) { h->err = "gusto_shoot: bad options"; return GUSTO_ERR_ARG; }
    const size_t B = h->batch_cap, N = h->N;
    if (!h->d_shX) {
        HIPCHK(h, dalloc(&h->d_shX, B * N * 3)); HIPCHK(h, dalloc(&h->d_shU, B * N)); HIPCHK(h, dalloc(&h->d_shP, B * 3));
        HIPCHK(h, dalloc(&h->d_shP0, B * 3)); HIPCHK(h, dalloc(&h->d_shRes, B)); HIPCHK(h, dalloc(&h->d_shSt, B)); HIPCHK(h, dalloc(&h->d_shIt, B));
    }
    if (p0) {
        HIPCHK(h, hipMemcpyAsync(h->d_shP0, p0, sizeof(double) * h->B * 3, hipMemcpyHostToDevice, h->stream));
    } else {   // SCPS.dual of every problem (st_d rows: [toggle, spare, dual[n]])
        HIPCHK(h, hipMemcpy2DAsync(h->d_shP0, sizeof(double) * 3, h->d_std + SD_DUAL, sizeof(double) * SD_ND, sizeof(double) * 3, h->B,
                                   hipMemcpyDeviceToDevice, h->stream));
    }
    ShootParams S{};
    S.B = h->B; S.N = h->N; S.substeps = o.substeps; S.max_newton = o.max_newton; S.ftol = o.ftol;
    S.v = h->mp.dubins_v; S.k = h->mp.dubins_k;
    S.x_init = h->d_xinit; S.goal_lo = h->d_glo; S.goal_hi = h->d_ghi; S.tf = h->d_tf; S.p0 = h->d_shP0;
    S.X = h->d_shX; S.U = h->d_shU; S.p_out = h->d_shP; S.resid = h->d_shRes; S.status = h->d_shSt; S.iters = h->d_shIt;
    hipLaunchKernelGGL(shoot_dubins_kernel, dim3((h->B + 63) / 64), dim3(64), 0, h->stream, S);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->have_shoot = true;
    return GUSTO_OK;
}

int gusto_get_shoot(gusto_handle h, int* status, int* newton_iters, double* resid, double* p0, double* X, double* U) {
    if (!h) return GUSTO_ERR_ARG;
    if (!h->have_shoot) { h->err = "gusto_get_shoot: call gusto_shoot first"; return GUSTO_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    const size_t B = h->B, N = h->N;
    if (status) HIPCHK(h, hipMemcpy(status, h->d_shSt, sizeof(int) * B, hipMemcpyDeviceToHost));
    if (newton_iters) HIPCHK(h, hipMemcpy(newton_iters, h->d_shIt, sizeof(int) * B, hipMemcpyDeviceToHost));
    if (resid) HIPCHK(h, hipMemcpy(resid, h->d_shRes, sizeof(double) * B, hipMemcpyDeviceToHost));
    if (p0) HIPCHK(h, hipMemcpy(p0, h->d_shP, sizeof(double) * B * 3, hipMemcpyDeviceToHost));
    if (X) HIPCHK(h, hipMemcpy(X, h->d_shX, sizeof(double) * B * N * 3, hipMemcpyDeviceToHost));
    if (U) HIPCHK(h, hipMemcpy(U, h->d_shU, sizeof(double) * B * N, hipMemcpyDeviceToHost));
    return GUSTO_OK;
}

}  // extern "C"
