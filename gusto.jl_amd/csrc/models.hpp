// models.hpp -- per-model dynamics f, A = df/dx, B = df/du and the analytic signed distances.
// The reference's model plug-in surface (multiple dispatch on the model type) becomes a compile-time
// trait: f_dyn/update_f!, A_dyn/update_A!, B_dyn of src/dynamics/<model>.jl.
#pragma once
#include "common.hpp"

namespace gusto {

GD void cross3(double* c, const double* a, const double* b) {
    c[0] = a[1] * b[2] - a[2] * b[1];
    c[1] = a[2] * b[0] - a[0] * b[2];
    c[2] = a[0] * b[1] - a[1] * b[0];
}

template <int MODEL> struct Dyn;

// freeflyer_se2.jl:182-206 : x = (r, th, v, w), u = (F, M); exactly linear
template <> struct Dyn<GUSTO_FREEFLYER_SE2> {
    static constexpr int n = 6, m = 3;
    GD static void f(const gusto_model_params& mp, const double* x, const double* u, double* f) {
        f[0] = x[3]; f[1] = x[4]; f[2] = x[5];
        f[3] = u[0] / mp.mass; f[4] = u[1] / mp.mass; f[5] = u[2] * (1.0 / mp.Jdiag[2]);
    }
    GD static void A(const gusto_model_params&, const double*, const double*, double* A) {
#pragma unroll
        for (int i = 0; i < n * n; i++) A[i] = 0;
        A[0 * n + 3] = 1; A[1 * n + 4] = 1; A[2 * n + 5] = 1;
    }
    GD static void B(const gusto_model_params& mp, double* B) {
#pragma unroll
        for (int i = 0; i < n * m; i++) B[i] = 0;
        B[3 * m + 0] = 1.0 / mp.mass; B[4 * m + 1] = 1.0 / mp.mass; B[5 * m + 2] = 1.0 / mp.Jdiag[2];
    }
};

// dubins_car.jl:161-181 : x = (x, y, th), u = turn rate
template <> struct Dyn<GUSTO_DUBINS_CAR> {
    static constexpr int n = 3, m = 1;
    GD static void f(const gusto_model_params& mp, const double* x, const double* u, double* f) {
        f[0] = mp.dubins_v * cos(x[2]); f[1] = mp.dubins_v * sin(x[2]); f[2] = mp.dubins_k * u[0];
    }
    GD static void A(const gusto_model_params& mp, const double* x, const double*, double* A) {
#pragma unroll
        for (int i = 0; i < n * n; i++) A[i] = 0;
        A[0 * n + 2] = -mp.dubins_v * sin(x[2]);
        A[1 * n + 2] = mp.dubins_v * cos(x[2]);
    }
    GD static void B(const gusto_model_params& mp, double* B) { B[0] = 0; B[1] = 0; B[2] = mp.dubins_k; }
    // f and A at a fixed point from two cached numbers (v cos th, v sin th): the same values as f() and A() -- a negation
    // is exact -- without their sin / cos (~140 instructions apiece in double precision)
    GD static void lin_cache(const gusto_model_params& mp, const double* x, double* c) {
        c[0] = mp.dubins_v * cos(x[2]); c[1] = mp.dubins_v * sin(x[2]);
    }
    GD static void f_cached(const gusto_model_params& mp, const double* c, const double* u, double* f) {
        f[0] = c[0]; f[1] = c[1]; f[2] = mp.dubins_k * u[0];
    }
    GD static void A_cached(const double* c, double* A) {
#pragma unroll
        for (int i = 0; i < n * n; i++) A[i] = 0;
        A[0 * n + 2] = -c[1];
        A[1 * n + 2] = c[0];
    }
};

// astrobee_se3.jl:180-241 + quat_functions.jl:253-257 : x = (r, v, p_MRP, w), u = (F, M)
template <> struct Dyn<GUSTO_ASTROBEE_SE3> {
    static constexpr int n = 12, m = 6;
    GD static void f(const gusto_model_params& mp, const double* x, const double* u, double* f) {
        const double* pp = x + 6; const double* w = x + 9;
        const double Jx = mp.Jdiag[0], Jy = mp.Jdiag[1], Jz = mp.Jdiag[2];
#pragma unroll
        for (int i = 0; i < 3; i++) { f[i] = x[3 + i]; f[3 + i] = u[i] / mp.mass; }
        const double p2 = pp[0] * pp[0] + pp[1] * pp[1] + pp[2] * pp[2];
        const double wp = w[0] * pp[0] + w[1] * pp[1] + w[2] * pp[2];
        double cr[3]; cross3(cr, w, pp);
#pragma unroll
        for (int i = 0; i < 3; i++) f[6 + i] = 0.25 * ((1 - p2) * w[i] - 2 * cr[i] + 2 * wp * pp[i]);
        double Jw[3] = {Jx * w[0], Jy * w[1], Jz * w[2]}, c2[3];
        cross3(c2, w, Jw);
        f[9] = (u[3] - c2[0]) / Jx; f[10] = (u[4] - c2[1]) / Jy; f[11] = (u[5] - c2[2]) / Jz;
    }
    GD static void A(const gusto_model_params& mp, const double* x, const double*, double* A) {
        const double Jx = mp.Jdiag[0], Jy = mp.Jdiag[1], Jz = mp.Jdiag[2];
        const double px = x[6], py = x[7], pz = x[8], wx = x[9], wy = x[10], wz = x[11];
#pragma unroll
        for (int i = 0; i < n * n; i++) A[i] = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) A[i * n + 3 + i] = 1.0;
#define AA(i, j) A[((i)-1) * n + ((j)-1)]
        AA(7, 7) = (px * wx) / 2 + (py * wy) / 2 + (pz * wz) / 2;
        AA(7, 8) = wz / 2 + (px * wy) / 2 - (py * wx) / 2;
        AA(7, 9) = (px * wz) / 2 - wy / 2 - (pz * wx) / 2;
        AA(7, 10) = px * px / 4 - py * py / 4 - pz * pz / 4 + 0.25;
        AA(7, 11) = (px * py) / 2 - pz / 2;
        AA(7, 12) = py / 2 + (px * pz) / 2;
        AA(8, 7) = (py * wx) / 2 - (px * wy) / 2 - wz / 2;
        AA(8, 8) = (px * wx) / 2 + (py * wy) / 2 + (pz * wz) / 2;
        AA(8, 9) = wx / 2 + (py * wz) / 2 - (pz * wy) / 2;
        AA(8, 10) = pz / 2 + (px * py) / 2;
        AA(8, 11) = -px * px / 4 + py * py / 4 - pz * pz / 4 + 0.25;
        AA(8, 12) = (py * pz) / 2 - px / 2;
        AA(9, 7) = wy / 2 - (px * wz) / 2 + (pz * wx) / 2;
        AA(9, 8) = (pz * wy) / 2 - (py * wz) / 2 - wx / 2;
        AA(9, 9) = (px * wx) / 2 + (py * wy) / 2 + (pz * wz) / 2;
        AA(9, 10) = (px * pz) / 2 - py / 2;
        AA(9, 11) = px / 2 + (py * pz) / 2;
        AA(9, 12) = -px * px / 4 - py * py / 4 + pz * pz / 4 + 0.25;
        AA(10, 11) = (Jy - Jz) * wz / Jx;
        AA(10, 12) = (Jy - Jz) * wy / Jx;
        AA(11, 10) = -(Jx - Jz) * wz / Jy;
        AA(11, 12) = -(Jx - Jz) * wx / Jy;
        AA(12, 10) = (Jx - Jy) * wy / Jz;
        AA(12, 11) = (Jx - Jy) * wx / Jz;
#undef AA
    }
    GD static void B(const gusto_model_params& mp, double* B) {
#pragma unroll
        for (int i = 0; i < n * m; i++) B[i] = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) B[(3 + i) * m + i] = 1.0 / mp.mass;
        B[9 * m + 3] = 1.0 / mp.Jdiag[0]; B[10 * m + 4] = 1.0 / mp.Jdiag[1]; B[11 * m + 5] = 1.0 / mp.Jdiag[2];
    }
};

// astrobee_se3_manifold.jl:231-304 : x = (r, v, q scalar-first, w)
template <> struct Dyn<GUSTO_ASTROBEE_SE3_MANIFOLD> {
    static constexpr int n = 13, m = 6;
    GD static void f(const gusto_model_params& mp, const double* x, const double* u, double* f) {
        const double qw = x[6], qx = x[7], qy = x[8], qz = x[9], wx = x[10], wy = x[11], wz = x[12];
        const double Jx = mp.Jdiag[0], Jy = mp.Jdiag[1], Jz = mp.Jdiag[2];
#pragma unroll
        for (int i = 0; i < 3; i++) { f[i] = x[3 + i]; f[3 + i] = u[i] / mp.mass; }
        f[6] = 0.5 * (-wx * qx - wy * qy - wz * qz);
        f[7] = 0.5 * (wx * qw - wz * qy + wy * qz);
        f[8] = 0.5 * (wy * qw + wz * qx - wx * qz);
        f[9] = 0.5 * (wz * qw - wy * qx + wx * qy);
        double w[3] = {wx, wy, wz}, Jw[3] = {Jx * wx, Jy * wy, Jz * wz}, c2[3];
        cross3(c2, w, Jw);
        f[10] = (u[3] - c2[0]) / Jx; f[11] = (u[4] - c2[1]) / Jy; f[12] = (u[5] - c2[2]) / Jz;
    }
    GD static void A(const gusto_model_params& mp, const double* x, const double*, double* A) {
        const double qw = x[6], qx = x[7], qy = x[8], qz = x[9], wx = x[10], wy = x[11], wz = x[12];
        const double Jx = mp.Jdiag[0], Jy = mp.Jdiag[1], Jz = mp.Jdiag[2];
#pragma unroll
        for (int i = 0; i < n * n; i++) A[i] = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) A[i * n + 3 + i] = 1.0;
#define AA(i, j) A[((i)-1) * n + ((j)-1)]
        AA(7, 8) = -wx / 2; AA(7, 9) = -wy / 2; AA(7, 10) = -wz / 2;
        AA(7, 11) = -qx / 2; AA(7, 12) = -qy / 2; AA(7, 13) = -qz / 2;
        AA(8, 7) = wx / 2; AA(8, 9) = -wz / 2; AA(8, 10) = wy / 2;
        AA(8, 11) = qw / 2; AA(8, 12) = qz / 2; AA(8, 13) = -qy / 2;
        AA(9, 7) = wy / 2; AA(9, 8) = wz / 2; AA(9, 10) = -wx / 2;
        AA(9, 11) = -qz / 2; AA(9, 12) = qw / 2; AA(9, 13) = qx / 2;
        AA(10, 7) = wz / 2; AA(10, 8) = -wy / 2; AA(10, 9) = wx / 2;
        AA(10, 11) = qy / 2; AA(10, 12) = -qx / 2; AA(10, 13) = qw / 2;
        AA(11, 12) = (Jy - Jz) * wz / Jx;
        AA(11, 13) = (Jy - Jz) * wy / Jx;
        AA(12, 11) = -(Jx - Jz) * wz / Jy;
        AA(12, 13) = -(Jx - Jz) * wx / Jy;
        AA(13, 11) = (Jx - Jy) * wy / Jz;
        AA(13, 12) = (Jx - Jy) * wx / Jz;
#undef AA
    }
    GD static void B(const gusto_model_params& mp, double* B) {
#pragma unroll
        for (int i = 0; i < n * m; i++) B[i] = 0;
#pragma unroll
        for (int i = 0; i < 3; i++) B[(3 + i) * m + i] = 1.0 / mp.mass;
        B[10 * m + 3] = 1.0 / mp.Jdiag[0]; B[11 * m + 4] = 1.0 / mp.Jdiag[1]; B[12 * m + 5] = 1.0 / mp.Jdiag[2];
    }
};

// ---- signed distance: replaces BulletCollision.distance(env, rb_idx, r, env_idx) -------------------
// (call sites freeflyer_se2.jl:257,275,413,419; astrobee_se3.jl:275,291,403,409; manifold.jl:352,490,629,635)
// robot body = disc (2-D models) / sphere (3-D) of radius mp.radius, obstacles = AABBs then spheres.
template <int WS> GD double sdf_box(const double* q, const double* lo, const double* hi, double* nh) {
    double v[WS], s2 = 0;
    bool outside = false;
#pragma unroll
    for (int i = 0; i < WS; i++) {
        double e = 0;
        if (q[i] < lo[i]) e = q[i] - lo[i];
        else if (q[i] > hi[i]) e = q[i] - hi[i];
        v[i] = e;
        outside = outside || (e != 0);
        s2 += e * e;
    }
    if (outside) {
        const double dist = sqrt(s2);
#pragma unroll
        for (int i = 0; i < WS; i++) nh[i] = v[i] / dist;
        return dist;
    }
    double best = q[0] - lo[0];
    int bi = 0;
    double bs = -1.0;
#pragma unroll
    for (int i = 0; i < WS; i++) {
        const double a = q[i] - lo[i], b = hi[i] - q[i];
        if (a < best) { best = a; bi = i; bs = -1.0; }
        if (b < best) { best = b; bi = i; bs = 1.0; }
    }
#pragma unroll
    for (int i = 0; i < WS; i++) nh[i] = (i == bi) ? bs : 0.0;
    return -best;
}

// The keep-out set is read-only for the whole launch and indexed by a wave-uniform obstacle number: reading it through
// the CONSTANT address space makes these scalar loads (s_load_dwordx4 into SGPRs, served by the scalar cache).  As plain
// global pointers hipcc cannot prove that none of the kernel's stores aliases them and emits one global_load +
// s_waitcnt vmcnt(0) per obstacle for all 64 lanes -- 14 (freeflyer) to 32 (ISS corner) serial memory round trips per
// distance loop, and rho alone walks that loop four times per trip.
// TrajOpt variants: the dynamics of the base model; B carries zero columns for the defect controls (they do not enter
// xdot: a defect belongs to ONE trapezoid row, ipm.hpp)
template <int MODEL, int BASE> struct DynTO {
    using D0 = Dyn<BASE>;
    static constexpr int n = D0::n, m0 = D0::m, m = m0 + n;
    GD static void f(const gusto_model_params& mp, const double* x, const double* u, double* f) { D0::f(mp, x, u, f); }
    GD static void A(const gusto_model_params& mp, const double* x, const double* u, double* A) { D0::A(mp, x, u, A); }
    GD static void B(const gusto_model_params& mp, double* B) {
        double B0[n * m0];
        D0::B(mp, B0);
#pragma unroll
        for (int i = 0; i < n; i++)
#pragma unroll
            for (int j = 0; j < m; j++) B[i * m + j] = j < m0 ? B0[i * m0 + j] : 0.0;
    }
};
template <> struct Dyn<GUSTO_TO_FREEFLYER_SE2> : DynTO<GUSTO_TO_FREEFLYER_SE2, GUSTO_FREEFLYER_SE2> {};
template <> struct Dyn<GUSTO_TO_ASTROBEE_SE3> : DynTO<GUSTO_TO_ASTROBEE_SE3, GUSTO_ASTROBEE_SE3> {};
template <> struct Dyn<GUSTO_TO_ASTROBEE_SE3_MANIFOLD> : DynTO<GUSTO_TO_ASTROBEE_SE3_MANIFOLD, GUSTO_ASTROBEE_SE3_MANIFOLD> {};

typedef const __attribute__((address_space(4))) double cdouble;
GD const cdouble* as_constant(const double* p) { return (const cdouble*)(uintptr_t)p; }

// The keep-out set of ONE problem (Workspace(robot, env), types.jl:12-24; every ProblemDefinition owns its env, :32-39):
// n_box AABBs, then n_obs - n_box spheres.  With gusto_set_env the whole batch shares one table (the kernel arguments);
// with gusto_set_env_batch problem b has its own slice of the concatenated tables, found through KParams::env -- an
// (offset, count) record per problem that is read ONCE per problem through the constant address space (b is wave-uniform),
// so the table base stays a scalar and the obstacle loop keeps its s_load form.
struct Env {
    const double* box;
    const double* sph;
    int n_box, n_obs;
};
GD Env problem_env(const KParams& P, int b) {
    Env e;
    if (P.env) {
        typedef const __attribute__((address_space(4))) int cint;
        const cint* q = (const cint*)(uintptr_t)(P.env + 4 * (size_t)b);
        const int bo = q[0], nb = q[1], so = q[2], ns = q[3];
        e.box = P.box + 6 * (size_t)bo; e.sph = P.sph + 4 * (size_t)so; e.n_box = nb; e.n_obs = nb + ns;
    } else {
        e.box = P.box; e.sph = P.sph; e.n_box = P.n_box; e.n_obs = P.n_obs;
    }
    return e;
}

template <int WS>
GD double signed_distance(const KParams& P, const Env& E, int comp, const double* r, int i, double* nh) {
    double q[WS];
#pragma unroll
    for (int j = 0; j < WS; j++) q[j] = r[j] + P.mp.comp_off[comp][j];
    if (i < E.n_box) {
        const cdouble* bx = as_constant(E.box) + 6 * i;
        double lo[WS], hi[WS];
#pragma unroll
        for (int j = 0; j < WS; j++) { lo[j] = bx[j]; hi[j] = bx[3 + j]; }
        return sdf_box<WS>(q, lo, hi, nh) - P.mp.radius;
    }
    const cdouble* sp = as_constant(E.sph) + 4 * (i - E.n_box);
    double v[WS], s2 = 0;
#pragma unroll
    for (int j = 0; j < WS; j++) { v[j] = q[j] - sp[j]; s2 += v[j] * v[j]; }
    const double nrm = sqrt(s2);
#pragma unroll
    for (int j = 0; j < WS; j++) nh[j] = v[j] / nrm;
    return nrm - sp[3] - P.mp.radius;
}

}  // namespace gusto
