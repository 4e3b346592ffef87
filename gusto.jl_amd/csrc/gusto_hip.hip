// gusto_hip.hip -- host side of libgusto_hip.so: the C ABI of include/gusto_hip.h over the HIP kernels.
// There is no CPU fallback: every entry point needs a gfx950 device and fails loudly without one.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "handle.hpp"

using namespace gusto;

thread_local std::string g_err;

extern "C" {

int gusto_model_dims(int model, int* n, int* m) {
    switch (model) {
    case GUSTO_FREEFLYER_SE2: *n = 6; *m = 3; return GUSTO_OK;
    case GUSTO_DUBINS_CAR: *n = 3; *m = 1; return GUSTO_OK;
    case GUSTO_ASTROBEE_SE3: *n = 12; *m = 6; return GUSTO_OK;
    case GUSTO_ASTROBEE_SE3_MANIFOLD: *n = 13; *m = 6; return GUSTO_OK;
    }
    return GUSTO_ERR_ARG;
}

int gusto_default_params(int model, gusto_scp_params* sp, gusto_model_params* mp) {
    int n, m;
    if (gusto_model_dims(model, &n, &m) || !sp || !mp) return GUSTO_ERR_ARG;
    memset(sp, 0, sizeof(*sp));
    memset(mp, 0, sizeof(*mp));
    sp->omega_max = 1.0e10; sp->beta_succ = 2.0; sp->beta_fail = 0.5; sp->omega0 = 1.0;
    mp->n_robot_comp = 1;
    const double pi = 3.14159265358979323846;
    switch (model) {
    case GUSTO_FREEFLYER_SE2:  // freeflyer_se2.jl:15-39, robot/freeflyer.jl:28-62
        sp->Delta0 = 3.0; sp->eps = 1.0e-2; sp->rho0 = 0.1; sp->rho1 = 0.3; sp->gamma_fail = 10.0;
        sp->convergence_threshold = 1.0e-2;
        mp->mass = 0.5 * (15.36 + 18.08);
        mp->Jdiag[0] = mp->Jdiag[1] = mp->Jdiag[2] = 0.184;
        mp->radius = 0.157; mp->clearance = 0.05;
        mp->hard_limit_vel = 0.2;
        mp->hard_limit_accel = 2 * 0.185 / mp->mass;
        mp->hard_limit_omega = 20 * pi / 180;
        mp->hard_limit_alpha = (1.0 / (0.184 / 6.43)) * 0.593;
        mp->n_robot_comp = 2;  // body + arm cylinder (freeflyer.jl:53-57)
        mp->comp_off[1][1] = 0.15;
        break;
    case GUSTO_DUBINS_CAR:  // dubins_car.jl:22-52
        sp->Delta0 = 10000.0; sp->eps = 1.0e-6; sp->rho0 = 0.4; sp->rho1 = 1.5; sp->gamma_fail = 5.0;
        sp->convergence_threshold = 1e-4;
        mp->dubins_v = 2.0; mp->dubins_k = 1.0;
        mp->x_max[0] = 100.0; mp->x_max[1] = 100.0; mp->x_max[2] = 2 * pi;
        for (int i = 0; i < 3; i++) mp->x_min[i] = -mp->x_max[i];
        mp->u_max = 10.0; mp->u_min = -10.0;
        mp->clearance = 0.01;
        break;
    case GUSTO_ASTROBEE_SE3:           // astrobee_se3.jl:16-40, robot/astrobee3D.jl:15-33
    case GUSTO_ASTROBEE_SE3_MANIFOLD:  // astrobee_se3_manifold.jl:18-46
        if (model == GUSTO_ASTROBEE_SE3) {
            sp->Delta0 = 10.0; sp->eps = 1.0e-6; sp->rho0 = 0.01; sp->rho1 = 0.05; sp->gamma_fail = 5.0;
            sp->convergence_threshold = 1e-2;
        } else {
            sp->Delta0 = 1000.0; sp->eps = 1.0e-1; sp->rho0 = 0.01; sp->rho1 = 100.0; sp->gamma_fail = 5.0;
            sp->convergence_threshold = 1e-4;
        }
        mp->mass = 7.0;
        mp->Jdiag[0] = mp->Jdiag[1] = mp->Jdiag[2] = 0.1083;
        mp->radius = sqrt(3.0) * 0.5 * 0.305;
        mp->clearance = 0.03;
        mp->hard_limit_vel = 0.5; mp->hard_limit_accel = 0.1;
        mp->hard_limit_omega = 45 * pi / 180; mp->hard_limit_alpha = 50 * pi / 180;
        break;
    }
    return GUSTO_OK;
}

int gusto_default_ipm_opts(gusto_ipm_opts* o) {
    if (!o) return GUSTO_ERR_ARG;
    o->tol = 1e-8; o->tol_acc = 1e-5; o->mu_floor = -1.0 /* the model's, common.hpp: warm_defaults */; o->tr_tol = 1e-6; o->max_iter = 60; o->acc_iter = 0;
    // negative = the model's warm-start triple and the algorithm's centring bound (0.1 for GuSTO, none for TrajOpt), common.hpp: warm_defaults
    o->mu_warm = -1.0; o->mu_warm_gain = -1.0; o->mu_warm_max = -1.0; o->sigma_max = -1.0;
    return GUSTO_OK;
}

const char* gusto_last_error(gusto_handle h) { return h ? h->err.c_str() : g_err.c_str(); }

static int create_impl(gusto_handle* out, int model, int N, int batch_cap, int hist_cap, int device, bool trajopt) {
    int n, m;
    if (!out || gusto_model_dims(model, &n, &m) || N < 3 || N > 256 || batch_cap < 1 || hist_cap < 4) {
        g_err = "gusto_create: bad argument (need 3 <= N <= 256, batch_cap >= 1, hist_cap >= 4)";
        return GUSTO_ERR_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || device < 0 || device >= ndev) {
        g_err = "gusto_create: no usable HIP device (libgusto_hip has no CPU fallback)";
        return GUSTO_ERR_NO_DEVICE;
    }
    if (trajopt && model != GUSTO_FREEFLYER_SE2 && model != GUSTO_ASTROBEE_SE3 && model != GUSTO_ASTROBEE_SE3_MANIFOLD) {
        g_err = "gusto_create_trajopt: FreeflyerSE2, AstrobeeSE3 and AstrobeeSE3Manifold have a TrajOpt variant";
        return GUSTO_ERR_ARG;
    }
    gusto_handle h = new gusto_handle_s();
    h->model = model; h->n = n; h->m = m; h->N = N; h->batch_cap = batch_cap; h->hist_cap = hist_cap; h->device = device;
    h->model_pub = model; h->m_pub = m; h->trajopt = trajopt;
    gusto_default_params(model, &h->sp, &h->mp);
    if (trajopt) {   // internal variant: controls (u, d), d = the n defect variables of a knot
        h->model = model == GUSTO_FREEFLYER_SE2 ? gusto::GUSTO_TO_FREEFLYER_SE2
                 : (model == GUSTO_ASTROBEE_SE3 ? gusto::GUSTO_TO_ASTROBEE_SE3 : gusto::GUSTO_TO_ASTROBEE_SE3_MANIFOLD);
        h->m = m = m + n;
        gusto_default_trajopt_params(model, &h->tp);
    }
    gusto_default_ipm_opts(&h->io);
    *out = h;
    HIPCHK(h, hipSetDevice(device));
    HIPCHK(h, hipStreamCreate(&h->stream));
    h->own_stream = true;
    HIPCHK(h, hipEventCreate(&h->ev0));
    HIPCHK(h, hipEventCreate(&h->ev1));
    const size_t B = batch_cap, H = hist_cap;
    HIPCHK(h, dalloc(&h->d_X, B * N * n)); HIPCHK(h, dalloc(&h->d_U, B * N * m));
    HIPCHK(h, dalloc(&h->d_xinit, B * n)); HIPCHK(h, dalloc(&h->d_glo, B * n)); HIPCHK(h, dalloc(&h->d_ghi, B * n));
    HIPCHK(h, dalloc(&h->d_tf, B));
    HIPCHK(h, dalloc(&h->d_sti, B * ST_NI)); HIPCHK(h, dalloc(&h->d_std, B * SD_ND));
    HIPCHK(h, dalloc(&h->d_Jt, B * H)); HIPCHK(h, dalloc(&h->d_Jf, B * H)); HIPCHK(h, dalloc(&h->d_conv, B * H));
    HIPCHK(h, dalloc(&h->d_Delta, B * H)); HIPCHK(h, dalloc(&h->d_omega, B * H)); HIPCHK(h, dalloc(&h->d_rho, B * H));
    HIPCHK(h, dalloc(&h->d_acc, B * H)); HIPCHK(h, dalloc(&h->d_scp, B * H)); HIPCHK(h, dalloc(&h->d_sol, B * H));
    HIPCHK(h, dalloc(&h->d_tr, B * H)); HIPCHK(h, dalloc(&h->d_cvx, B * H)); HIPCHK(h, dalloc(&h->d_ipm, B * H));
    HIPCHK(h, dalloc(&h->d_subD, B)); HIPCHK(h, dalloc(&h->d_subW, B)); HIPCHK(h, dalloc(&h->d_subT, B));
    HIPCHK(h, dalloc(&h->d_subX, B * N * n)); HIPCHK(h, dalloc(&h->d_subU, B * N * m)); HIPCHK(h, dalloc(&h->d_subObj, B));
    HIPCHK(h, dalloc(&h->d_subSt, B)); HIPCHK(h, dalloc(&h->d_subIt, B));
    HIPCHK(h, dalloc(&h->d_box, 1)); HIPCHK(h, dalloc(&h->d_sph, 1));
    if (trajopt) {
        HIPCHK(h, dalloc(&h->d_to_mu, B * H)); HIPCHK(h, dalloc(&h->d_to_xtol, B * H));
        HIPCHK(h, dalloc(&h->d_to_ftol, B * H)); HIPCHK(h, dalloc(&h->d_to_ctol, B * H));
    }
    return GUSTO_OK;
}
int gusto_create(gusto_handle* out, int model, int N, int batch_cap, int hist_cap, int device) {
    return create_impl(out, model, N, batch_cap, hist_cap, device, false);
}
int gusto_create_trajopt(gusto_handle* out, int model, int N, int batch_cap, int hist_cap, int device) {
    return create_impl(out, model, N, batch_cap, hist_cap, device, true);
}
int gusto_default_trajopt_params(int model, gusto_trajopt_params* tp) {   // freeflyer_se2.jl:49-64, astrobee_se3.jl:50-65, astrobee_se3_manifold.jl:56-70
    if (!tp || (model != GUSTO_FREEFLYER_SE2 && model != GUSTO_ASTROBEE_SE3 && model != GUSTO_ASTROBEE_SE3_MANIFOLD)) return GUSTO_ERR_ARG;
    memset(tp, 0, sizeof(*tp));
    tp->mu0 = 1.0; tp->c = 10.0; tp->tau_plus = 2.0; tp->tau_minus = 0.5; tp->k = 5.0; tp->ftol = 0.01; tp->ctol = 0.01;
    tp->max_penalty_iteration = 5; tp->max_convex_iteration = 5; tp->max_trust_iteration = 5;
    if (model == GUSTO_FREEFLYER_SE2) { tp->s0 = 1.0; tp->xtol = 0.1; } else { tp->s0 = 10.0; tp->xtol = 0.01; }
    return GUSTO_OK;
}

int gusto_destroy(gusto_handle h) {
    if (!h) return GUSTO_ERR_ARG;
    hipSetDevice(h->device);
    void* ptrs[] = {h->d_X, h->d_U, h->d_xinit, h->d_glo, h->d_ghi, h->d_tf, h->d_sti, h->d_std, h->d_Jt, h->d_Jf, h->d_conv,
                    h->d_Delta, h->d_omega, h->d_rho, h->d_acc, h->d_scp, h->d_sol, h->d_tr, h->d_cvx, h->d_ipm, h->d_ws,
                    h->d_prof, h->d_subD, h->d_subW, h->d_subT, h->d_subX, h->d_subU, h->d_subObj, h->d_subSt, h->d_subIt, h->d_box, h->d_sph,
                    h->d_to_mu, h->d_to_xtol, h->d_to_ftol, h->d_to_ctol, h->d_Upub, h->d_env, h->d_gX, h->d_gU, h->d_active};
    for (void* p : ptrs) if (p) hipFree(p);
    for (void* p : {(void*)h->d_shX, (void*)h->d_shU, (void*)h->d_shP, (void*)h->d_shP0, (void*)h->d_shRes, (void*)h->d_shSt, (void*)h->d_shIt, (void*)h->d_shList, (void*)h->d_shXt, (void*)h->d_shUt}) if (p) hipFree(p);
    if (h->d_order) hipFree(h->d_order);
    if (h->d_queue) hipFree(h->d_queue);
    if (h->d_sched_ord) hipFree(h->d_sched_ord);
    if (h->h_sched_err) hipHostFree(h->h_sched_err);
    if (h->ev_gather) hipEventDestroy(h->ev_gather);
    if (h->ev0) hipEventDestroy(h->ev0);
    if (h->ev1) hipEventDestroy(h->ev1);
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    delete h;
    return GUSTO_OK;
}

// every setter first completes an enqueued solve (gusto_solve_async) on the handle's own device; a latched scheduler error
// is not the setter's business (handle.hpp: it is surfaced by the getters and the solves until gusto_set_problems)
static int setter_enter(gusto_handle h) {
    HIPCHK(h, hipSetDevice(h->device));
    return gusto_complete(h);
}

int gusto_set_params(gusto_handle h, const gusto_scp_params* sp, const gusto_model_params* mp) {
    if (!h) return GUSTO_ERR_ARG;
    { int rc = setter_enter(h); if (rc) return rc; }
    if (sp) h->sp = *sp;
    if (mp) h->mp = *mp;
    return GUSTO_OK;
}
int gusto_set_ipm_opts(gusto_handle h, const gusto_ipm_opts* o) {
    if (!h || !o) return GUSTO_ERR_ARG;
    { int rc = setter_enter(h); if (rc) return rc; }
    h->io = *o;
    return GUSTO_OK;
}
int gusto_set_schedule(gusto_handle h, int probe_iters, int min_batch) {
    if (!h || probe_iters < 0 || min_batch < 1) return GUSTO_ERR_ARG;
    { int rc = setter_enter(h); if (rc) return rc; }
    h->probe_iters = probe_iters; h->probe_min_batch = min_batch; h->sched_forced = true;
    return GUSTO_OK;
}

int gusto_set_decomposition(gusto_handle h, int decomposition) {
    if (!h || decomposition < GUSTO_DECOMP_AUTO || decomposition > GUSTO_DECOMP_WAVE4) return GUSTO_ERR_ARG;
    { int rc = setter_enter(h); if (rc) return rc; }
    // (a lane per problem exists for the models without obstacle rows -- dubins_car; the others keep their wave per problem)
#ifndef GUSTO_WITH_LANE
    if (decomposition == GUSTO_DECOMP_LANE) { h->err = "the lane-per-problem kernel is not in this build (-DGUSTO_WITH_LANE)"; return GUSTO_ERR_ARG; }
#endif
    h->decomposition = decomposition;
    return GUSTO_OK;
}

int gusto_set_stream(gusto_handle h, void* s) {
    if (!h) return GUSTO_ERR_ARG;
    { int rc = setter_enter(h); if (rc) return rc; }   // a pending solve is synchronised on the stream it runs on
    if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
    h->stream = nullptr; h->own_stream = false;
    if (s) { h->stream = (hipStream_t)s; return GUSTO_OK; }
    hipStream_t ns = nullptr;
    HIPCHK(h, hipStreamCreate(&ns));
    h->stream = ns; h->own_stream = true;
    return GUSTO_OK;
}

int gusto_set_env(gusto_handle h, int n_box, const double* box, int n_sph, const double* sph) {
    if (!h || n_box < 0 || n_sph < 0 || (n_box && !box) || (n_sph && !sph)) return GUSTO_ERR_ARG;
    if (n_box + n_sph > 64) { h->err = "gusto_set_env: at most 64 keep-out components"; return GUSTO_ERR_ARG; }
    { int rc = setter_enter(h); if (rc) return rc; }
    hipFree(h->d_box); hipFree(h->d_sph);
    h->d_box = h->d_sph = nullptr;
    if (h->d_env) { hipFree(h->d_env); h->d_env = nullptr; }   // back to one keep-out set for the whole batch
    h->env_B = 0; h->n_obs_max = 0;
    HIPCHK(h, dalloc(&h->d_box, (size_t)6 * n_box)); HIPCHK(h, dalloc(&h->d_sph, (size_t)4 * n_sph));
    if (n_box) HIPCHK(h, hipMemcpy(h->d_box, box, sizeof(double) * 6 * n_box, hipMemcpyHostToDevice));
    if (n_sph) HIPCHK(h, hipMemcpy(h->d_sph, sph, sizeof(double) * 4 * n_sph, hipMemcpyHostToDevice));
    h->n_box = n_box; h->n_sph = n_sph;
    return GUSTO_OK;
}

// One Workspace per problem: in the reference every ProblemDefinition owns its env (types.jl:32-39) and Workspace(robot, env)
// is built per problem (types.jl:12-24).  The tables of all problems are concatenated in problem order.
int gusto_set_env_batch(gusto_handle h, int B, const int* n_box, const double* box, const int* n_sph, const double* sph) {
    if (!h || B < 1 || !n_box || !n_sph) return GUSTO_ERR_ARG;
    if (B > h->batch_cap) { h->err = "gusto_set_env_batch: more problems than the handle's batch_cap"; return GUSTO_ERR_ARG; }
    std::vector<int> rec((size_t)4 * B);
    size_t tb = 0, ts = 0;
    int mx = 0;
    for (int b = 0; b < B; b++) {
        if (n_box[b] < 0 || n_sph[b] < 0 || n_box[b] + n_sph[b] > 64) {
            h->err = "gusto_set_env_batch: between 0 and 64 keep-out components per problem";
            return GUSTO_ERR_ARG;
        }
        if (tb + n_box[b] > (size_t)INT_MAX / 8 || ts + n_sph[b] > (size_t)INT_MAX / 8) {
            h->err = "gusto_set_env_batch: tables too large";
            return GUSTO_ERR_ARG;
        }
        rec[4 * b] = (int)tb; rec[4 * b + 1] = n_box[b]; rec[4 * b + 2] = (int)ts; rec[4 * b + 3] = n_sph[b];
        tb += n_box[b]; ts += n_sph[b];
        mx = std::max(mx, n_box[b] + n_sph[b]);
    }
    if ((tb && !box) || (ts && !sph)) return GUSTO_ERR_ARG;
    { int rc = setter_enter(h); if (rc) return rc; }
    hipFree(h->d_box); hipFree(h->d_sph);
    h->d_box = h->d_sph = nullptr;
    if (h->d_env) { hipFree(h->d_env); h->d_env = nullptr; }
    h->env_B = 0; h->n_obs_max = 0; h->n_box = 0; h->n_sph = 0;
    HIPCHK(h, dalloc(&h->d_box, 6 * tb)); HIPCHK(h, dalloc(&h->d_sph, 4 * ts)); HIPCHK(h, dalloc(&h->d_env, (size_t)4 * B));
    if (tb) HIPCHK(h, hipMemcpy(h->d_box, box, sizeof(double) * 6 * tb, hipMemcpyHostToDevice));
    if (ts) HIPCHK(h, hipMemcpy(h->d_sph, sph, sizeof(double) * 4 * ts, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_env, rec.data(), sizeof(int) * rec.size(), hipMemcpyHostToDevice));
    h->env_B = B; h->n_obs_max = mx;
    return GUSTO_OK;
}

}  // extern "C"

static int do_init(gusto_handle h, bool straight) {
    switch (h->model) {
    case 0: return gusto_launch_init_m0(h, straight);
    case 1: return gusto_launch_init_m1(h, straight);
    case 2: return gusto_launch_init_m2(h, straight);
    case 3: return gusto_launch_init_m3(h, straight);
    case 4: return gusto_launch_init_m4(h, straight);
    case 5: return gusto_launch_init_m5(h, straight);
    case 6: return gusto_launch_init_m6(h, straight);
    }
    return GUSTO_ERR_ARG;
}
static int do_trajopt(gusto_handle h, int mode, int max_iter) {
    switch (h->model) {
    case 4: return gusto_launch_trajopt_m4(h, mode, max_iter);
    case 5: return gusto_launch_trajopt_m5(h, mode, max_iter);
    case 6: return gusto_launch_trajopt_m6(h, mode, max_iter);
    }
    h->err = "not a TrajOpt handle (gusto_create_trajopt)";
    return GUSTO_ERR_STATE;
}
// U between the caller (u_dim columns) and the device (TrajOpt handles keep u | defect per knot: m columns)
static hipError_t copy_U(gusto_handle h, double* dst, const double* src, bool to_device, hipMemcpyKind kind) {
    const size_t rows = (size_t)h->B * h->N;
    if (!h->trajopt) return hipMemcpyAsync(dst, src, sizeof(double) * rows * h->m, kind, h->stream);
    if (to_device) {
        hipError_t e = hipMemsetAsync(dst, 0, sizeof(double) * rows * h->m, h->stream);   // defects start at 0
        if (e != hipSuccess) return e;
        return hipMemcpy2DAsync(dst, sizeof(double) * h->m, src, sizeof(double) * h->m_pub, sizeof(double) * h->m_pub, rows, kind, h->stream);
    }
    return hipMemcpy2DAsync(dst, sizeof(double) * h->m_pub, src, sizeof(double) * h->m, sizeof(double) * h->m_pub, rows, kind, h->stream);
}
static int do_scp(gusto_handle h, int mode, int max_iter, int force) {
    if (h->trajopt) { h->err = "TrajOpt handle: use gusto_solve_trajopt / gusto_subproblem_trajopt"; return GUSTO_ERR_STATE; }
    switch (h->model) {
    case 0: return gusto_launch_scp_m0(h, mode, max_iter, force);
    case 1: return gusto_launch_scp_m1(h, mode, max_iter, force);
    case 2: return gusto_launch_scp_m2(h, mode, max_iter, force);
    case 3: return gusto_launch_scp_m3(h, mode, max_iter, force);
    }
    return GUSTO_ERR_ARG;
}

extern "C" {

static int set_problems_impl(gusto_handle h, int B, const double* x_init, const double* glo, const double* ghi,
                             const double* tf, const double* X0, const double* U0, hipMemcpyKind kind) {
    if (!h || B < 1 || B > h->batch_cap || !x_init || !glo || !ghi || !tf || ((X0 == nullptr) != (U0 == nullptr))) {
        if (h) h->err = "gusto_set_problems: bad argument";
        return GUSTO_ERR_ARG;
    }
    HIPCHK(h, hipSetDevice(h->device));
    { int rcw = gusto_complete(h); if (rcw) return rcw; }
    h->sched_err = 0;   // (a latched scheduler error ends here: every problem is set again)
    const size_t n = h->n, m = h->m, N = h->N;
    h->B = B;
    HIPCHK(h, hipMemcpyAsync(h->d_xinit, x_init, sizeof(double) * B * n, kind, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_glo, glo, sizeof(double) * B * n, kind, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_ghi, ghi, sizeof(double) * B * n, kind, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_tf, tf, sizeof(double) * B, kind, h->stream));
    if (X0) {
        HIPCHK(h, hipMemcpyAsync(h->d_X, X0, sizeof(double) * B * N * n, kind, h->stream));
        HIPCHK(h, copy_U(h, h->d_U, U0, true, kind));
    }
    int rc = do_init(h, X0 == nullptr);
    if (rc) return rc;
    h->have_problems = true; h->have_shoot = false;
    h->n_active = -1;   // (gusto_set_active belongs to the problems it was set for)
    return GUSTO_OK;
}

int gusto_set_problems(gusto_handle h, int B, const double* x_init, const double* glo, const double* ghi, const double* tf,
                       const double* X0, const double* U0) {
    return set_problems_impl(h, B, x_init, glo, ghi, tf, X0, U0, hipMemcpyHostToDevice);
}
int gusto_set_problems_dev(gusto_handle h, int B, const double* x_init, const double* glo, const double* ghi,
                           const double* tf, const double* X0, const double* U0) {
    return set_problems_impl(h, B, x_init, glo, ghi, tf, X0, U0, hipMemcpyDeviceToDevice);
}

int gusto_solve(gusto_handle h, int max_iter, int force) {
    if (!h || max_iter < 0) return GUSTO_ERR_ARG;
    if (!h->have_problems) { h->err = "gusto_solve: call gusto_set_problems first"; return GUSTO_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    int rc = gusto_finish(h);   // (a batch that lost a problem to a scheduler error is not resumed: gusto_set_problems first)
    if (rc) return rc;
    rc = do_scp(h, 0, max_iter, force ? 1 : 0);
    return rc ? rc : gusto_finish(h);
}

int gusto_solve_async(gusto_handle h, int max_iter, int force) {
    if (!h || max_iter < 0) return GUSTO_ERR_ARG;
    if (!h->have_problems) { h->err = "gusto_solve_async: call gusto_set_problems first"; return GUSTO_ERR_STATE; }
    HIPCHK(h, hipSetDevice(h->device));
    int rc = gusto_finish(h);
    if (rc) return rc;
    return do_scp(h, 0, max_iter, force ? 1 : 0);
}

// Which problems of the batch the following gusto_solve / gusto_solve_async / gusto_shoot calls work on.  The reference's
// drivers loop per problem -- solve_SCPshooting! runs another SCP iteration and another shooting attempt only `while
// !SCPS.converged && SCPS.iterations < max_iter` (traj_opt.jl:23) --; a batch needs that condition per problem.
int gusto_set_active(gusto_handle h, const int* active) {
    if (!h) return GUSTO_ERR_ARG;
    if (h->trajopt) { h->err = "gusto_set_active: TrajOpt handle (solve_trajopt_jump! has no resume)"; return GUSTO_ERR_STATE; }
    if (!h->have_problems) { h->err = "gusto_set_active: call gusto_set_problems first"; return GUSTO_ERR_STATE; }
    { int rc = setter_enter(h); if (rc) return rc; }
    if (!active) { h->n_active = -1; return GUSTO_OK; }
    std::vector<int> buf((size_t)2 * h->batch_cap, 0);
    int na = 0;
    for (int b = 0; b < h->B; b++) {
        buf[b] = active[b] != 0;
        if (active[b]) buf[(size_t)h->batch_cap + na++] = b;
    }
    if (!h->d_active) HIPCHK(h, dalloc(&h->d_active, (size_t)2 * h->batch_cap));
    HIPCHK(h, hipMemcpy(h->d_active, buf.data(), sizeof(int) * buf.size(), hipMemcpyHostToDevice));
    h->n_active = na;
    return GUSTO_OK;
}

int gusto_wait(gusto_handle h) {
    if (!h) return GUSTO_ERR_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    return gusto_finish(h);
}

// development hook (GUSTO_PROFILE builds): per-problem phase cycle counters [B][16]; not part of gusto_hip.h
int gusto_dev_get_prof(gusto_handle h, long long* out) {
    if (!h || !h->d_prof) return GUSTO_ERR_STATE;
    HIPCHK(h, hipMemcpy(out, h->d_prof, sizeof(long long) * (size_t)h->B * PROF_N, hipMemcpyDeviceToHost));
    return GUSTO_OK;
}

// development hook: shape of the last launch (persistent workgroups, dynamic LDS per workgroup, workgroups per CU)
int gusto_dev_launch_info(gusto_handle h, int* slots, int* lds_bytes, int* per_cu) {
    if (!h) return GUSTO_ERR_ARG;
    if (h->slots <= 0) { h->err = "gusto_dev_launch_info: nothing launched yet"; return GUSTO_ERR_STATE; }
    if (slots) *slots = h->slots;
    if (lds_bytes) *lds_bytes = h->lds_bytes;
    if (per_cu) *per_cu = h->per_cu;
    return GUSTO_OK;
}

int gusto_dev_workspace_bytes(gusto_handle h, long long* bytes) {
    if (!h || !bytes) return GUSTO_ERR_ARG;
    *bytes = (long long)(h->ws_doubles * sizeof(double));
    return GUSTO_OK;
}

int gusto_last_solve_ms(gusto_handle h, double* ms) {
    if (h) { int rcw = gusto_finish(h); if (rcw) return rcw; }
    if (!h || !ms) return GUSTO_ERR_ARG;
    *ms = h->last_ms;
    return GUSTO_OK;
}

int gusto_get_traj(gusto_handle h, double* X, double* U) {
    if (h) { int rcw = gusto_finish(h); if (rcw) return rcw; }
    if (!h || !h->have_problems) return GUSTO_ERR_STATE;
    HIPCHK(h, hipSetDevice(h->device));
    if (X) HIPCHK(h, hipMemcpy(X, h->d_X, sizeof(double) * h->B * h->N * h->n, hipMemcpyDeviceToHost));
    if (U) {
        HIPCHK(h, copy_U(h, U, h->d_U, false, hipMemcpyDeviceToHost));
        HIPCHK(h, hipStreamSynchronize(h->stream));
    }
    return GUSTO_OK;
}
int gusto_get_traj_dev(gusto_handle h, const double** X, const double** U) {
    if (h) { int rcw = gusto_finish(h); if (rcw) return rcw; }
    if (!h) return GUSTO_ERR_ARG;
    if (!h->have_problems) { h->err = "gusto_get_traj_dev: call gusto_set_problems first"; return GUSTO_ERR_STATE; }
    if (X) *X = h->d_X;
    if (U) {
        if (h->trajopt) {   // device rows are (u | defect), pitch u_dim + x_dim: hand out a compact [B][N][u_dim] copy
            HIPCHK(h, hipSetDevice(h->device));
            if (!h->d_Upub) HIPCHK(h, dalloc(&h->d_Upub, (size_t)h->batch_cap * h->N * h->m_pub));
            HIPCHK(h, copy_U(h, h->d_Upub, h->d_U, false, hipMemcpyDeviceToDevice));
            HIPCHK(h, hipStreamSynchronize(h->stream));
            *U = h->d_Upub;
        } else *U = h->d_U;
    }
    return GUSTO_OK;
}

// The final gather of a multi-GPU run below the host language (SURVEY.md 8(b) threading row, 8(e); north_star: "RCCL only
// for the batch split and final gather").  One process, one handle per GPU: every shard goes to the GPU of `dst` with ONE
// direct peer copy over xGMI (hipMemcpyPeerAsync: one hop per source, all links busy -- the fan-in 8(e) asks for, not a
// ring), each enqueued on its SOURCE handle's stream right behind that shard's solve; dst's stream waits for their events.
int gusto_gather_peer(gusto_handle dst, int n_src, const gusto_handle* src, const double** X_dev, const double** U_dev,
                      double* X_host, double* U_host, int* B_total) {
    if (!dst || n_src < 1 || !src) return GUSTO_ERR_ARG;
    size_t tot = 0;
    for (int i = 0; i < n_src; i++) {
        gusto_handle q = src[i];
        if (!q || !q->have_problems) { dst->err = "gusto_gather_peer: a source handle holds no problems"; return GUSTO_ERR_STATE; }
        if (q->model_pub != dst->model_pub || q->N != dst->N || q->trajopt != dst->trajopt) {
            dst->err = "gusto_gather_peer: every handle must hold the same model, algorithm and N";
            return GUSTO_ERR_ARG;
        }
        tot += q->B;
    }
    HIPCHK(dst, hipSetDevice(dst->device));
    // (dst's own solve, if it is a source, is completed like the others below; a pending solve of a dst that is NOT among the
    // sources is completed here)
    bool dst_is_src = false;
    for (int i = 0; i < n_src; i++) dst_is_src = dst_is_src || src[i] == dst;
    if (!dst_is_src) { int rc = gusto_finish(dst); if (rc) return rc; }
    const size_t N = dst->N, n = dst->n, mp = dst->m_pub;
    if (tot > dst->gather_cap) {
        if (dst->d_gX) hipFree(dst->d_gX);
        if (dst->d_gU) hipFree(dst->d_gU);
        dst->d_gX = dst->d_gU = nullptr; dst->gather_cap = 0;
        HIPCHK(dst, dalloc(&dst->d_gX, tot * N * n)); HIPCHK(dst, dalloc(&dst->d_gU, tot * N * mp));
        dst->gather_cap = tot;
    }
    // Every shard's copy is enqueued on ITS OWN handle's stream, right behind that shard's solve: no host-side wait for any
    // solve before the first copy is queued, and the copies of different sources run side by side -- one xGMI link per source
    // GPU, the direct fan-in of SURVEY.md 8(e) with all links busy -- instead of one after the other on dst's stream.
    size_t at = 0;
    for (int i = 0; i < n_src; i++) {
        gusto_handle q = src[i];
        HIPCHK(q, hipSetDevice(q->device));
        if (q->device != dst->device) {   // (best effort: without peer access the runtime stages the copy)
            int can = 0;
            if (hipDeviceCanAccessPeer(&can, q->device, dst->device) == hipSuccess && can) {
                (void)hipDeviceEnablePeerAccess(dst->device, 0);
                (void)hipGetLastError();
            }
        }
        const double* su = q->d_U;
        if (q->trajopt) {     // (device rows are (u | defect): compact them on the source GPU first, on the same stream)
            if (!q->d_Upub) HIPCHK(q, dalloc(&q->d_Upub, (size_t)q->batch_cap * q->N * q->m_pub));
            HIPCHK(q, copy_U(q, q->d_Upub, q->d_U, false, hipMemcpyDeviceToDevice));
            su = q->d_Upub;
        }
        HIPCHK(q, hipMemcpyPeerAsync(dst->d_gX + at * N * n, dst->device, q->d_X, q->device, sizeof(double) * q->B * N * n, q->stream));
        HIPCHK(q, hipMemcpyPeerAsync(dst->d_gU + at * N * mp, dst->device, su, q->device, sizeof(double) * q->B * N * mp, q->stream));
        if (!q->ev_gather) HIPCHK(q, hipEventCreateWithFlags(&q->ev_gather, hipEventDisableTiming));
        HIPCHK(q, hipEventRecord(q->ev_gather, q->stream));
        at += q->B;
    }
    HIPCHK(dst, hipSetDevice(dst->device));
    for (int i = 0; i < n_src; i++) HIPCHK(dst, hipStreamWaitEvent(dst->stream, src[i]->ev_gather, 0));
    if (X_host) HIPCHK(dst, hipMemcpyAsync(X_host, dst->d_gX, sizeof(double) * tot * N * n, hipMemcpyDeviceToHost, dst->stream));
    if (U_host) HIPCHK(dst, hipMemcpyAsync(U_host, dst->d_gU, sizeof(double) * tot * N * mp, hipMemcpyDeviceToHost, dst->stream));
    HIPCHK(dst, hipStreamSynchronize(dst->stream));
    // the solves are complete now (their streams reached the copies): take their times and any latched scheduler error
    int rc_all = GUSTO_OK;
    for (int i = 0; i < n_src; i++) {
        gusto_handle q = src[i];
        HIPCHK(q, hipSetDevice(q->device));
        const int rc = gusto_finish(q);
        if (rc && !rc_all) { rc_all = rc; dst->err = "gusto_gather_peer: source: " + q->err; }
    }
    HIPCHK(dst, hipSetDevice(dst->device));
    if (rc_all) return rc_all;
    if (X_dev) *X_dev = dst->d_gX;
    if (U_dev) *U_dev = dst->d_gU;
    if (B_total) *B_total = (int)tot;
    return GUSTO_OK;
}

int gusto_get_status(gusto_handle h, int* iterations, int* converged, int* successful, int* stop, int* ipm) {
    if (h) { int rcw = gusto_finish(h); if (rcw) return rcw; }
    if (!h || !h->have_problems) return GUSTO_ERR_STATE;
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<int> st((size_t)h->B * ST_NI);
    HIPCHK(h, hipMemcpy(st.data(), h->d_sti, sizeof(int) * st.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < h->B; b++) {
        if (iterations) iterations[b] = st[(size_t)b * ST_NI + ST_ITER];
        if (converged) converged[b] = st[(size_t)b * ST_NI + ST_CONV];
        if (successful) successful[b] = st[(size_t)b * ST_NI + ST_SUCC];
        if (stop) stop[b] = st[(size_t)b * ST_NI + ST_STOP];
        if (ipm) ipm[b] = st[(size_t)b * ST_NI + ST_IPM];
    }
    return GUSTO_OK;
}

int gusto_get_dual(gusto_handle h, double* dual) {
    if (h) { int rcw = gusto_finish(h); if (rcw) return rcw; }
    if (!h || !dual || !h->have_problems) return GUSTO_ERR_STATE;
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<double> sd((size_t)h->B * SD_ND);
    HIPCHK(h, hipMemcpy(sd.data(), h->d_std, sizeof(double) * sd.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < h->B; b++)
        for (int i = 0; i < h->n; i++) dual[(size_t)b * h->n + i] = sd[(size_t)b * SD_ND + SD_DUAL + i];
    return GUSTO_OK;
}

int gusto_get_hist_cap(gusto_handle h, int* hist_cap) {
    if (!h || !hist_cap) return GUSTO_ERR_ARG;
    *hist_cap = h->hist_cap;
    return GUSTO_OK;
}

int gusto_get_history(gusto_handle h, gusto_history* o) {
    if (h && h->trajopt) { h->err = "gusto_get_history: TrajOpt handle, use gusto_get_trajopt_history"; return GUSTO_ERR_STATE; }
    if (h) { int rcw = gusto_finish(h); if (rcw) return rcw; }
    if (!h || !o || !h->have_problems) return GUSTO_ERR_STATE;
    // o->hist_cap is the row capacity of the CALLER's arrays; rows are written with that pitch
    if (o->hist_cap < h->hist_cap) {
        h->err = "gusto_get_history: hist_cap of the output arrays is smaller than the handle's (gusto_get_hist_cap)";
        return GUSTO_ERR_ARG;
    }
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<int> st((size_t)h->B * ST_NI);
    HIPCHK(h, hipMemcpy(st.data(), h->d_sti, sizeof(int) * st.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < h->B; b++) {
        if (o->n_hist) o->n_hist[b] = st[(size_t)b * ST_NI + ST_NHIST];
        if (o->nJ) o->nJ[b] = st[(size_t)b * ST_NI + ST_NJ];
        if (o->n_rho) o->n_rho[b] = st[(size_t)b * ST_NI + ST_NRHO];
    }
    const size_t H = h->hist_cap, Ho = o->hist_cap;
#define CPD(dst, src) if (dst) HIPCHK(h, hipMemcpy2D(dst, sizeof(*(dst)) * Ho, src, sizeof(*(dst)) * H, sizeof(*(dst)) * H, h->B, hipMemcpyDeviceToHost))
    CPD(o->J_true, h->d_Jt); CPD(o->J_full, h->d_Jf); CPD(o->convergence_measure, h->d_conv); CPD(o->Delta, h->d_Delta);
    CPD(o->omega, h->d_omega); CPD(o->rho, h->d_rho); CPD(o->accept_solution, h->d_acc); CPD(o->scp_status, h->d_scp);
    CPD(o->solver_status, h->d_sol); CPD(o->trust_region_satisfied, h->d_tr); CPD(o->convex_ineq_satisfied, h->d_cvx);
    CPD(o->ipm_iters, h->d_ipm);
#undef CPD
    return GUSTO_OK;
}

// SCPParam_GuSTO supplied by the caller (scp_gusto.jl:60 keeps a param.alg that is already defined): overwrite the
// LAST entries of Delta_vec / omega_vec of every problem, i.e. the trust region and penalty the next trip uses
int gusto_set_trust_state(gusto_handle h, const double* Delta, const double* omega) {
    if (!h || (!Delta && !omega)) return GUSTO_ERR_ARG;
    if (h->trajopt) { h->err = "gusto_set_trust_state: TrajOpt handle (its trust region s and penalty mu follow SCPParam_TrajOpt)"; return GUSTO_ERR_STATE; }
    if (!h->have_problems) { h->err = "gusto_set_trust_state: call gusto_set_problems first"; return GUSTO_ERR_STATE; }
    { int rc = setter_enter(h); if (rc) return rc; }
    std::vector<int> st((size_t)h->B * ST_NI);
    HIPCHK(h, hipMemcpy(st.data(), h->d_sti, sizeof(int) * st.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < h->B; b++) {
        const size_t at = (size_t)b * h->hist_cap + st[(size_t)b * ST_NI + ST_NHIST] - 1;
        if (Delta) HIPCHK(h, hipMemcpy(h->d_Delta + at, Delta + b, sizeof(double), hipMemcpyHostToDevice));
        if (omega) HIPCHK(h, hipMemcpy(h->d_omega + at, omega + b, sizeof(double), hipMemcpyHostToDevice));
    }
    return GUSTO_OK;
}

int gusto_subproblem(gusto_handle h, int B, const double* Xp, const double* Up, const double* Delta, const double* omega,
                     const double* toggle, double* Xn, double* Un, double* obj, int* status, int* iters) {
    // (before any copy: on a TrajOpt handle h->m is u_dim + x_dim and the caller's [B][N][u_dim] buffer would be over-read)
    if (h && h->trajopt) { h->err = "TrajOpt handle: use gusto_solve_trajopt / gusto_subproblem_trajopt"; return GUSTO_ERR_STATE; }
    if (!h || !h->have_problems || B != h->B || !Xp || !Up || !Delta || !omega || !toggle) {
        if (h) h->err = "gusto_subproblem: call gusto_set_problems with the same B first";
        return GUSTO_ERR_STATE;
    }
    HIPCHK(h, hipSetDevice(h->device));
    { int rcw = gusto_finish(h); if (rcw) return rcw; }
    const size_t n = h->n, m = h->m, N = h->N;
    HIPCHK(h, hipMemcpy(h->d_X, Xp, sizeof(double) * B * N * n, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_U, Up, sizeof(double) * B * N * m, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_subD, Delta, sizeof(double) * B, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_subW, omega, sizeof(double) * B, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpy(h->d_subT, toggle, sizeof(double) * B, hipMemcpyHostToDevice));
    int rc = do_scp(h, 1, 0, 0);
    if (!rc) rc = gusto_finish(h);
    if (rc) return rc;
    if (Xn) HIPCHK(h, hipMemcpy(Xn, h->d_subX, sizeof(double) * B * N * n, hipMemcpyDeviceToHost));
    if (Un) HIPCHK(h, hipMemcpy(Un, h->d_subU, sizeof(double) * B * N * m, hipMemcpyDeviceToHost));
    if (obj) HIPCHK(h, hipMemcpy(obj, h->d_subObj, sizeof(double) * B, hipMemcpyDeviceToHost));
    if (status) HIPCHK(h, hipMemcpy(status, h->d_subSt, sizeof(int) * B, hipMemcpyDeviceToHost));
    if (iters) HIPCHK(h, hipMemcpy(iters, h->d_subIt, sizeof(int) * B, hipMemcpyDeviceToHost));
    return GUSTO_OK;
}

int gusto_set_trajopt_params(gusto_handle h, const gusto_trajopt_params* tp) {
    if (!h || !tp || !h->trajopt) return GUSTO_ERR_ARG;
    { int rc = setter_enter(h); if (rc) return rc; }
    h->tp = *tp;
    return GUSTO_OK;
}

static int solve_trajopt_impl(gusto_handle h, int max_iter, bool wait, const char* who) {
    if (!h || max_iter < 0) return GUSTO_ERR_ARG;
    if (!h->trajopt) { h->err = std::string(who) + ": not a TrajOpt handle (gusto_create_trajopt)"; return GUSTO_ERR_STATE; }
    if (!h->have_problems) { h->err = std::string(who) + ": call gusto_set_problems first"; return GUSTO_ERR_STATE; }
    const int total = h->tp.max_penalty_iteration * h->tp.max_convex_iteration * h->tp.max_trust_iteration;
    if (h->hist_cap < 2 * std::min(total, max_iter) + 8) {
        h->err = std::string(who) + ": hist_cap of the handle is below 2 * min(max_iter, max_penalty * max_convex * max_trust) + 8";
        return GUSTO_ERR_ARG;
    }
    HIPCHK(h, hipSetDevice(h->device));
    int rc = gusto_finish(h);
    if (rc) return rc;
    rc = do_trajopt(h, 0, max_iter);
    return (rc || !wait) ? rc : gusto_finish(h);
}
int gusto_solve_trajopt(gusto_handle h, int max_iter) { return solve_trajopt_impl(h, max_iter, true, "gusto_solve_trajopt"); }
// the TrajOpt counterpart of gusto_solve_async: the one launch of the batch is enqueued on the handle's stream and the call
// returns; gusto_wait (or any getter) completes it.  Shards on several GPUs, or two handles on one, then run side by side.
int gusto_solve_trajopt_async(gusto_handle h, int max_iter) { return solve_trajopt_impl(h, max_iter, false, "gusto_solve_trajopt_async"); }

int gusto_get_trajopt_history(gusto_handle h, gusto_trajopt_history* o) {
    if (h) { int rcw = gusto_finish(h); if (rcw) return rcw; }
    if (!h || !o || !h->trajopt || !h->have_problems) return GUSTO_ERR_STATE;
    if (o->hist_cap < h->hist_cap) { h->err = "gusto_get_trajopt_history: hist_cap of the output arrays is smaller than the handle's"; return GUSTO_ERR_ARG; }
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<int> st((size_t)h->B * ST_NI);
    HIPCHK(h, hipMemcpy(st.data(), h->d_sti, sizeof(int) * st.size(), hipMemcpyDeviceToHost));
    for (int b = 0; b < h->B; b++) {
        const int* q = st.data() + (size_t)b * ST_NI;
        if (o->n_solves) o->n_solves[b] = q[ST_ITER];
        if (o->n_mu) o->n_mu[b] = q[ST_NMU];
        if (o->n_xtol) o->n_xtol[b] = q[ST_NXTOL];
        if (o->n_ftol) o->n_ftol[b] = q[ST_NFTOL];
        if (o->n_ctol) o->n_ctol[b] = q[ST_NCTOL];
    }
    const size_t H = h->hist_cap, Ho = o->hist_cap;
#define CPD(dst, src) if (dst) HIPCHK(h, hipMemcpy2D(dst, sizeof(*(dst)) * Ho, src, sizeof(*(dst)) * H, sizeof(*(dst)) * H, h->B, hipMemcpyDeviceToHost))
    CPD(o->rho_vec, h->d_rho); CPD(o->s_vec, h->d_Delta); CPD(o->mu_vec, h->d_to_mu); CPD(o->xtol_vec, h->d_to_xtol);
    CPD(o->ftol_vec, h->d_to_ftol); CPD(o->ctol_vec, h->d_to_ctol); CPD(o->J_true, h->d_Jt); CPD(o->J_full, h->d_Jf);
    CPD(o->convergence_measure, h->d_conv); CPD(o->solver_status, h->d_sol); CPD(o->ipm_iters, h->d_ipm);
#undef CPD
    return GUSTO_OK;
}

int gusto_subproblem_trajopt(gusto_handle h, int B, const double* Xp, const double* Up, const double* mu, const double* s,
                             double* Xn, double* Un, double* Dn, double* obj, int* status, int* iters) {
    if (!h || !h->trajopt || !h->have_problems || B != h->B || !Xp || !Up || !mu || !s) {
        if (h) h->err = "gusto_subproblem_trajopt: TrajOpt handle and gusto_set_problems with the same B first";
        return GUSTO_ERR_STATE;
    }
    HIPCHK(h, hipSetDevice(h->device));
    { int rcw = gusto_finish(h); if (rcw) return rcw; }
    const size_t n = h->n, m = h->m, mp = h->m_pub, N = h->N, rows = (size_t)B * N;
    HIPCHK(h, hipMemcpy(h->d_X, Xp, sizeof(double) * rows * n, hipMemcpyHostToDevice));
    HIPCHK(h, copy_U(h, h->d_U, Up, true, hipMemcpyHostToDevice));
    HIPCHK(h, hipMemcpyAsync(h->d_subD, s, sizeof(double) * B, hipMemcpyHostToDevice, h->stream));
    HIPCHK(h, hipMemcpyAsync(h->d_subW, mu, sizeof(double) * B, hipMemcpyHostToDevice, h->stream));
    int rc = do_trajopt(h, 1, 0);
    if (!rc) rc = gusto_finish(h);
    if (rc) return rc;
    if (Xn) HIPCHK(h, hipMemcpy(Xn, h->d_subX, sizeof(double) * rows * n, hipMemcpyDeviceToHost));
    if (Un) HIPCHK(h, hipMemcpy2D(Un, sizeof(double) * mp, h->d_subU, sizeof(double) * m, sizeof(double) * mp, rows, hipMemcpyDeviceToHost));
    if (Dn) HIPCHK(h, hipMemcpy2D(Dn, sizeof(double) * n, h->d_subU + mp, sizeof(double) * m, sizeof(double) * n, rows, hipMemcpyDeviceToHost));
    if (obj) HIPCHK(h, hipMemcpy(obj, h->d_subObj, sizeof(double) * B, hipMemcpyDeviceToHost));
    if (status) HIPCHK(h, hipMemcpy(status, h->d_subSt, sizeof(int) * B, hipMemcpyDeviceToHost));
    if (iters) HIPCHK(h, hipMemcpy(iters, h->d_subIt, sizeof(int) * B, hipMemcpyDeviceToHost));
    return GUSTO_OK;
}


}  // extern "C"
