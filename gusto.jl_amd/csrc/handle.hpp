// handle.hpp -- the opaque gusto_handle and the per-model launch entry points (one translation unit per model).
#pragma once
#include <hip/hip_runtime.h>

#include <string>

#include "common.hpp"

struct gusto_handle_s {
    int model = 0, n = 0, m = 0, N = 0, batch_cap = 0, hist_cap = 0, device = 0, B = 0;
    // TrajOpt handles (gusto_create_trajopt): `model` is the internal variant (common.hpp: GUSTO_TO_*), `m` its control
    // dimension u_dim + x_dim (u | defect); the C ABI moves U with the model's u_dim columns, `m_pub`
    bool trajopt = false;
    int m_pub = 0, model_pub = 0;
    gusto_trajopt_params tp{};
    double *d_to_mu = nullptr, *d_to_xtol = nullptr, *d_to_ftol = nullptr, *d_to_ctol = nullptr;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    gusto_scp_params sp{};
    gusto_model_params mp{};
    gusto_ipm_opts io{};
    int n_box = 0, n_sph = 0;
    double *d_box = nullptr, *d_sph = nullptr;
    // gusto_set_env_batch: one keep-out set per problem -- d_box / d_sph hold the concatenated tables, d_env the
    // (box offset, n_box, sphere offset, n_sph) record of every problem, env_B their number, n_obs_max the largest count
    int* d_env = nullptr;
    int env_B = 0, n_obs_max = 0;
    double *d_X = nullptr, *d_U = nullptr, *d_xinit = nullptr, *d_glo = nullptr, *d_ghi = nullptr, *d_tf = nullptr;
    int* d_sti = nullptr;
    double* d_std = nullptr;
    double *d_Jt = nullptr, *d_Jf = nullptr, *d_conv = nullptr, *d_Delta = nullptr, *d_omega = nullptr, *d_rho = nullptr;
    int *d_acc = nullptr, *d_scp = nullptr, *d_sol = nullptr, *d_tr = nullptr, *d_cvx = nullptr, *d_ipm = nullptr;
    double* d_ws = nullptr;
    long long* d_prof = nullptr;
    size_t ws_doubles = 0;
    double *d_subD = nullptr, *d_subW = nullptr, *d_subT = nullptr, *d_subX = nullptr, *d_subU = nullptr, *d_subObj = nullptr;
    int *d_subSt = nullptr, *d_subIt = nullptr;
    hipEvent_t ev0 = nullptr, ev1 = nullptr;
    hipEvent_t ev_gather = nullptr;   // gusto_gather_peer: this shard's copy to the gathering GPU has been enqueued up to here
    double last_ms = 0.0;
    bool pending = false;  // a gusto_solve_async launch has not been waited for yet
    int probe_iters = 2, probe_min_batch = 2048;  // longest-first schedule (gusto_set_schedule)
    bool sched_forced = false;                    // gusto_set_schedule was called: the caller's choice overrides the model default
    int* d_order = nullptr;   // waiting lists of the scheduler, [SCHED_LEVELS][probe_iters * batch_cap]
    size_t order_ints = 0;
    int* d_queue = nullptr;   // work-queue heads, one per launch of a gusto_solve call
    int* d_sched_ord = nullptr;   // [2][batch_cap]: difficulty bucket and hand-out order of the fresh problems (hardest first)
    int slots = 0;            // resident workgroups the last launch used (persistent kernel)
    int lds_bytes = 0, per_cu = 0;   // ... its dynamic LDS per workgroup and workgroups per CU (gusto_dev_launch_info)
    int sched_init[gusto::SQ_WORDS] = {0};   // initial scheduler words of a launch (host side of an async copy)
    bool have_problems = false, have_shoot = false;
    int decomposition = 0;         // gusto_set_decomposition: 0 auto, 1 a wave per problem, 2 a lane per problem (lane.hpp)
    int waves = 0;                 // waves per problem of the GuSTO kernel (0 = one per 64 knots; development builds: GUSTO_DEV_WAVES)
    // gusto_set_active: the problems the next gusto_solve calls iterate (n_active < 0: all of them); d_active = the mask [B]
    // (gusto_shoot reads it), d_active + batch_cap = the list of active problems (the hand-out order of the launch)
    int* d_active = nullptr;
    int n_active = -1;
    int sched_err = 0;             // latched scheduler error of the last solve (gusto_finish): getters and solves fail until the next set_problems
    int* h_sched_err = nullptr;    // pinned host word the error flag is copied to on the handle's stream, before the stream is waited for
    double *d_gX = nullptr, *d_gU = nullptr;   // gusto_gather_peer: the shards of several handles, one after the other, on this handle's GPU
    size_t gather_cap = 0;                     // ... capacity in problems
    double* d_Upub = nullptr;      // TrajOpt handles: U compacted to the public [B][N][u_dim] layout for gusto_get_traj_dev
    // indirect shooting (shoot.hip): trajectories, converged costates, seeds, residuals, status, Newton iterations
    double *d_shX = nullptr, *d_shU = nullptr, *d_shP = nullptr, *d_shP0 = nullptr, *d_shRes = nullptr;
    double *d_shXt = nullptr, *d_shUt = nullptr;   // knot-major staging of the shooting trajectories ([N][n][B])
    int *d_shSt = nullptr, *d_shIt = nullptr, *d_shList = nullptr;
    std::string err;
};

extern thread_local std::string g_err;

#define HIPCHK(h, call)                                                                        \
    do {                                                                                       \
        hipError_t e_ = (call);                                                                \
        if (e_ != hipSuccess) {                                                                \
            std::string msg_ = std::string(#call) + ": " + hipGetErrorString(e_);             \
            if (h) (h)->err = msg_;                                                            \
            g_err = msg_;                                                                      \
            return GUSTO_ERR_HIP;                                                              \
        }                                                                                      \
    } while (0)

template <class Tp> static hipError_t dalloc(Tp** p, size_t count) {
    return hipMalloc(reinterpret_cast<void**>(p), (count ? count : 1) * sizeof(Tp));
}


// completes an enqueued solve: blocks on the handle's stream and takes the kernel time from its events
static inline int gusto_sched_err_rc(gusto_handle h) {
    if (!h->sched_err) return GUSTO_OK;
    h->err = h->sched_err == 1 ? "scheduler: a claimed waiting-list entry never arrived (problem lost); set the problems again"
                               : "scheduler: workgroups gave up waiting for problems still in their probing slices; set the problems again";
    return GUSTO_ERR_STATE;
}
// The device-side scheduler reports a problem it lost instead of leaving it half-solved (scp.hpp: sched_pop).  The flag is
// copied on the handle's OWN stream into pinned memory right behind the kernel (launch.hpp), so reading it here needs no
// blocking copy on the null stream (which would serialise with the other handle of an overlapped pair).  The error is
// LATCHED until gusto_set_problems: every getter and every further solve fails with it (a batch that lost a problem is
// not resumed), the setters (environment, parameters, stream, ...) do not -- they may come before or after the problems.
// gusto_complete: waits for an enqueued solve and latches its error; GUSTO_ERR_HIP only.
static inline int gusto_complete(gusto_handle h) {
    if (!h->pending) return GUSTO_OK;
    HIPCHK(h, hipStreamSynchronize(h->stream));
    float ms = 0;
    HIPCHK(h, hipEventElapsedTime(&ms, h->ev0, h->ev1));
    h->last_ms = ms;
    h->pending = false;
    if (h->h_sched_err && *h->h_sched_err) h->sched_err = *h->h_sched_err;
    return GUSTO_OK;
}
// gusto_finish: gusto_complete, then the latched scheduler error (getters, gusto_wait, gusto_solve*)
static inline int gusto_finish(gusto_handle h) {
    const int rc = gusto_complete(h);
    return rc ? rc : gusto_sched_err_rc(h);
}
// enqueued behind a solve kernel on the handle's stream: the scheduler's error word -> pinned host memory
static inline hipError_t gusto_fetch_sched_err(gusto_handle h) {
    if (!h->h_sched_err) {
        hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&h->h_sched_err), sizeof(int), hipHostMallocDefault);
        if (e != hipSuccess) return e;
    }
    *h->h_sched_err = 0;
    if (!h->d_queue) return hipSuccess;
    return hipMemcpyAsync(h->h_sched_err, h->d_queue + gusto::SQ_ERR, sizeof(int), hipMemcpyDeviceToHost, h->stream);
}

// defined in model_<id>.hip (the scp launch only enqueues; gusto_finish completes it)
int gusto_launch_init_m0(gusto_handle h, bool straight);
int gusto_launch_init_m1(gusto_handle h, bool straight);
int gusto_launch_init_m2(gusto_handle h, bool straight);
int gusto_launch_init_m3(gusto_handle h, bool straight);
int gusto_launch_scp_m0(gusto_handle h, int mode, int max_iter, int force);
int gusto_launch_scp_m1(gusto_handle h, int mode, int max_iter, int force);
int gusto_launch_scp_m2(gusto_handle h, int mode, int max_iter, int force);
int gusto_launch_scp_m3(gusto_handle h, int mode, int max_iter, int force);
int gusto_launch_init_m4(gusto_handle h, bool straight);
int gusto_launch_init_m5(gusto_handle h, bool straight);
int gusto_launch_trajopt_m4(gusto_handle h, int mode, int max_iter);
int gusto_launch_trajopt_m5(gusto_handle h, int mode, int max_iter);
int gusto_launch_init_m6(gusto_handle h, bool straight);
int gusto_launch_trajopt_m6(gusto_handle h, int mode, int max_iter);
