// launch.hpp -- kernel launch templates, instantiated once per model in model_<id>.hip
#pragma once
#include <algorithm>
#include <cstdlib>
#include <cstdio>
#include <cstring>

#include "handle.hpp"
#include "scp.hpp"
// the lane-per-problem kernel of dubins_car (lane.hpp: parity-green, 4.3x slower than the wave kernel at config 3) is built only
// with -DGUSTO_WITH_LANE; without it gusto_set_decomposition(GUSTO_DECOMP_LANE) is refused
#ifdef GUSTO_WITH_LANE
#include "lane.hpp"
#endif

using namespace gusto;

// Development knobs (occupancy / scheduling experiments, tools/build_variant.sh) are environment variables of a
// -DGUSTO_DEV_KNOBS build only: the shipped library reads none, its launch shape follows from the handle alone.
#ifdef GUSTO_DEV_KNOBS
static inline const char* dev_env(const char* k) { return getenv(k); }
#else
static inline const char* dev_env(const char*) { return nullptr; }
#endif

// waves per problem of the GuSTO kernel: one per 64 knots unless the caller (or GUSTO_DEV_WAVES) asks for more -- the generic
// multi-wave phases then run with the extra waves splitting the entry-parallel sweeps (a latency / throughput trade for
// batches smaller than the GPU)
static inline int launch_waves(gusto_handle h) {
    int w = h->waves;
    if (const char* e = dev_env("GUSTO_DEV_WAVES")) w = atoi(e);
    const int need = (h->N + 63) / 64;
    return std::min(4, std::max(w, need));
}
// ---- kernel dispatch ---------------------------------------------------------------------------------
template <int MODEL> static int fill_params(gusto_handle h, KParams& P, int B, bool need_env = true) {
    using T = MT<MODEL>;
    memset(&P, 0, sizeof(P));
    P.N = h->N; P.B = B; P.n_fresh = B; P.n_box = h->n_box; P.n_sph = h->n_sph;
    P.n_obs = T::HAS_OBS ? h->n_box + h->n_sph : 0;
    if (T::HAS_OBS && h->d_env) {   // one keep-out set per problem (gusto_set_env_batch): n_obs sizes the slots, the records say the rest
        if (need_env && h->env_B != B) {
            h->err = "gusto_set_env_batch described a different number of problems than gusto_set_problems";
            return GUSTO_ERR_STATE;
        }
        P.n_obs = h->n_obs_max; P.n_box = 0; P.n_sph = 0; P.env = h->d_env;
    }
    P.hist_cap = h->hist_cap;
    P.sp = h->sp; P.mp = h->mp; P.io = h->io;
    warm_defaults(MODEL, P.io);
    P.box = h->d_box; P.sph = h->d_sph; P.X = h->d_X; P.U = h->d_U;
    P.x_init = h->d_xinit; P.goal_lo = h->d_glo; P.goal_hi = h->d_ghi; P.tf = h->d_tf;
    P.sub_Delta = h->d_subD; P.sub_omega = h->d_subW; P.sub_toggle = h->d_subT; P.sub_X = h->d_subX; P.sub_U = h->d_subU;
    P.sub_obj = h->d_subObj; P.sub_status = h->d_subSt; P.sub_iters = h->d_subIt;
    P.st_i = h->d_sti; P.st_d = h->d_std;
    P.J_true = h->d_Jt; P.J_full = h->d_Jf; P.conv = h->d_conv; P.Delta = h->d_Delta; P.omega = h->d_omega; P.rho = h->d_rho;
    P.accept = h->d_acc; P.scp_status = h->d_scp; P.solver_status = h->d_sol; P.tr_sat = h->d_tr; P.cvx_sat = h->d_cvx;
    P.ipm_it = h->d_ipm;
    P.tp = h->tp; P.to_mu = h->d_to_mu; P.to_xtol = h->d_to_xtol; P.to_ftol = h->d_to_ftol; P.to_ctol = h->d_to_ctol;
    P.wl = make_ws_layout<MODEL>(h->N, P.n_obs);
    P.ll = make_lds_layout<MODEL>(h->N, launch_waves(h) > (h->N + 63) / 64);
    if (!h->d_queue) HIPCHK(h, dalloc(&h->d_queue, (size_t)SQ_WORDS));
#ifdef GUSTO_PROFILE
    if (!h->d_prof) HIPCHK(h, dalloc(&h->d_prof, (size_t)h->batch_cap * PROF_N));
#endif
    P.prof = h->d_prof;
    return GUSTO_OK;
}

template <int MODEL> static int launch_scp(gusto_handle h, int mode, int max_iter, int force) {
    using T = MT<MODEL>;
    KParams P;
    int rc = fill_params<MODEL>(h, P, h->B);
    if (rc) return rc;
    P.mode = mode; P.max_iter = max_iter; P.force = force;
    const int NT = 64 * launch_waves(h);
    size_t lds = (size_t)P.ll.total * sizeof(double);
    if (const char* pad = dev_env("GUSTO_DEV_LDS_KB")) lds = std::max(lds, (size_t)atoi(pad) * 1024);  // occupancy experiments
    if (lds > 160 * 1024) { h->err = "problem does not fit the 160 KiB LDS of a CU"; return GUSTO_ERR_ARG; }
    // a problem with N <= 64 knots runs as one wave per workgroup (no barriers at all)
    auto kern = (NT == 64) ? &scp_kernel<MODEL, true> : &scp_kernel<MODEL, false>;
    int per_cu = 0, cus = 0;
    HIPCHK(h, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
    const bool masked = mode == 0 && h->n_active >= 0;    // gusto_set_active: only the listed problems are handed out
    if (masked) P.n_fresh = h->n_active;
    int NTL = NT;
    const bool want_chains = h->decomposition == GUSTO_DECOMP_WAVE2 || h->decomposition == GUSTO_DECOMP_WAVE4;
    bool have_chains = false;
    [[maybe_unused]] int nch_used = 0;
#if GUSTO_SEG_W2
    // a batch that leaves SIMDs without a wave runs several waves per problem, the KKT solve's sequential phases as Riccati segments
    // side by side and the obstacle rows shared (scp_kernel_w2, segw.hpp): four waves up to six problems per CU (one four-wave
    // workgroup is resident per CU: up to six rounds under the longest-first scheduler), two up to the batch size where the one-wave
    // kernel's three problems per CU win (measured, profiles/r06_wave_per_chain.txt: astrobeeSE3 4096 .. 8192, astrobeeSE3manifold
    // 2048 .. 4096)
    if constexpr (seg2_big<MODEL>()) {
        constexpr int TWO_UP_TO = (MODEL == GUSTO_ASTROBEE_SE3_MANIFOLD) ? 8 : 16;
        int nch = 0;
        if (NT == 64 && P.n_fresh <= 6 * cus && h->N >= 4 * GUSTO_SEG_MIN_N) nch = 4;
        else if (NT == 64 && P.n_fresh <= TWO_UP_TO * cus && h->N >= 2 * GUSTO_SEG_MIN_N) nch = 2;
        if (h->decomposition == GUSTO_DECOMP_WAVE) nch = 0;
        if (want_chains) { nch = (h->decomposition == GUSTO_DECOMP_WAVE4) ? 4 : 2; if (NT != 64 || h->N < nch * GUSTO_SEG_MIN_N) nch = 0; }
        if (const char* e = dev_env("GUSTO_DEV_W2")) { nch = atoi(e); if (nch == 1) nch = 2; if ((nch != 2 && nch != 4) || NT != 64 || h->N < nch * GUSTO_SEG_MIN_N) nch = 0; }
        if (nch) {
            P.ll = make_lds_layout<MODEL>(h->N, false, nch);
            lds = (size_t)P.ll.total * sizeof(double);
            kern = (nch == 4) ? &scp_kernel_w2<MODEL, 4> : &scp_kernel_w2<MODEL, 2>; NTL = 64 * nch;
            have_chains = true; nch_used = nch;
        }
    }
#endif
    if (want_chains && !have_chains) {
        h->err = "GUSTO_DECOMP_WAVE2 / WAVE4: no such kernel for this model, horizon or waves-per-problem setting";
        return GUSTO_ERR_ARG;
    }
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    // Persistent launch: as many workgroups as the GPU keeps resident (slots), each with its own workspace
    HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), NTL, lds));
    int slots = std::max(1, per_cu) * std::max(1, cus);
    if (const char* e = dev_env("GUSTO_DEV_SLOTS")) slots = std::max(1, atoi(e));   // occupancy experiments
    slots = std::min(slots, std::max(1, P.n_fresh));
    h->slots = slots; h->lds_bytes = (int)lds; h->per_cu = per_cu;
    {
        const size_t need = P.wl.total * (size_t)slots;
        if (need > h->ws_doubles) {
            if (h->d_ws) hipFree(h->d_ws);
            h->d_ws = nullptr; h->ws_doubles = 0;
            HIPCHK(h, dalloc(&h->d_ws, need));
            h->ws_doubles = need;
        }
        P.ws = h->d_ws;
    }
    // scheduler state of this launch (scp.hpp): counters to 0, waiting lists to -1
    // number of probing slices: the caller's (gusto_set_schedule) or the model's default (2; dubins_car 1)
    const int probe = h->sched_forced ? h->probe_iters : MT<MODEL>::SCHED_PROBE;
    // ... for batches of probe_min_batch problems and more (2048 unless the caller says otherwise) and for any batch that does not
    // fit the resident workgroups at once (1024 for freeflyerSE2, 768 one-wave / 512 two-wave / 256 four-wave workgroups of the
    // 12/13-state models; round 6, measured: freeflyerSE2 B = 1536 28.3 -> 26.4 ms, astrobeeSE3 B = 1024 two waves 25.4 -> 24.5 ms,
    // B = 512 four waves 19.1 -> 17.7 ms, astrobeeSE3manifold B = 768 four waves 44.6 -> 37.1 ms, B = 1536 one wave 96.9 -> 83.7 ms;
    // the manifold model's four-wave kernel from the third round on: 36.2 -> 36.9 ms at 512)
    int min_batch = h->probe_min_batch;
    if (!h->sched_forced)
        min_batch = std::min(min_batch, ((nch_used == 4 && MODEL == GUSTO_ASTROBEE_SE3_MANIFOLD) ? 2 * slots : slots) + 1);
    const bool dyn = mode == 0 && probe > 0 && probe < 128 && max_iter > probe && h->B >= min_batch && h->B < (1 << 24);
    memset(h->sched_init, 0, sizeof(h->sched_init));
    h->sched_init[SQ_PROBING] = dyn ? P.n_fresh : 0;     // every problem starts with its probing slices still ahead
    HIPCHK(h, hipMemcpyAsync(h->d_queue, h->sched_init, SQ_WORDS * sizeof(int), hipMemcpyHostToDevice, h->stream));
    int slice_q = dyn ? (dev_env("GUSTO_SLICE_Q") ? atoi(dev_env("GUSTO_SLICE_Q")) : MT<MODEL>::SCHED_SLICE) : 0;
    int pushes = probe + (slice_q > 0 ? (max_iter + slice_q - 1) / slice_q + 1 : 0);   // finite slices of a problem at most
    // a waiting-list entry keeps the slice count in its high byte, (slices + 1) << 24 | problem, and a negative entry means
    // "not published yet": more than 126 finite slices per problem do not fit -- then a problem of level 0 runs to its end
    // after its probing slices (slicing only moves time)
    if (pushes >= 127) { slice_q = 0; pushes = probe; }
    if (dyn) {
        const size_t need = (size_t)SCHED_LEVELS * pushes * h->batch_cap;
        if (need > h->order_ints) {
            if (h->d_order) hipFree(h->d_order);
            h->d_order = nullptr; h->order_ints = 0;
            HIPCHK(h, dalloc(&h->d_order, need));
            h->order_ints = need;
        }
        P.list_cap = pushes * h->B;
        HIPCHK(h, hipMemsetAsync(h->d_order, 0xFF, (size_t)SCHED_LEVELS * P.list_cap * sizeof(int), h->stream));
    }
    P.queue = h->d_queue; P.lists = h->d_order; P.probe_visits = dyn ? probe : 0; P.slice_q = slice_q;
    P.order = masked ? h->d_active + h->batch_cap : nullptr;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));   // (the solve's time includes the ordering kernels below)
    if (!masked && dyn && T::HAS_OBS && P.n_obs > 0 && !dev_env("GUSTO_DEV_NO_ORDER")) {   // hardest first (scp.hpp: sched_key_kernel)
        if (!h->d_sched_ord) HIPCHK(h, dalloc(&h->d_sched_ord, (size_t)2 * h->batch_cap));
        int* bucket = h->d_sched_ord;
        int* order = h->d_sched_ord + h->batch_cap;
        HIPCHK(h, hipMemsetAsync(bucket, 0, (size_t)h->B * sizeof(int), h->stream));
        const int tot = h->B * h->N;
        hipLaunchKernelGGL(sched_key_kernel<MODEL>, dim3((tot + 255) / 256), dim3(256), 0, h->stream, P, bucket);
        hipLaunchKernelGGL(sched_order_kernel, dim3(1), dim3(1024), 0, h->stream, h->B, bucket, order);
        HIPCHK(h, hipGetLastError());
        P.order = order;
    }
    if (dev_env("GUSTO_DEV_DEBUG"))
        fprintf(stderr, "launch: B %d slots %d dyn %d probe %d list_cap %d queue %p lists %p..%p ws %p..%p X %p st_i %p..%p hist Delta %p\n", h->B, slots,
                (int)dyn, P.probe_visits, P.list_cap, (void*)P.queue, (void*)P.lists, (void*)(P.lists + h->order_ints), (void*)P.ws,
                (void*)(P.ws + h->ws_doubles), (void*)P.X, (void*)P.st_i, (void*)(P.st_i + (size_t)h->batch_cap * ST_NI), (void*)P.Delta);
    hipLaunchKernelGGL(kern, dim3(slots), dim3(NTL), lds, h->stream, P);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    h->sched_err = 0;
    HIPCHK(h, gusto_fetch_sched_err(h));
    h->pending = true;   // completed by gusto_finish (handle.hpp)
    return GUSTO_OK;
}

#ifdef GUSTO_WITH_LANE
// Which decomposition a GuSTO solve of this handle runs (models that have the lane-per-problem kernel, lane.hpp): the
// caller's choice (gusto_set_decomposition), else GUSTO_DEV_LANE=0/1, else a wave per problem -- measured on MI355X
// (profiles/r04_lane_vs_wave.txt) the lane kernel is the slower one at every batch size of BASELINE.json: an interior point
// iteration of 64 problems issues ~170 k instructions (0.33 ms on an idle GPU) and a wave runs as long as the longest of
// its 64 problems (813 iterations + trips in the config-3 batch, against 80 on average), DESIGN.md section 3
static inline bool lane_decomposition(gusto_handle h) {
    if (h->decomposition == 1) return false;
    if (h->decomposition == 2) return true;
    if (const char* e = dev_env("GUSTO_DEV_LANE")) return atoi(e) != 0;   // (development builds)
    return false;
}
// One lane per problem (lane.hpp): ceil(B / lanes per wave) one-wave workgroups, each with its own block of the lane
// workspace; no device-side scheduler (every problem is resident from the start, a lane runs its problem to the end)
template <int MODEL> static int launch_lane(gusto_handle h, int mode, int max_iter, int force) {
    using Y = LaneLay<MODEL>;
    KParams P;
    int rc = fill_params<MODEL>(h, P, h->B);
    if (rc) return rc;
    if (h->n_active >= 0) { h->err = "gusto_set_active: not with a lane per problem (gusto_set_decomposition)"; return GUSTO_ERR_STATE; }
    P.mode = mode; P.max_iter = max_iter; P.force = force;
    int cus = 0;
    HIPCHK(h, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
    // lanes per wave: 64 unless the batch is too small to give every SIMD a wave (then fewer problems per wave: the
    // instruction stream of a wave costs the same for 16 problems as for 64)
    int lpw = 64;
    if (const char* e = dev_env("GUSTO_DEV_LANES_PER_WAVE")) lpw = std::max(1, std::min(64, atoi(e)));
    else while (lpw > 8 && (h->B + lpw - 1) / lpw < 4 * std::max(1, cus)) lpw >>= 1;
    // persistent lanes: at most the waves the GPU keeps resident (one per SIMD at this kernel's register budget); a lane that
    // finishes takes the next problem of the batch (lane.hpp)
    int per_cu = 0;
    HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(&lane_kernel<MODEL>), 64, 0));
    int nw = std::min((h->B + lpw - 1) / lpw, std::max(1, per_cu) * std::max(1, cus));
    if (const char* e = dev_env("GUSTO_DEV_SLOTS")) nw = std::max(1, std::min(nw, atoi(e)));
    const size_t need = (size_t)nw * (size_t)h->N * (size_t)(Y::EK * 64);
    if (need > h->ws_doubles) {
        if (h->d_ws) hipFree(h->d_ws);
        h->d_ws = nullptr; h->ws_doubles = 0;
        HIPCHK(h, dalloc(&h->d_ws, need));
        h->ws_doubles = need;
    }
    P.ws = h->d_ws;
    h->slots = nw; h->lds_bytes = 0; h->per_cu = per_cu;
    memset(h->sched_init, 0, sizeof(h->sched_init));
    h->sched_init[SQ_HEAD_A] = std::min(h->B, nw * lpw);   // the problems handed out at launch: the counter's start
    HIPCHK(h, hipMemcpyAsync(h->d_queue, h->sched_init, SQ_WORDS * sizeof(int), hipMemcpyHostToDevice, h->stream));
    P.queue = h->d_queue;
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(lane_kernel<MODEL>, dim3(nw), dim3(64), 0, h->stream, P, lpw);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    h->sched_err = 0;
    HIPCHK(h, gusto_fetch_sched_err(h));
    h->pending = true;
    return GUSTO_OK;
}
#endif   // GUSTO_WITH_LANE

// TrajOpt: every problem of the batch through trajopt_kernel (scp.hpp); mode 1 = one subproblem per problem (parity hook)
template <int MODEL> static int launch_trajopt(gusto_handle h, int mode, int max_iter) {
    KParams P;
    int rc = fill_params<MODEL>(h, P, h->B);
    if (rc) return rc;
    P.mode = mode; P.max_iter = max_iter; P.force = 0;
    const int NT = 64 * ((h->N + 63) / 64);
    const size_t lds = (size_t)P.ll.total * sizeof(double);
    if (lds > 160 * 1024) { h->err = "problem does not fit the 160 KiB LDS of a CU"; return GUSTO_ERR_ARG; }
    auto kern = &trajopt_kernel<MODEL>;
    HIPCHK(h, hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    int per_cu = 0, cus = 0;
    HIPCHK(h, hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void*>(kern), NT, lds));
    HIPCHK(h, hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, h->device));
    const int slots = std::min(std::max(1, per_cu) * std::max(1, cus), h->B);
    h->slots = slots; h->lds_bytes = (int)lds; h->per_cu = per_cu;
    const size_t need = P.wl.total * (size_t)slots;
    if (need > h->ws_doubles) {
        if (h->d_ws) hipFree(h->d_ws);
        h->d_ws = nullptr; h->ws_doubles = 0;
        HIPCHK(h, dalloc(&h->d_ws, need));
        h->ws_doubles = need;
    }
    P.ws = h->d_ws;
    if (h->d_queue) HIPCHK(h, hipMemsetAsync(h->d_queue, 0, SQ_WORDS * sizeof(int), h->stream));
    HIPCHK(h, hipEventRecord(h->ev0, h->stream));
    hipLaunchKernelGGL(kern, dim3(slots), dim3(NT), lds, h->stream, P);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipEventRecord(h->ev1, h->stream));
    h->sched_err = 0;
    HIPCHK(h, gusto_fetch_sched_err(h));
    h->pending = true;
    return GUSTO_OK;
}

template <int MODEL> static int launch_init(gusto_handle h, bool straight) {
    KParams P;
    int rc = fill_params<MODEL>(h, P, h->B, false);   // (the init kernels do not look at the keep-out sets)
    if (rc) return rc;
    if (straight) {
        const int tot = h->B * h->N;
        hipLaunchKernelGGL(init_straightline_kernel<MODEL>, dim3((tot + 255) / 256), dim3(256), 0, h->stream, P);
        HIPCHK(h, hipGetLastError());
    }
    hipLaunchKernelGGL(reset_state_kernel<MODEL>, dim3((h->B + 255) / 256), dim3(256), 0, h->stream, P);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return GUSTO_OK;
}

