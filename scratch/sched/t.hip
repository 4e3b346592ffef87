#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <unistd.h>
#define GD __device__ __forceinline__
namespace gusto {
constexpr int SCHED_LEVELS = 16, SQ_HEAD_A = 0, SQ_PROBING = 1, SQ_TAIL = 2, SQ_HEAD = 2 + SCHED_LEVELS, SQ_WORDS = 2 + 2 * SCHED_LEVELS;
struct KParams { int B, probe_visits, mode, list_cap; int* queue; int* lists; int* count; };
constexpr int SCHED_SPIN_LIMIT = 1 << 18;
// wave-uniform primitives: every lane of the wave executes them, the result is the same scalar in every lane
GD int uload(const int* p) { return __builtin_amdgcn_readfirstlane(__hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)); }
GD int uadd(int* p, int v) {
    int r = 0;
    if ((threadIdx.x & 63) == 0) r = atomicAdd(p, v);
    return __builtin_amdgcn_readfirstlane(r);
}
GD int ucas(int* p, int expected, int desired) {   // returns the value found (== expected: the swap happened)
    int r = expected;
    if ((threadIdx.x & 63) == 0) {
        __hip_atomic_compare_exchange_strong(p, &r, desired, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    return __builtin_amdgcn_readfirstlane(r);
}
// executed by a WHOLE wave with uniform control flow (a loop with exits inside `if (lane == 0)` does not survive the
// structurizer: the wave never reconverges)
GD int sched_pop(const KParams& P, bool& cont) {
    int* Q = P.queue;
    for (int spin = 0;; spin++) {
        const int probing_seen = uload(Q + SQ_PROBING);
        if (uload(Q + SQ_HEAD_A) < P.B) {
            const int q = uadd(Q + SQ_HEAD_A, 1);
            if (q < P.B) { cont = false; return q; }
        }
        for (int L = SCHED_LEVELS - 1; L >= 0; L--) {
            int h = uload(Q + SQ_HEAD + L);
            while (h < uload(Q + SQ_TAIL + L)) {
                const int found = ucas(Q + SQ_HEAD + L, h, h + 1);
                if (found == h) {
                    int e = -1;
                    for (int w = 0; w < SCHED_SPIN_LIMIT && (e = uload(P.lists + (size_t)L * P.list_cap + h)) < 0; w++) __builtin_amdgcn_s_sleep(2);
                    if (e < 0) return -1;
                    cont = true;
                    return e;
                }
                h = found;
            }
        }
        if (probing_seen == 0 || spin > SCHED_SPIN_LIMIT) return -1;
        if (spin < 16) __builtin_amdgcn_s_sleep(8); else __builtin_amdgcn_s_sleep(64);
    }
}
__global__ void __launch_bounds__(64, 1) k(const KParams P) {
    for (;;) {
        bool cont = false;
        int b = sched_pop(P, cont);
        if (b < 0) return;
        const int visits = cont ? (b >> 24) : 0;
        b &= (1 << 24) - 1;
        if (cont && threadIdx.x == 0) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        const int trips = (visits < P.probe_visits) ? 1 : (1 << 30);
        // fake slice: some work, problem b needs (b % 7) + 1 slices in total when sliced
        { volatile double acc = 1.0; for (int i = 0; i < 3000 * ((b % 5) + 1); i++) acc = acc * 1.0000001 + 1e-9; }
        if (threadIdx.x == 0) atomicAdd(P.count + b, 1);
        int lvl = -1;
        if (trips == 1 && (b % 3) != 0) lvl = b % 4;      // comes back
        if (threadIdx.x == 0) {
            if (lvl >= 0) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const int idx = atomicAdd(P.queue + SQ_TAIL + lvl, 1);
                __hip_atomic_store(P.lists + (size_t)lvl * P.list_cap + idx, ((visits + 1) << 24) | b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (trips == 1 && (lvl < 0 || visits + 1 >= P.probe_visits)) atomicSub(P.queue + SQ_PROBING, 1);
        }
    }
}
}
int main() {
    using namespace gusto;
    for (int probe : {0, 2}) for (int B : {64, 4096}) {
        KParams P{}; P.B = B; P.probe_visits = probe; P.list_cap = (probe ? probe : 1) * B;
        hipMalloc(&P.queue, SQ_WORDS * 4); hipMalloc(&P.lists, (size_t)SCHED_LEVELS * P.list_cap * 4); hipMalloc(&P.count, B * 4);
        int init[SQ_WORDS] = {0}; init[SQ_PROBING] = probe ? B : 0;
        hipMemcpy(P.queue, init, sizeof(init), hipMemcpyHostToDevice);
        hipMemset(P.lists, 0xFF, (size_t)SCHED_LEVELS * P.list_cap * 4); hipMemset(P.count, 0, B * 4);
        hipLaunchKernelGGL(k, dim3(B < 1024 ? B : 1024), dim3(64), 0, 0, P);
        hipEvent_t ev; hipEventCreate(&ev); hipEventRecord(ev, 0);
        { int waited = 0; while (hipEventQuery(ev) != hipSuccess) { usleep(1000); if (++waited > 8000) { printf("probe %d B %d: HANG\n", probe, B); fflush(stdout); _exit(1); } } }
        hipError_t e = hipDeviceSynchronize();
        std::vector<int> c(B); hipMemcpy(c.data(), P.count, B * 4, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int b = 0; b < B; b++) { int want = probe == 0 ? 1 : ((b % 3) != 0 ? 3 : 1); if (c[b] != want) bad++; }
        int q[SQ_WORDS]; hipMemcpy(q, P.queue, sizeof(q), hipMemcpyDeviceToHost);
        printf("probe %d B %d: %s bad %d headA %d probing %d\n", probe, B, hipGetErrorString(e), bad, q[0], q[1]); fflush(stdout);
    }
    return 0;
}
