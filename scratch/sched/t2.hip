#include <hip/hip_runtime.h>
#include <cstdio>
#include <unistd.h>
#include <chrono>
#define GD __device__ __forceinline__
struct P_ { int B; int* q; int* count; };
// stage 1: plain atomicAdd queue
__global__ void __launch_bounds__(64, 1) k1(const P_ P) {
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = atomicAdd(P.q, 1);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b >= P.B) return;
        if (threadIdx.x == 0) atomicAdd(P.count + b, 1);
    }
}
// stage 2: relaxed agent atomic load before the atomicAdd
__global__ void __launch_bounds__(64, 1) k2(const P_ P) {
    for (;;) {
        int b = -1;
        if (threadIdx.x == 0) {
            if (__hip_atomic_load(P.q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < P.B) { int q = atomicAdd(P.q, 1); if (q < P.B) b = q; }
        }
        b = __builtin_amdgcn_readfirstlane(b);
        if (b < 0) return;
        if (threadIdx.x == 0) atomicAdd(P.count + b, 1);
    }
}
// stage 3: the pop as a function with an inner spin loop and s_sleep
GD int pop3(const P_& P) {
    for (int spin = 0;; spin++) {
        const int probing = __hip_atomic_load(P.q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_load(P.q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < P.B) { int q = atomicAdd(P.q, 1); if (q < P.B) return q; }
        if (probing == 0 || spin > 1000) return -1;
        __builtin_amdgcn_s_sleep(8);
    }
}
__global__ void __launch_bounds__(64, 1) k3(const P_ P) {
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = pop3(P);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b < 0) return;
        if (threadIdx.x == 0) atomicAdd(P.count + b, 1);
    }
}
// stage 4: + the level scan with the CAS loop (lists stay empty)
GD int pop4(const P_& P) {
    for (int spin = 0;; spin++) {
        const int probing = __hip_atomic_load(P.q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (__hip_atomic_load(P.q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < P.B) { int q = atomicAdd(P.q, 1); if (q < P.B) return q; }
        for (int L = 15; L >= 0; L--) {
            int h = __hip_atomic_load(P.q + 18 + L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (h < __hip_atomic_load(P.q + 2 + L, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) {
                if (__hip_atomic_compare_exchange_strong(P.q + 18 + L, &h, h + 1, __ATOMIC_RELAXED, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) return 1 << 30;
            }
        }
        if (probing == 0 || spin > 1000) return -1;
        __builtin_amdgcn_s_sleep(8);
    }
}
__global__ void __launch_bounds__(64, 1) k4(const P_ P) {
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = pop4(P);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b < 0) return;
        if (threadIdx.x == 0) atomicAdd(P.count + b, 1);
    }
}
// stage 5: k3 + fake work with s_sleep in a loop, all lanes
__global__ void __launch_bounds__(64, 1) k5(const P_ P) {
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = pop3(P);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b < 0) return;
        for (int i = 0; i < 2000 * ((b % 5) + 1); i++) __builtin_amdgcn_s_sleep(1);
        if (threadIdx.x == 0) atomicAdd(P.count + b, 1);
    }
}
// stage 6: k3 + release fence / waitcnt / acquire fence
__global__ void __launch_bounds__(64, 1) k6(const P_ P) {
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = pop3(P);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b < 0) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (threadIdx.x == 0) {
            atomicAdd(P.count + b, 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(P.q + 40, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}
template <class K> void run(const char* name, K kern, int B) {
    P_ P{}; P.B = B; hipMalloc(&P.q, 64 * 4); hipMalloc(&P.count, B * 4);
    hipMemset(P.q, 0, 64 * 4); hipMemset(P.count, 0, B * 4);
    hipEvent_t ev; hipEventCreate(&ev);
    hipLaunchKernelGGL(kern, dim3(B < 1024 ? B : 1024), dim3(64), 0, 0, P);
    hipEventRecord(ev, 0);
    auto t0 = std::chrono::steady_clock::now();
    while (hipEventQuery(ev) != hipSuccess) {
        if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) { printf("%s B=%d: HANG\n", name, B); fflush(stdout); _exit(1); }
        usleep(1000);
    }
    printf("%s B=%d: ok\n", name, B); fflush(stdout);
}


__global__ void __launch_bounds__(64, 1) k6a(const P_ P) {
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = pop3(P);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b < 0) return;
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (threadIdx.x == 0) {
            atomicAdd(P.count + b, 1);
            
            __hip_atomic_store(P.q + 40, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void __launch_bounds__(64, 1) k6b(const P_ P) {
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = pop3(P);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b < 0) return;
        
        if (threadIdx.x == 0) {
            atomicAdd(P.count + b, 1);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            __hip_atomic_store(P.q + 40, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void __launch_bounds__(64, 1) k6c(const P_ P) {
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = pop3(P);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b < 0) return;
        
        if (threadIdx.x == 0) {
            atomicAdd(P.count + b, 1);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __hip_atomic_store(P.q + 40, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void __launch_bounds__(64, 1) k6d(const P_ P) {
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = pop3(P);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b < 0) return;
        
        if (threadIdx.x == 0) {
            atomicAdd(P.count + b, 1);
            
            __hip_atomic_store(P.q + 40, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void __launch_bounds__(64, 1) k7b(const P_ P) {
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = pop3(P);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b < 0) return;
        __threadfence();
        if (threadIdx.x == 0) {
            
            atomicAdd(P.count + b, 1);
            __hip_atomic_store(P.q + 40, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void __launch_bounds__(64, 1) k7c(const P_ P) {
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = pop3(P);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b < 0) return;
        asm volatile("buffer_inv sc1" ::: "memory");
        if (threadIdx.x == 0) {
            
            atomicAdd(P.count + b, 1);
            __hip_atomic_store(P.q + 40, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void __launch_bounds__(64, 1) k7e(const P_ P) {
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = pop3(P);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b < 0) return;
        
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            atomicAdd(P.count + b, 1);
            __hip_atomic_store(P.q + 40, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void __launch_bounds__(64, 1) k7f(const P_ P) {
    for (;;) {
        int b = 0;
        if (threadIdx.x == 0) b = pop3(P);
        b = __builtin_amdgcn_readfirstlane(b);
        if (b < 0) return;
        P.count[P.B + threadIdx.x] = b; __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (threadIdx.x == 0) {
            
            atomicAdd(P.count + b, 1);
            __hip_atomic_store(P.q + 40, b, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}

__global__ void k7g(const P_ P) {   // no loop at all
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (threadIdx.x == 0) atomicAdd(P.count + blockIdx.x, 1);
}
#include <cstring>
template <class K> void run2(const char* name, K kern, int B) {
    P_ P{}; P.B = B; hipMalloc(&P.q, 64 * 4); hipMalloc(&P.count, (B + 64) * 4);
    hipMemset(P.q, 0, 64 * 4); hipMemset(P.count, 0, (B + 64) * 4);
    auto t0 = std::chrono::steady_clock::now();
    hipLaunchKernelGGL(kern, dim3(B < 1024 ? B : 1024), dim3(64), 0, 0, P);
    hipError_t e = hipDeviceSynchronize();
    printf("%s B=%d: %s %.3f ms\n", name, B, hipGetErrorString(e), 1e3 * std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count()); fflush(stdout);
}
int main(int argc, char** argv) {
    const char* w = argc > 1 ? argv[1] : "k3";
    if (!strcmp(w, "k3")) { run2("k3", k3, 64); run2("k3", k3, 4096); }
    if (!strcmp(w, "k4")) { run2("k4", k4, 64); run2("k4", k4, 4096); }
    if (!strcmp(w, "k6")) { run2("k6", k6, 64); run2("k6", k6, 4096); }
    if (!strcmp(w, "k6a")) { run2("k6a", k6a, 64); run2("k6a", k6a, 4096); }
    if (!strcmp(w, "k6b")) { run2("k6b", k6b, 64); run2("k6b", k6b, 4096); }
    if (!strcmp(w, "k6c")) { run2("k6c", k6c, 64); run2("k6c", k6c, 4096); }
    if (!strcmp(w, "k6d")) { run2("k6d", k6d, 64); run2("k6d", k6d, 4096); }
    if (!strcmp(w, "k7b")) { run2("k7b", k7b, 64); run2("k7b", k7b, 4096); }
    if (!strcmp(w, "k7c")) { run2("k7c", k7c, 64); run2("k7c", k7c, 4096); }
    if (!strcmp(w, "k7e")) { run2("k7e", k7e, 64); run2("k7e", k7e, 4096); }
    if (!strcmp(w, "k7f")) { run2("k7f", k7f, 64); run2("k7f", k7f, 4096); }
    if (!strcmp(w, "k7g")) { run2("k7g", k7g, 64); run2("k7g", k7g, 1024); }
    return 0;
}
