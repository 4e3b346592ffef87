#include <hip/hip_runtime.h>
#include <cstdio>
// one wave: throughput of back-to-back LDS instructions (32 independent ops, one wait at the end)
#define REP4(x) x x x x
#define REP32(x) REP4(REP4(x)) REP4(REP4(x))
template <int MODE> __device__ long long run(double* lds, int t) {
    double v0, v1, v2, v3;
    unsigned a = (unsigned)(size_t)lds;   // LDS byte address
    unsigned addr = a + 8 * t;            // conflict free
    if (MODE == 1) addr = a;              // broadcast
    if (MODE == 4) addr = a + 8 * (t & 15) * 32;   // 16-way bank conflict
    if (MODE == 7) addr = a + 16 * t;     // b128
    long long t0, t1;
    __builtin_amdgcn_sched_barrier(0);
    t0 = clock64();
    __builtin_amdgcn_sched_barrier(0);
    if (MODE == 0 || MODE == 1 || MODE == 4) {
        REP32(asm volatile("ds_read_b64 %0, %1" : "=v"(v0) : "v"(addr) : "memory");)
    } else if (MODE == 2) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 w;
        REP32(asm volatile("ds_read2_b64 %0, %1 offset0:0 offset1:64" : "=v"(w) : "v"(addr) : "memory");)
        v0 = w.x;
    } else if (MODE == 3) {
        if (t < 16) { REP32(asm volatile("ds_read_b64 %0, %1" : "=v"(v0) : "v"(addr) : "memory");) }
    } else if (MODE == 5) {
        v0 = t;
        REP32(asm volatile("ds_write_b64 %1, %0" : : "v"(v0), "v"(addr) : "memory");)
    } else if (MODE == 6) {
        if (t < 6) { REP32(asm volatile("ds_read_b64 %0, %1" : "=v"(v0) : "v"(addr) : "memory");) }
    } else if (MODE == 7) {
        typedef double d2 __attribute__((ext_vector_type(2)));
        d2 w;
        REP32(asm volatile("ds_read_b128 %0, %1" : "=v"(w) : "v"(addr) : "memory");)
        v0 = w.x;
    } else if (MODE == 8) {
        float f0;
        REP32(asm volatile("ds_read_b32 %0, %1" : "=v"(f0) : "v"(addr) : "memory");)
        v0 = f0;
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    t1 = clock64();
    __builtin_amdgcn_sched_barrier(0);
    asm volatile("" :: "v"(v0));
    return t1 - t0;
}
__global__ void k(long long* out) {
    __shared__ double lds[4096];
    const int t = threadIdx.x;
    for (int i = t; i < 4096; i += 64) lds[i] = i;
    __syncthreads();
    long long r[9];
    r[0] = run<0>(lds, t); r[1] = run<1>(lds, t); r[2] = run<2>(lds, t); r[3] = run<3>(lds, t); r[4] = run<4>(lds, t);
    r[5] = run<5>(lds, t); r[6] = run<6>(lds, t); r[7] = run<7>(lds, t); r[8] = run<8>(lds, t);
    if (t == 0) for (int i = 0; i < 9; i++) out[i] = r[i];
}
int main() {
    long long* o; (void)hipMalloc(&o, 64 * 8);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o);
    long long h[16]; (void)hipMemcpy(h, o, 16 * 8, hipMemcpyDeviceToHost);
    const char* nm[] = {"32 ds_read_b64 conflict-free", "32 ds_read_b64 broadcast", "32 ds_read2_b64", "32 ds_read_b64, 16 lanes active", "32 ds_read_b64 16-way conflict",
                        "32 ds_write_b64", "32 ds_read_b64, 6 lanes active", "32 ds_read_b128", "32 ds_read_b32"};
    for (int i = 0; i < 9; i++) printf("%-36s %6lld ticks  (%.1f per op)\n", nm[i], h[i], (h[i] - 40) / 32.0);
    return 0;
}
