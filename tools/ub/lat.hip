#include <hip/hip_runtime.h>
#include <cstdio>
// one wave: cycles per instruction for dependent / independent f64 FMA chains, LDS reads, readlanes
__global__ void k(long long* out, double* sink, double x0) {
    __shared__ double lds[1024];
    const int t = threadIdx.x;
    for (int i = t; i < 1024; i += 64) lds[i] = i * 1e-3;
    __syncthreads();
    double a = x0, b = x0 * 1.1, c = x0 * 1.2, d = x0 * 1.3, m = 1.0000001, q = 1e-9;
    long long t0, t1;
    // dependent chain, 256 fma
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 256; i++) a = __builtin_fma(a, m, q);
    asm volatile("" : "+v"(a));
    __builtin_amdgcn_sched_barrier(0); t1 = clock64(); __builtin_amdgcn_sched_barrier(0);
    if (t == 0) out[0] = t1 - t0;
    // 4 independent chains, 256 fma total
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 64; i++) { a = __builtin_fma(a, m, q); b = __builtin_fma(b, m, q); c = __builtin_fma(c, m, q); d = __builtin_fma(d, m, q); }
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    __builtin_amdgcn_sched_barrier(0); t1 = clock64(); __builtin_amdgcn_sched_barrier(0);
    if (t == 0) out[1] = t1 - t0;
    // 2 independent chains
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 128; i++) { a = __builtin_fma(a, m, q); b = __builtin_fma(b, m, q); }
    asm volatile("" : "+v"(a), "+v"(b));
    __builtin_amdgcn_sched_barrier(0); t1 = clock64(); __builtin_amdgcn_sched_barrier(0);
    if (t == 0) out[2] = t1 - t0;
    // dependent rsq chain (64)
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 64; i++) a = __builtin_amdgcn_rsq(a + 2.0);
    asm volatile("" : "+v"(a));
    __builtin_amdgcn_sched_barrier(0); t1 = clock64(); __builtin_amdgcn_sched_barrier(0);
    if (t == 0) out[3] = t1 - t0;
    // LDS: 64 independent ds_read_b64 issued back to back then waited
    double acc = 0;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; i++) acc += lds[(t + 64 * i) & 1023];
    asm volatile("" : "+v"(acc));
    __builtin_amdgcn_sched_barrier(0); t1 = clock64(); __builtin_amdgcn_sched_barrier(0);
    if (t == 0) out[4] = t1 - t0;
    // LDS dependent round trips: write then read another lane's value, 16 times
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 16; i++) { lds[t] = acc; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); acc = lds[(t + 1) & 63] + 1.0; __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); }
    asm volatile("" : "+v"(acc));
    __builtin_amdgcn_sched_barrier(0); t1 = clock64(); __builtin_amdgcn_sched_barrier(0);
    if (t == 0) out[5] = t1 - t0;
    // readlane + use as scalar operand, 64 times dependent
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 64; i++) {
        unsigned long long u = __builtin_bit_cast(unsigned long long, a);
        unsigned lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffu), i & 63), hi = __builtin_amdgcn_readlane((int)(u >> 32), i & 63);
        a = __builtin_fma(b, __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo), a);
    }
    asm volatile("" : "+v"(a));
    __builtin_amdgcn_sched_barrier(0); t1 = clock64(); __builtin_amdgcn_sched_barrier(0);
    if (t == 0) out[6] = t1 - t0;
    // empty interval (tick overhead)
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0); t1 = clock64();
    if (t == 0) out[7] = t1 - t0;
    // mul+add mix dependent: 128 x (mul, add)
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int i = 0; i < 128; i++) { a = a * m; a = a + q; }
    asm volatile("" : "+v"(a));
    __builtin_amdgcn_sched_barrier(0); t1 = clock64(); __builtin_amdgcn_sched_barrier(0);
    if (t == 0) out[8] = t1 - t0;
    sink[t] = a + b + c + d + acc;
}
int main() {
    long long* o; double* s;
    (void)hipMalloc(&o, 64 * 8); (void)hipMalloc(&s, 64 * 8);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, s, 1.0);
    long long h[16]; (void)hipMemcpy(h, o, 16 * 8, hipMemcpyDeviceToHost);
    const char* nm[] = {"256 dependent fma", "256 fma in 4 chains", "256 fma in 2 chains", "64 dependent rsq(+add)", "16 indep ds_read_b64 + adds", "16 LDS write->read round trips", "64 x (2 readlane + fma)", "empty", "128 x dependent (mul, add)"};
    for (int i = 0; i < 9; i++) printf("%-34s %6lld ticks\n", nm[i], h[i]);
    return 0;
}
