#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
#define TICK(t) __builtin_amdgcn_sched_barrier(0); t = clock64(); __builtin_amdgcn_sched_barrier(0)
// one wave: (0) 32 independent f64 MFMAs, (1) 32 dependent (same accumulator), (2) 512 independent FMAs (4 chains),
// (3) 32 MFMAs each followed by 16 FMAs (sum of (0) and (2) if they do not overlap, max if they do), (4) MFMA + 8 FMAs
__global__ void k(long long* out, double* sink, double x0) {
    const int t = threadIdx.x;
    double a = x0 + t, b = x0 * 1.1, c = x0 * 1.2, d = x0 * 1.3, m = 1.0000001, q = 1e-9;
    v4d A0 = {0,0,0,0}, A1 = {0,0,0,0}, A2 = {0,0,0,0}, A3 = {0,0,0,0};
    long long t0, t1;
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    TICK(t0);
#pragma unroll
    for (int i = 0; i < 8; i++) {
        A0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, A0, 0, 0, 0); A1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, c, A1, 0, 0, 0);
        A2 = __builtin_amdgcn_mfma_f64_16x16x4f64(b, c, A2, 0, 0, 0); A3 = __builtin_amdgcn_mfma_f64_16x16x4f64(c, d, A3, 0, 0, 0);
    }
    asm volatile("" : "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3));
    TICK(t1);
    if (t == 0) out[0] = t1 - t0;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < 32; i++) A0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, A0, 0, 0, 0);
    asm volatile("" : "+v"(A0));
    TICK(t1);
    if (t == 0) out[1] = t1 - t0;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < 128; i++) { a = __builtin_fma(a, m, q); b = __builtin_fma(b, m, q); c = __builtin_fma(c, m, q); d = __builtin_fma(d, m, q); }
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d));
    TICK(t1);
    if (t == 0) out[2] = t1 - t0;
    double e = x0 * 0.7, f = x0 * 0.6;
    asm volatile("" : "+v"(e), "+v"(f));
    TICK(t0);
#pragma unroll
    for (int i = 0; i < 32; i++) {
        if ((i & 3) == 0) A0 = __builtin_amdgcn_mfma_f64_16x16x4f64(e, f, A0, 0, 0, 0);
        if ((i & 3) == 1) A1 = __builtin_amdgcn_mfma_f64_16x16x4f64(e, f, A1, 0, 0, 0);
        if ((i & 3) == 2) A2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e, f, A2, 0, 0, 0);
        if ((i & 3) == 3) A3 = __builtin_amdgcn_mfma_f64_16x16x4f64(e, f, A3, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; j++) { a = __builtin_fma(a, m, q); b = __builtin_fma(b, m, q); c = __builtin_fma(c, m, q); d = __builtin_fma(d, m, q); }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3));
    TICK(t1);
    if (t == 0) out[3] = t1 - t0;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < 32; i++) {
        if ((i & 3) == 0) A0 = __builtin_amdgcn_mfma_f64_16x16x4f64(e, f, A0, 0, 0, 0);
        if ((i & 3) == 1) A1 = __builtin_amdgcn_mfma_f64_16x16x4f64(e, f, A1, 0, 0, 0);
        if ((i & 3) == 2) A2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e, f, A2, 0, 0, 0);
        if ((i & 3) == 3) A3 = __builtin_amdgcn_mfma_f64_16x16x4f64(e, f, A3, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 2; j++) { a = __builtin_fma(a, m, q); b = __builtin_fma(b, m, q); c = __builtin_fma(c, m, q); d = __builtin_fma(d, m, q); }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3));
    TICK(t1);
    if (t == 0) out[4] = t1 - t0;
    // (5) 32 MFMAs each followed by 16 32-bit integer VALU ops
    int ia = t, ib = t * 3, ic = t * 5, id = t * 7;
    asm volatile("" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id));
    TICK(t0);
#pragma unroll
    for (int i = 0; i < 32; i++) {
        if ((i & 3) == 0) A0 = __builtin_amdgcn_mfma_f64_16x16x4f64(e, f, A0, 0, 0, 0);
        if ((i & 3) == 1) A1 = __builtin_amdgcn_mfma_f64_16x16x4f64(e, f, A1, 0, 0, 0);
        if ((i & 3) == 2) A2 = __builtin_amdgcn_mfma_f64_16x16x4f64(e, f, A2, 0, 0, 0);
        if ((i & 3) == 3) A3 = __builtin_amdgcn_mfma_f64_16x16x4f64(e, f, A3, 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int j = 0; j < 4; j++) { ia = ia * 3 + ib; ib = ib * 5 + ic; ic = ic * 7 + id; id = id * 9 + ia; }
        __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id), "+v"(A0), "+v"(A1), "+v"(A2), "+v"(A3));
    TICK(t1);
    if (t == 0) out[5] = t1 - t0;
    TICK(t0);
#pragma unroll
    for (int i = 0; i < 32; i++) {
#pragma unroll
        for (int j = 0; j < 4; j++) { ia = ia * 3 + ib; ib = ib * 5 + ic; ic = ic * 7 + id; id = id * 9 + ia; }
    }
    asm volatile("" : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id));
    TICK(t1);
    if (t == 0) out[6] = t1 - t0;
    sink[t] = a + b + c + d + A0[0] + A1[1] + A2[2] + A3[3] + ia + ib + ic + id;
}
int main() {
    long long* o; double* s;
    (void)hipMalloc(&o, 64 * 8); (void)hipMalloc(&s, 64 * 8);
    for (int r = 0; r < 2; r++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, s, 1.0);
    long long h[8]; (void)hipMemcpy(h, o, sizeof h, hipMemcpyDeviceToHost);
    const char* nm[] = {"32 independent mfma_f64_16x16x4", "32 dependent mfma (same acc)", "512 fma (4 chains)", "32 x (mfma + 16 fma)", "32 x (mfma + 8 fma)", "32 x (mfma + 16 x (v_mul_lo+add))", "32 x 16 x (mul_lo+add)"};
    for (int i = 0; i < 7; i++) printf("%-40s %lld cycles\n", nm[i], h[i]);
    return 0;
}
