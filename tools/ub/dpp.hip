#include <hip/hip_runtime.h>
#include <cstdio>
// one wave: cycles per instruction of the fp64 DPP forms gfx950 has (v_fmac_f64_dpp row_newbcast, v_mov_b64_dpp row_shr / row_newbcast)
// against plain fma and the v_readlane pair they replace
#define T0() asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); __builtin_amdgcn_sched_barrier(0); t0 = clock64(); __builtin_amdgcn_sched_barrier(0)
#define T1(slot) asm volatile("" : "+v"(a), "+v"(b), "+v"(c), "+v"(d)); __builtin_amdgcn_sched_barrier(0); t1 = clock64(); __builtin_amdgcn_sched_barrier(0); if (t == 0) out[slot] = t1 - t0
__global__ void k(long long* out, double* sink, double x0) {
    const int t = threadIdx.x;
    double a = x0 + t * 1e-3, b = x0 * 1.1, c = x0 * 1.2, d = x0 * 1.3, m = 1.0000001, q = 1e-9;
    long long t0, t1;
    T0();   // 0: 256 dependent plain fma
#pragma unroll
    for (int i = 0; i < 256; i++) a = __builtin_fma(a, m, q);
    T1(0);
    T0();   // 1: 256 dependent fmac_dpp (acc chain, source b constant): acc += bcast(b, 3) * m
#pragma unroll
    for (int i = 0; i < 256; i++) asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(b), "v"(m));
    T1(1);
    T0();   // 2: 256 fmac_dpp, 4 independent accumulators
#pragma unroll
    for (int i = 0; i < 64; i++) {
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(a) : "v"(q), "v"(m));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(b) : "v"(q), "v"(m));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(c) : "v"(q), "v"(m));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(d) : "v"(q), "v"(m));
    }
    T1(2);
    T0();   // 3: 128 steps of the recurrence pattern: the DPP source is the previous result (write -> s_nop 1 -> dpp read)
#pragma unroll
    for (int i = 0; i < 128; i++) {
        double acc = q;
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(m));
        a = acc;
    }
    T1(3);
    T0();   // 4: 128 x (v_mov_b64_dpp row_shr:3 then fma with it), dependent
#pragma unroll
    for (int i = 0; i < 128; i++) {
        unsigned long long u = __builtin_bit_cast(unsigned long long, a);
        unsigned lo = (unsigned)u, hi = (unsigned)(u >> 32), lo2, hi2;
        asm volatile("s_nop 1\n\tv_mov_b32_dpp %0, %2 row_shr:3 row_mask:0xf bank_mask:0xf\n\tv_mov_b32_dpp %1, %3 row_shr:3 row_mask:0xf bank_mask:0xf" : "=&v"(lo2), "=&v"(hi2) : "v"(lo), "v"(hi));
        a = __builtin_fma(__builtin_bit_cast(double, ((unsigned long long)hi2 << 32) | lo2), m, q);
    }
    T1(4);
    T0();   // 5: 128 x (two v_readlane + fma), dependent: what the sweeps do today
#pragma unroll
    for (int i = 0; i < 128; i++) {
        unsigned long long u = __builtin_bit_cast(unsigned long long, a);
        unsigned lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffu), 3), hi = __builtin_amdgcn_readlane((int)(u >> 32), 3);
        a = __builtin_fma(b, __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo), q);
    }
    T1(5);
    T0();   // 6: one recurrence step of n = 6 as the sweep would issue it, 64 times: 6 fmac_dpp on one accumulator from the previous vector
#pragma unroll
    for (int i = 0; i < 64; i++) {
        double acc = q;
        asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:0 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(m));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:1 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:2 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(c));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(d));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:4 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(m));
        asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(a), "v"(b));
        a = acc * 1e-3;
    }
    T1(6);
    T0();   // 7: the same step with 12 readlanes + 6 fma
#pragma unroll
    for (int i = 0; i < 64; i++) {
        double acc = q;
#pragma unroll
        for (int l = 0; l < 6; l++) {
            unsigned long long u = __builtin_bit_cast(unsigned long long, a);
            unsigned lo = __builtin_amdgcn_readlane((int)(u & 0xffffffffu), l), hi = __builtin_amdgcn_readlane((int)(u >> 32), l);
            acc = __builtin_fma(l & 1 ? b : m, __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo), acc);
        }
        a = acc * 1e-3;
    }
    T1(7);
    sink[t] = a + b + c + d;
    // correctness of the broadcast: every lane of a row gets lane L of ITS row
    double v = (double)t, r = 0.0;
    asm volatile("s_nop 1\n\tv_fmac_f64_dpp %0, %1, %2 row_newbcast:5 row_mask:0xf bank_mask:0xf" : "+v"(r) : "v"(v), "v"(m));
    sink[64 + t] = r;
}
int main() {
    long long* out; double* sink;
    hipMalloc(&out, 16 * sizeof(long long)); hipMalloc(&sink, 256 * sizeof(double));
    for (int rep = 0; rep < 2; rep++) hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, out, sink, 1.0);
    hipDeviceSynchronize();
    long long h[16]; double s[128];
    hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost); hipMemcpy(s, sink, sizeof(s), hipMemcpyDeviceToHost);
    const char* nm[] = {"256 dep fma", "256 dep fmac_dpp (acc chain)", "256 fmac_dpp, 4 accumulators", "128 recurrence fmac_dpp (src = prev result)",
                        "128 x (2 mov_b32_dpp row_shr + fma)", "128 x (2 readlane + fma)", "64 steps n=6: 6 fmac_dpp", "64 steps n=6: 12 readlane + 6 fma"};
    const int cnt[] = {256, 256, 256, 128, 128, 128, 64, 64};
    for (int i = 0; i < 8; i++) printf("%-48s %6lld cycles  %.1f per item\n", nm[i], h[i], (double)h[i] / cnt[i]);
    printf("row_newbcast:5 of lane ids: lane 0 -> %.7g, lane 17 -> %.7g, lane 40 -> %.7g, lane 63 -> %.7g (expect 5 21 37 53 x 1.0000001)\n", s[64], s[64 + 17], s[64 + 40], s[64 + 63]);
    return 0;
}
