#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
// D(16x16) = A(16x4) B(4x16) + C; lane l: A[l&15][l>>4], B[l>>4][l&15]; D reg r: row (l>>4)+4r, col l&15
__global__ void k(const double* A, const double* B, double* D, int K) {
    const int l = threadIdx.x, i = l & 15, q = l >> 4;
    v4d acc = {0, 0, 0, 0};
    for (int s = 0; s < K / 4; s++) {
        const double a = A[i * K + q + 4 * s], b = B[(q + 4 * s) * 16 + i];
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
    }
    for (int r = 0; r < 4; r++) D[(q + 4 * r) * 16 + i] = acc[r];
}
int main() {
    const int K = 12;
    double hA[16 * K], hB[K * 16], hD[256], ref[256];
    for (int i = 0; i < 16; i++) for (int k = 0; k < K; k++) hA[i * K + k] = 0.1 * i - 0.37 * k + (i == k);
    for (int k = 0; k < K; k++) for (int j = 0; j < 16; j++) hB[k * 16 + j] = 1.0 + 0.01 * k * k - 0.3 * j;
    for (int i = 0; i < 16; i++) for (int j = 0; j < 16; j++) { double s = 0; for (int k = 0; k < K; k++) s += hA[i * K + k] * hB[k * 16 + j]; ref[i * 16 + j] = s; }
    double *dA, *dB, *dD;
    hipMalloc(&dA, sizeof(hA)); hipMalloc(&dB, sizeof(hB)); hipMalloc(&dD, sizeof(hD));
    hipMemcpy(dA, hA, sizeof(hA), hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof(hB), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, K);
    hipMemcpy(hD, dD, sizeof(hD), hipMemcpyDeviceToHost);
    double e = 0; for (int i = 0; i < 256; i++) e = fmax(e, fabs(hD[i] - ref[i]));
    printf("max err %g\n", e);
    return e < 1e-12 ? 0 : 1;
}
