// FETCH_SIZE / WRITE_SIZE calibration on access patterns of known size: run under
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ...  and  --pmc WRITE_SIZE ...   (tools/fetch_calib.sh)
// Every kernel touches exactly BYTES bytes of a buffer much larger than the 256 MiB Infinity Cache, once.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
constexpr size_t BYTES = size_t(1) << 30;
// (a) 8 B per lane, consecutive lanes consecutive doubles: the access of the row-state arrays ([var][slot][k], lane = k)
__global__ void read8_coalesced(const double* a, double* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    double s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) s += a[i];
    if (s == 1.2345e300) out[0] = s;
}
// (b) 16 B per lane (global_load_dwordx4): the calibration pattern of the guide
__global__ void read16_coalesced(const double2* a, double* out, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    double s = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) { const double2 v = a[i]; s += v.x + v.y; }
    if (s == 1.2345e300) out[0] = s;
}
// (c) record walk: lane k reads entry e of ITS record (stride 192 doubles = 1536 B between lanes), e = 0 .. 143 in storage
// order -- how the stage-parallel phases of the 12/13-state kernels walk K | D | Phicl records; 144 of every 192 doubles are read
__global__ void read8_records(const double* a, double* out, size_t nrec) {
    size_t r = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    double s = 0;
    for (; r < nrec; r += (size_t)gridDim.x * blockDim.x) {
        const double* p = a + r * 192;
#pragma unroll 8
        for (int e = 0; e < 144; e++) s += p[e];
    }
    if (s == 1.2345e300) out[0] = s;
}
// (d) 8 B per lane coalesced stores
__global__ void write8_coalesced(double* a, size_t n) {
    size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) a[i] = (double)i;
}
// (e) record stores: 64 lanes store 64 consecutive doubles of a 192-double record, three times (the factor sweep's record stores)
__global__ void write8_records(double* a, size_t nrec) {
    const int lane = threadIdx.x & 63;
    size_t r = (blockIdx.x * (size_t)blockDim.x + threadIdx.x) >> 6;
    for (; r < nrec; r += ((size_t)gridDim.x * blockDim.x) >> 6) {
        double* p = a + r * 192;
        p[lane] = 1.0; p[64 + lane] = 2.0; if (lane < 16) p[128 + lane] = 3.0;   // 144 of 192
    }
}
int main() {
    double *a, *out;
    hipMalloc(&a, BYTES); hipMalloc(&out, 64);
    hipMemset(a, 0, BYTES);
    const size_t n = BYTES / 8, nrec = BYTES / (192 * 8);
    hipDeviceSynchronize();
    read8_coalesced<<<4096, 256>>>(a, out, n);
    read16_coalesced<<<4096, 256>>>((const double2*)a, out, n / 2);
    read8_records<<<4096, 64>>>(a, out, nrec);
    write8_coalesced<<<4096, 256>>>(a, n);
    write8_records<<<4096, 256>>>(a, nrec);
    hipDeviceSynchronize();
    printf("bytes read8_coalesced %zu read16_coalesced %zu read8_records(read) %zu (spanned %zu) write8_coalesced %zu write8_records(written) %zu\n",
           BYTES, BYTES, nrec * 144 * 8, nrec * 192 * 8, BYTES, nrec * 144 * 8);
    return 0;
}
