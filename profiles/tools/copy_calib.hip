// Calibration for rocprofv3 FETCH_SIZE / WRITE_SIZE on gfx950 (MI355X_MICROARCH.md §HBM says
// FETCH_SIZE reads 1/2 of a wide coalesced stream; WRITE_SIZE is uncalibrated): copies a known
// byte count with the same dwordx4 access width the step kernel uses, so the PMC numbers of the
// real kernel can be corrected by the ratio measured here.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void copy16(const float4* __restrict__ src, float4* __restrict__ dst, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
__global__ void copy4(const float* __restrict__ src, float* __restrict__ dst, size_t n) {  // 4 B / lane, like the record loads
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] = src[i];
}
int main() {
    const size_t bytes = 1ull << 30;  // 1 GiB each way, far past the 256 MiB Infinity Cache
    float4 *a, *b;
    hipMalloc(&a, bytes); hipMalloc(&b, bytes);
    hipMemset(a, 1, bytes); hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int it = 0; it < 5; ++it) {
        hipEventRecord(e0);
        copy16<<<2048, 256>>>(a, b, bytes / 16);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("copy16 %zu B read + %zu B written in %.3f ms = %.1f GB/s\n", bytes, bytes, ms, 2.0 * bytes / ms / 1e6);
    }
    for (int it = 0; it < 3; ++it) {
        hipEventRecord(e0);
        copy4<<<8192, 256>>>((const float*)a, (float*)b, bytes / 4);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("copy4  %zu B read + %zu B written in %.3f ms = %.1f GB/s\n", bytes, bytes, ms, 2.0 * bytes / ms / 1e6);
    }
    return 0;
}
