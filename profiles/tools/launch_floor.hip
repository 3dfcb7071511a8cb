// Measurement aid: what does one dependent kernel launch cost on this device, independent of our kernel?
//   empty        1024 x 256 threads, no work                         -> dispatch + completion floor
//   store_only   each workgroup writes its 18176-byte observation chunk (float4 stores) and nothing else
//   sleep_store  each workgroup idles ~5.5 us (s_sleep) and then writes the chunk: the shape of the step kernel
// Build: hipcc --offload-arch=gfx950 -O3 -o launch_floor launch_floor.hip ; run: ./launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void empty_kernel() {}
__global__ void store_only(float4 *out, int n4_per_wg) {
    float4 *o = out + (size_t)blockIdx.x * n4_per_wg;
    for (int i = threadIdx.x; i < n4_per_wg; i += blockDim.x) o[i] = float4{1.f, 0.f, 0.f, 1.f};
}
__global__ void sleep_store(float4 *out, int n4_per_wg, int sleeps) {
    for (int k = 0; k < sleeps; ++k) __builtin_amdgcn_s_sleep(16);
    float4 *o = out + (size_t)blockIdx.x * n4_per_wg;
    for (int i = threadIdx.x; i < n4_per_wg; i += blockDim.x) o[i] = float4{1.f, 0.f, 0.f, 1.f};
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

int main() {
    const int n_wg = 1024, n4 = 1136, iters = 1000;
    float4 *buf; CK(hipMalloc(&buf, (size_t)n_wg * n4 * sizeof(float4)));
    hipStream_t s; CK(hipStreamCreate(&s));
    hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    auto timeit = [&](const char *name, auto launch) {
        for (int i = 0; i < 200; ++i) launch();
        hipStreamSynchronize(s);
        hipEventRecord(a, s);
        for (int i = 0; i < iters; ++i) launch();
        hipEventRecord(b, s);
        hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b);
        printf("%-28s %7.2f us per launch\n", name, ms * 1000.f / iters);
    };
    timeit("empty 1024x256", [&] { hipLaunchKernelGGL(empty_kernel, dim3(n_wg), dim3(256), 0, s); });
    timeit("empty 1x64", [&] { hipLaunchKernelGGL(empty_kernel, dim3(1), dim3(64), 0, s); });
    for (int frac : {1, 2, 4, 8, 16})
        { char nm[64]; snprintf(nm, 64, "store_only %.1f MB", 18.6 / frac);
          timeit(nm, [&] { hipLaunchKernelGGL(store_only, dim3(n_wg), dim3(256), 0, s, buf, n4 / frac); }); }
    timeit("store_only 18.6MB 4096 wg", [&] { hipLaunchKernelGGL(store_only, dim3(4 * n_wg), dim3(256), 0, s, buf, n4 / 4); });
    timeit("store_only 18.6MB 256 wg", [&] { hipLaunchKernelGGL(store_only, dim3(n_wg / 4), dim3(256), 0, s, buf, n4 * 4); });
    timeit("store_only 18.6MB 512 wg x512", [&] { hipLaunchKernelGGL(store_only, dim3(n_wg / 2), dim3(512), 0, s, buf, n4 * 2); });
    {   // the observation batch of rware-large-16ag, sensor_range 2, B = 16384: 2048 workgroups x 93 696 B = 192 MB
        float4 *big; CK(hipMalloc(&big, (size_t)2048 * 5856 * sizeof(float4)));
        timeit("store_only 192 MB (2048 wg)", [&] { hipLaunchKernelGGL(store_only, dim3(2048), dim3(256), 0, s, big, 5856); });
        timeit("store_only 96 MB (2048 wg)", [&] { hipLaunchKernelGGL(store_only, dim3(2048), dim3(256), 0, s, big, 2928); });
        timeit("store_only 74 MB = 4x small-4ag", [&] { hipLaunchKernelGGL(store_only, dim3(4096), dim3(256), 0, s, big, 1136); });
        hipFree(big);
        // beyond the 256 MiB Infinity Cache: B = 262144 of small-4ag (16384 workgroups x 18 176 B = 298 MB)
        float4 *huge; CK(hipMalloc(&huge, (size_t)16384 * 1136 * sizeof(float4)));
        timeit("store_only 298 MB (16384 wg)", [&] { hipLaunchKernelGGL(store_only, dim3(16384), dim3(256), 0, s, huge, 1136); });
        hipFree(huge);
    }
    for (int sl : {0, 3, 6, 9, 12, 15})
        { char nm[64]; snprintf(nm, 64, "sleep(%d x1024clk)+store", sl);
          timeit(nm, [&] { hipLaunchKernelGGL(sleep_store, dim3(n_wg), dim3(256), 12000, s, buf, n4, sl); }); }
    return 0;
}
