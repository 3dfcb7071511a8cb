// Multi-round companion of placement_probe.hip: more workgroups than fit at once (n workgroups of 4 wavefronts, `lds` bytes each), each
// spinning a pseudo-random 4..8 us, one raw record per WAVEFRONT: blockIdx wave xcc se cu simd wave_slot tg_id t_start t_end (100 MHz
// ticks).  Analysed offline (profiles/tools/placement_analyse.py): which wavefront-0s are alive on a SIMD at the same time.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

__global__ void __launch_bounds__(256) probe(uint64_t *out) {
    extern __shared__ int smem[];
    const int wave = threadIdx.x >> 6;
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    const uint64_t t0 = wall_clock64();
    const int spin = 400 + (int)((blockIdx.x * 2654435761u) >> 24) * 400 / 256;
    smem[threadIdx.x] = (int)hw;
    while ((int64_t)(wall_clock64() - t0) < spin) { }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        uint64_t *o = out + (size_t)(blockIdx.x * 4 + wave) * 4;
        o[0] = hw; o[1] = xcc; o[2] = t0; o[3] = wall_clock64();
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 8192, lds = argc > 2 ? atoi(argv[2]) : 17000;
    uint64_t *d;
    (void)hipMalloc(&d, (size_t)n * 4 * 4 * 8);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(n), dim3(256), lds, 0, d);
        (void)hipDeviceSynchronize();
    }
    std::vector<uint64_t> h((size_t)n * 16);
    (void)hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
    uint64_t tmin = ~0ull;
    for (int i = 0; i < n * 4; ++i) if (h[i * 4 + 2] < tmin) tmin = h[i * 4 + 2];
    printf("# n %d lds %d\n# blockIdx wave xcc se cu simd slot tg t_start t_end\n", n, lds);
    for (int b = 0; b < n; ++b)
        for (int w = 0; w < 4; ++w) {
            const uint64_t *o = &h[(size_t)(b * 4 + w) * 4];
            const uint32_t hw = (uint32_t)o[0];
            printf("%d %d %d %d %d %d %d %d %lld %lld\n", b, w, (int)(o[1] & 0xf), (hw >> 13) & 7, (hw >> 8) & 0xf, (hw >> 4) & 3, hw & 0xf, (hw >> 16) & 0xf,
                   (long long)(o[2] - tmin), (long long)(o[3] - tmin));
        }
    return 0;
}
