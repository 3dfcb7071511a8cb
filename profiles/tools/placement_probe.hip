// Where do the wavefronts of a 256-thread workgroup land?  One record per wavefront: XCC, SE, CU, SIMD, wave slot (HW_ID / XCC_ID
// hardware registers, gfx950), for a grid shaped like the step kernel's (B / E workgroups of 4 wavefronts, ~16 KB of LDS each, all
// co-resident for a few microseconds).  Answers two questions the step kernel's role layout depends on:
//   - does wavefront k of EVERY workgroup sit on the same SIMD (then the one wavefront that runs the agent phases shares its SIMD
//     with the agent wavefronts of the 7 other workgroups on the CU, while three SIMDs idle)?
//   - which blockIdx values share a CU (what a rotation of the roles has to be a function of)?
//   hipcc --offload-arch=gfx950 -O2 -o placement_probe profiles/tools/placement_probe.hip && ./placement_probe [n_workgroups] [lds_bytes]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <vector>
#include <tuple>

__global__ void __launch_bounds__(256) probe(uint32_t *out, int spin_ticks) {
    extern __shared__ int smem[];
    const int wave = threadIdx.x >> 6;
    const uint32_t hw = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_REG_HW_ID
    const uint32_t xcc = __builtin_amdgcn_s_getreg((31 << 11) | 20);  // HW_REG_XCC_ID
    smem[threadIdx.x] = (int)hw;
    const uint64_t t0 = wall_clock64();
    while ((int64_t)(wall_clock64() - t0) < spin_ticks) { }
    __syncthreads();
    if ((threadIdx.x & 63) == 0) {
        out[(blockIdx.x * 4 + wave) * 2 + 0] = hw;
        out[(blockIdx.x * 4 + wave) * 2 + 1] = xcc;
    }
}

int main(int argc, char **argv) {
    const int n = argc > 1 ? atoi(argv[1]) : 2048, lds = argc > 2 ? atoi(argv[2]) : 16 * 1024;
    uint32_t *d;
    hipMalloc(&d, (size_t)n * 4 * 2 * 4);
    hipMemset(d, 0, (size_t)n * 4 * 2 * 4);
    for (int rep = 0; rep < 2; ++rep) {
        hipLaunchKernelGGL(probe, dim3(n), dim3(256), lds, 0, d, 600);  // 600 ticks of the 100 MHz clock = 6 us
        hipDeviceSynchronize();
    }
    std::vector<uint32_t> h((size_t)n * 8);
    hipMemcpy(h.data(), d, h.size() * 4, hipMemcpyDeviceToHost);
    int simd_hist[4][4] = {};  // [wave][simd]
    std::map<std::tuple<int, int, int>, std::vector<int>> per_cu;
    for (int b = 0; b < n; ++b)
        for (int w = 0; w < 4; ++w) {
            const uint32_t hw = h[(b * 4 + w) * 2], xcc = h[(b * 4 + w) * 2 + 1] & 0xf;
            const int simd = (hw >> 4) & 3, cu = (hw >> 8) & 0xf, se = (hw >> 13) & 7;
            simd_hist[w][simd]++;
            if (w == 0) per_cu[{(int)xcc, se, cu}].push_back(b | (simd << 24));
        }
    printf("%d workgroups x 4 wavefronts, %d B of LDS each; CUs seen: %zu\n", n, lds, per_cu.size());
    printf("SIMD of wavefront k (rows) -> count per SIMD 0..3\n");
    for (int w = 0; w < 4; ++w) printf("  wave %d: %6d %6d %6d %6d\n", w, simd_hist[w][0], simd_hist[w][1], simd_hist[w][2], simd_hist[w][3]);
    int shown = 0;
    for (auto &kv : per_cu) {
        if (shown++ >= 12) break;
        printf("  xcc %d se %d cu %2d: ", std::get<0>(kv.first), std::get<1>(kv.first), std::get<2>(kv.first));
        for (int v : kv.second) printf(" %d(s%d)", v & 0xffffff, v >> 24);
        printf("\n");
    }
    // how many distinct wave-0 SIMDs per CU on average
    double acc = 0;
    for (auto &kv : per_cu) { int m = 0; for (int v : kv.second) m |= 1 << (v >> 24); acc += __builtin_popcount(m); }
    printf("distinct SIMDs hosting wavefront 0 per CU: %.2f (4 = spread, 1 = all on one SIMD)\n", acc / per_cu.size());
    return 0;
}
