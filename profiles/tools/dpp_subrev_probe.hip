// MEASUREMENT / DIAGNOSTIC TOOL (not product code): does `x_k - x_me - 1` with x_k fetched by a DPP quad_perm give the
// same answer on the GPU as on the host?  hipcc folds the DPP move into the subtract (v_subrev_u32_dpp); the agent-phase
// winner test written that way (round 2) passed the host emulation and FAILED a golden trace on the GPU.
//   hipcc --offload-arch=gfx950 -O3 -o dpp_subrev_probe profiles/tools/dpp_subrev_probe.hip && ./dpp_subrev_probe
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <vector>
#include "../../robotic-warehouse_amd/csrc/rware_cdna4.h"

__global__ void probe_folded(const uint32_t *in, uint32_t *folded) {   // every gathered value has ONE use: hipcc folds the DPP move
    const uint32_t vme = in[threadIdx.x];
    int kv[4];
    rw::env_gather<4>((int)vme, (int)(threadIdx.x & ~3u), kv);
    uint32_t beat = 0xffffffffu;
#pragma unroll
    for (int k = 0; k < 4; ++k) beat = min(beat, (uint32_t)kv[k] - vme - 1u);
    folded[threadIdx.x] = beat < 127u ? 1u : 0u;
}

__global__ void probe_diff(const uint32_t *in, uint32_t *diff0) {   // plain `lane0 - me`, one use as well
    const uint32_t vme = in[threadIdx.x];
    diff0[threadIdx.x] = (uint32_t)rw::quad_perm<0, 0, 0, 0>((int)vme) - vme;
}

__global__ void probe_unfolded(const uint32_t *in, uint32_t *unfolded) {
    const uint32_t vme = in[threadIdx.x];
    int kv[4];
    rw::env_gather<4>((int)vme, (int)(threadIdx.x & ~3u), kv);
    uint32_t lose = 0;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const uint32_t w = rw::opaque((uint32_t)kv[k]);   // the value in a register of its own: nothing to fold
        lose |= (((w ^ vme) < 256u) & (w > vme)) ? 1u : 0u;
    }
    unfolded[threadIdx.x] = lose;
}

int main() {
    std::vector<uint32_t> h(64), f(64), u(64), d(64);
    for (int i = 0; i < 64; ++i) {
        const int quad = i >> 2, a = i & 3;
        // quads 0..7: agents 1 and 3 contest one cell, 0 and 2 stand; quads 8..15: all four contest with depths a
        h[i] = quad < 8 ? ((a & 1) ? (194u << 8) | (15u - a) : 0x7fff0000u | ((uint32_t)a << 8))
                        : (77u << 8) | ((uint32_t)a << 4) | (15u - a);
    }
    uint32_t *din, *df, *du, *dd;
    hipMalloc(&din, 256); hipMalloc(&df, 256); hipMalloc(&du, 256); hipMalloc(&dd, 256);
    hipMemcpy(din, h.data(), 256, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe_folded, dim3(1), dim3(64), 0, 0, din, df);
    hipLaunchKernelGGL(probe_unfolded, dim3(1), dim3(64), 0, 0, din, du);
    hipLaunchKernelGGL(probe_diff, dim3(1), dim3(64), 0, 0, din, dd);
    hipMemcpy(f.data(), df, 256, hipMemcpyDeviceToHost);
    hipMemcpy(u.data(), du, 256, hipMemcpyDeviceToHost);
    hipMemcpy(d.data(), dd, 256, hipMemcpyDeviceToHost);
    int bad_f = 0, bad_u = 0, bad_d = 0;
    for (int i = 0; i < 64; ++i) {
        uint32_t want = 0;
        for (int k = 0; k < 4; ++k) {
            const uint32_t w = h[(i & ~3) + k];
            want |= (((w ^ h[i]) < 256u) && w > h[i]) ? 1u : 0u;
        }
        bad_f += f[i] != want;
        bad_u += u[i] != want;
        bad_d += d[i] != h[i & ~3] - h[i];
    }
    printf("lanes wrong: subtract with the DPP move folded in %d / 64, xor-compare on unfolded values %d / 64, plain lane0 - me %d / 64\n",
           bad_f, bad_u, bad_d);
    return 0;
}
