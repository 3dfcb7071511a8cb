// rware_selftest.hip — rw_selftest(): the two toolchain / hardware facts the step kernels are written around, checked ON THE DEVICE by
// code that ships inside the library (no hipcc needed where it runs): called by __graft_entry__.smoke() and by a `-m gpu` test.
//
//   1. Cross-lane exchange (rware_kernels.h P2b).  hipcc folds a DPP move into the instruction that uses it; folded into a
//      NON-commutative use (`v_subrev_u32_dpp`) the operands came out swapped on gfx950 (round 2: a winner test written as
//      `x_k - x_me - 1` passed the host emulation and failed a golden trace on the GPU).  The kernels therefore compare gathered
//      values that sit in registers of their own (xor / compare).  Checked: env_gather<4> / env_gather<2> / env_gather<6> /
//      env_or<8> deliver what a host loop computes, and the winner test in the kernels' form is exact.
//   2. LDS-DMA across a barrier (rware_cdna4.h dma_wait).  A run-time compiled build once left its stage-in DMA in flight across
//      `__syncthreads()` (hipRTC's header does not make the compiler wait for vmcnt there).  The kernels wait explicitly.  Checked:
//      wavefronts 1..3 DMA a chunk into LDS, `dma_wait(); lds_barrier()`, wavefront 0 reads it back — under load, many workgroups,
//      several rounds with fresh data — and finds exactly the source bytes.
//
// RWARE_SELFTEST_BREAK=1 (tests): check 1 runs the folded subtract instead and check 2 skips the wait — the self-test has to FAIL
// then (a guard that cannot fail guards nothing; on a toolchain where the folded form happens to be right, check 2 still trips).
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <string>
#include <vector>

#include "../../include/rware_hip.h"
#include <rware_cdna4.h>
#include "rware_hooks.h"

namespace rw {

// what the agent phases do: every lane of a quad announces (cell << 8 | depth << 4 | 15 - index); lane loses iff another lane of its
// quad announces the same cell with a larger priority
template <bool kBroken>
__global__ void selftest_exchange_kernel(const uint32_t *in, uint32_t *out) {
    const int tid = threadIdx.x, lane = tid & 63;
    const uint32_t vme = in[tid];
    int kv[4];
    env_gather<4>((int)vme, lane & ~3, kv);
    uint32_t lose = 0;
    if (kBroken) {  // the form the kernels avoid: one use per gathered value, a subtract — the DPP move is folded into it
        uint32_t beat = 0xffffffffu;
#pragma unroll
        for (int k = 0; k < 4; ++k) beat = min(beat, (uint32_t)kv[k] - vme - 1u);
        lose = beat < 127u ? 1u : 0u;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) lose |= ((((uint32_t)kv[k] ^ vme) < 256u) & ((uint32_t)kv[k] > vme)) ? 1u : 0u;
    }
    int k2[2], k6[6];
    env_gather<2>((int)vme, lane & ~1, k2);
    const int g6 = lane / 6, base6 = (g6 < 10 ? g6 : 9) * 6;  // (the 4 idle tail lanes gather from the last group, as in the kernels)
    env_gather<6>((int)vme, base6, k6);
    uint32_t x6 = 0;
#pragma unroll
    for (int k = 0; k < 6; ++k) x6 ^= (uint32_t)k6[k] * (uint32_t)(k + 1);
    const uint32_t o8 = (uint32_t)env_or<8>((int)(1u << (lane & 7)) | (int)(vme & 0xff00u), lane & ~7);
    out[4 * tid + 0] = lose;
    out[4 * tid + 1] = (uint32_t)k2[0] * 3u + (uint32_t)k2[1];
    out[4 * tid + 2] = x6;
    out[4 * tid + 3] = o8;
}

// wavefronts 1..3 DMA `pieces` 16-byte pieces of this workgroup's source chunk into LDS; wavefront 0 reads them back
template <bool kBroken>
__global__ void selftest_dma_kernel(const uint32_t *src, uint32_t *dst, int pieces, int rounds) {
    extern __shared__ __align__(16) int32_t smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int r = 0; r < rounds; ++r) {
        const RW_GLOBAL char *s = as_bytes(as_global(src + ((size_t)blockIdx.x * rounds + r) * pieces * 4));
        if (wave > 0)
            for (int c = (wave - 1) * 64; c < pieces; c += 3 * 64)
                if (c + lane < pieces) lds_dma_b128(s + (size_t)(c + lane) * 16, smem + 4 * c);
        if (!kBroken) dma_wait();
        lds_barrier();
        if (wave == 0)
            for (int i = lane; i < pieces * 4; i += 64) dst[((size_t)blockIdx.x * rounds + r) * pieces * 4 + i] = (uint32_t)smem[i];
        lds_barrier();  // (the next round's DMA overwrites what wavefront 0 has just read)
    }
}

}  // namespace rw

extern "C" int rw_selftest(int32_t device_id, char *log, size_t log_len) {
    std::string msg;
    int failed = 0;
    int caller_dev = -1;  // the caller's current HIP device is put back on every way out
    if (hipGetDevice(&caller_dev) != hipSuccess) { caller_dev = -1; (void)hipGetLastError(); }
    auto done = [&](int rc) {
        if (log && log_len) snprintf(log, log_len, "%s", msg.c_str());
        if (caller_dev >= 0) (void)hipSetDevice(caller_dev);
        return rc;
    };
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || device_id < 0 || device_id >= n_dev) {
        msg = "no such HIP device";
        return done(RW_ERR_NO_DEVICE);
    }
    if (hipSetDevice(device_id) != hipSuccess) { msg = "hipSetDevice failed"; return done(RW_ERR_HIP); }
    const char *br = rw_hook("RWARE_SELFTEST_BREAK");
    const bool broken = br && br[0] == '1';
    char line[256];

    // ---- 1. cross-lane exchange
    {
        const int T = 256;
        std::vector<uint32_t> h(T), got(4 * T);
        for (int i = 0; i < T; ++i) {
            const int quad = i >> 2, a = i & 3;
            // even quads: agents 1 and 3 contest one cell, 0 and 2 stand still; odd quads: all four contest with depths a
            h[i] = (quad & 1) == 0 ? ((a & 1) ? ((uint32_t)(190 + quad) << 8) | (15u - a) : 0x7fff0000u | ((uint32_t)a << 8))
                                   : ((uint32_t)(70 + quad) << 8) | ((uint32_t)a << 4) | (15u - a);
        }
        uint32_t *din = nullptr, *dout = nullptr;
        bool ok = hipMalloc(&din, T * 4) == hipSuccess && hipMalloc(&dout, 4 * T * 4) == hipSuccess &&
                  hipMemcpy(din, h.data(), T * 4, hipMemcpyHostToDevice) == hipSuccess;
        if (ok) {
            if (broken) hipLaunchKernelGGL(rw::selftest_exchange_kernel<true>, dim3(1), dim3(T), 0, 0, (const uint32_t *)din, dout);
            else hipLaunchKernelGGL(rw::selftest_exchange_kernel<false>, dim3(1), dim3(T), 0, 0, (const uint32_t *)din, dout);
            ok = hipGetLastError() == hipSuccess && hipMemcpy(got.data(), dout, 4 * T * 4, hipMemcpyDeviceToHost) == hipSuccess;
        }
        if (din) (void)hipFree(din);
        if (dout) (void)hipFree(dout);
        if (!ok) { msg = "exchange check: HIP error"; return done(RW_ERR_HIP); }
        int bad[4] = {0, 0, 0, 0};
        for (int i = 0; i < T; ++i) {
            const int lane = i & 63, w0 = i & ~63;
            uint32_t lose = 0;
            for (int k = 0; k < 4; ++k) {
                const uint32_t w = h[(i & ~3) + k];
                lose |= (((w ^ h[i]) < 256u) && w > h[i]) ? 1u : 0u;
            }
            const uint32_t p2 = h[i & ~1] * 3u + h[(i & ~1) + 1];
            const int g6 = lane / 6, base6 = (g6 < 10 ? g6 : 9) * 6;
            uint32_t x6 = 0;
            for (int k = 0; k < 6; ++k) x6 ^= h[w0 + base6 + k] * (uint32_t)(k + 1);
            uint32_t o8 = 0;
            for (int k = 0; k < 8; ++k) o8 |= (1u << k) | (h[(i & ~7) + k] & 0xff00u);
            bad[0] += got[4 * i] != lose; bad[1] += got[4 * i + 1] != p2; bad[2] += got[4 * i + 2] != x6; bad[3] += got[4 * i + 3] != o8;
        }
        snprintf(line, sizeof line, "exchange: wrong lanes — winner test %d, pair gather %d, 6-lane gather %d, 8-lane OR %d of %d%s; ",
                 bad[0], bad[1], bad[2], bad[3], T, broken ? " (RWARE_SELFTEST_BREAK: folded subtract)" : "");
        msg += line;
        if (bad[0] | bad[1] | bad[2] | bad[3]) failed |= 1;
    }
    // ---- 2. LDS-DMA across the barrier
    {
        const int WG = 2048, pieces = 768 /* 12 KiB per round */, rounds = 6;
        const size_t n = (size_t)WG * rounds * pieces * 4;
        std::vector<uint32_t> h(n), got(n);
        uint32_t x = 0x9e3779b9u;
        for (size_t i = 0; i < n; ++i) { x = x * 1664525u + 1013904223u; h[i] = x; }
        uint32_t *dsrc = nullptr, *ddst = nullptr;
        bool ok = hipMalloc(&dsrc, n * 4) == hipSuccess && hipMalloc(&ddst, n * 4) == hipSuccess &&
                  hipMemcpy(dsrc, h.data(), n * 4, hipMemcpyHostToDevice) == hipSuccess && hipMemsetAsync(ddst, 0, n * 4, nullptr) == hipSuccess;
        if (ok) {
            if (broken) hipLaunchKernelGGL(rw::selftest_dma_kernel<true>, dim3(WG), dim3(256), pieces * 16, 0, (const uint32_t *)dsrc, ddst, pieces, rounds);
            else hipLaunchKernelGGL(rw::selftest_dma_kernel<false>, dim3(WG), dim3(256), pieces * 16, 0, (const uint32_t *)dsrc, ddst, pieces, rounds);
            ok = hipGetLastError() == hipSuccess && hipMemcpy(got.data(), ddst, n * 4, hipMemcpyDeviceToHost) == hipSuccess;
        }
        if (dsrc) (void)hipFree(dsrc);
        if (ddst) (void)hipFree(ddst);
        if (!ok) { msg += "DMA check: HIP error"; return done(RW_ERR_HIP); }
        size_t bad = 0;
        for (size_t i = 0; i < n; ++i) bad += got[i] != h[i];
        snprintf(line, sizeof line, "LDS-DMA across the barrier: %zu of %zu dwords wrong (%d workgroups x %d rounds x %d KiB)%s", bad, n, WG, rounds,
                 pieces * 16 / 1024, broken ? " (RWARE_SELFTEST_BREAK: no wait)" : "");
        msg += line;
        if (bad) failed |= 2;
    }
    return done(failed ? RW_ERR_SELFTEST : RW_OK);
}
