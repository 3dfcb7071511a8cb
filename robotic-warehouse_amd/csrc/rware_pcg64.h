// numpy-exact random draws for the RWARE engine (device + host).
//
// The reference draws from `numpy.random.Generator(PCG64)` at rware/warehouse.py:781 (agent
// cells), :788 (directions), :797 (request queue) and :916 (replacement request).  numpy is
// not vendored by the reference (setup.py:25, unpinned; 2.2.6 in the build image), so its
// published algorithms are restated here:
//   PCG64      128-bit LCG, multiplier 0x2360ED051FC65DA44385DF649FCCF645, XSL-RR 128/64
//              output of the post-step state; next_uint32 hands out the low half of a fresh
//              64-bit draw and buffers the high half (has_uint32 / uinteger).
//   bounded    random_bounded_uint64 -> buffered_bounded_lemire_uint32 (range < 2^32-1).
//   choice     Generator.choice(pop, size=k, replace=False): Floyd's sampling followed by a
//              Fisher-Yates pass (pop <= 10000).
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#define RW_HD __host__ __device__ __forceinline__

namespace rw {

typedef unsigned __int128 u128;

struct Pcg64 {
    u128 state, inc;
    uint32_t has_uint32, uinteger;
};

RW_HD u128 pcg_mult() { return ((((u128)0x2360ED051FC65DA4ULL) << 64) | 0x4385DF649FCCF645ULL); }

RW_HD uint64_t pcg_next64(Pcg64 &g) {
    g.state = g.state * pcg_mult() + g.inc;
    const uint64_t hi = (uint64_t)(g.state >> 64), lo = (uint64_t)g.state;
    const uint64_t x = hi ^ lo;
    const unsigned rot = (unsigned)(hi >> 58);
    return (x >> rot) | (x << ((0u - rot) & 63u));
}

RW_HD uint32_t pcg_next32(Pcg64 &g) {
    if (g.has_uint32) {
        g.has_uint32 = 0;
        return g.uinteger;
    }
    const uint64_t n = pcg_next64(g);
    g.has_uint32 = 1;
    g.uinteger = (uint32_t)(n >> 32);
    return (uint32_t)n;
}

// uniform integer in [0, rng] (inclusive); rng == 0 consumes nothing, as in numpy
RW_HD uint32_t pcg_bounded(Pcg64 &g, uint32_t rng) {
    if (rng == 0) return 0;
    const uint32_t rng_excl = rng + 1u;
    uint64_t m = (uint64_t)pcg_next32(g) * rng_excl;
    uint32_t leftover = (uint32_t)m;
    if (leftover < rng_excl) {
        const uint32_t threshold = (0xFFFFFFFFu - rng) % rng_excl;
        while (leftover < threshold) {
            m = (uint64_t)pcg_next32(g) * rng_excl;
            leftover = (uint32_t)m;
        }
    }
    return (uint32_t)(m >> 32);
}

// k distinct indices out of [0, pop) into out[0..k): Floyd + Fisher-Yates, numpy's draw order
template <typename IntPtr>
RW_HD void pcg_choice_no_replace(Pcg64 &g, int pop, int k, IntPtr out) {
    const int base = pop - k;
    for (int j = base; j < pop; ++j) {
        const int val = (int)pcg_bounded(g, (uint32_t)j);
        bool seen = false;
        for (int i = 0; i < j - base; ++i) seen |= (out[i] == val);
        out[j - base] = seen ? j : val;
    }
    for (int i = k - 1; i >= 1; --i) {
        const int j = (int)pcg_bounded(g, (uint32_t)i);
        const int t = out[j];
        out[j] = out[i];
        out[i] = t;
    }
}

}  // namespace rw
