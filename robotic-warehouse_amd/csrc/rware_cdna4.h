// gfx950 (CDNA4) primitives used by the step kernel.
//
//   lds_dma_b128 / lds_dma_b32   global -> LDS DMA (global_load_lds_dwordx4 / _dword): no VGPR
//                                round trip; LDS destination = wave-uniform base + lane * size,
//                                global source per lane; asynchronous until `s_waitcnt vmcnt`.
//   lds_barrier                  workgroup barrier that orders LDS traffic only: it waits
//                                lgkmcnt(0), NOT vmcnt, so HBM stores already in flight keep
//                                draining while the workgroup moves on (a plain __syncthreads()
//                                would stall every wave until they are acknowledged).
//   wave_sync                    orders the LDS traffic of ONE wavefront (64 lanes run in
//                                lockstep, so cross-lane exchange through LDS needs no s_barrier).
#pragma once
#ifndef __HIPCC_RTC__
#include <stdint.h>
#endif

#define RW_GLOBAL __attribute__((address_space(1)))  // (see as_global below)

namespace rw {

__device__ __forceinline__ void lds_dma_b128(const RW_GLOBAL void *g_lane, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(g_lane,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void lds_dma_b32(const RW_GLOBAL void *g_lane, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds(g_lane,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 4, 0, 0);
}
// dma_wait(): every LDS-DMA (and every other vector-memory load) this wavefront has issued is complete.  The stage-in barrier
// used to rely on __syncthreads() for this; hipRTC's runtime header defines __syncthreads() with a fence that does NOT make the
// compiler wait for vmcnt there, so a run-time compiled build left the DMA in flight across the barrier (round 4: wrong
// `requested` bits under load — the queue segment is the last DMA issued).  The wait is explicit now, whatever the header says.
__device__ __forceinline__ void dma_wait() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_wait() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
// RW_GLOBAL / as_global(p): a pointer that comes out of the parameter block (i.e. was loaded from memory) is a "generic"
// pointer to hipcc, addressed with flat_* instructions: those tick BOTH memory counters (every later `s_waitcnt lgkmcnt(0)`
// for an LDS read then also waits for them), cannot take the scalar-base addressing form, and take the long way through the
// address-aperture check.  as_global says what such a pointer is — device memory — so the accesses become global_*.
template <typename T>
__device__ __forceinline__ RW_GLOBAL T *as_global(T *p) { return (RW_GLOBAL T *)p; }
template <typename T>
__device__ __forceinline__ const RW_GLOBAL char *as_bytes(const RW_GLOBAL T *p) { return (const RW_GLOBAL char *)p; }
// HW_ID (hwreg 4): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]; XCC_ID (hwreg 20): xcc[3:0]
__device__ __forceinline__ void nap() { __builtin_amdgcn_s_sleep(2); }  // ~128 cycles off the issue slots
// wave_priority<P>(): s_setprio P — user priority of this wavefront for the SIMD's issue arbiter, 0 (the launch default) .. 3
template <int P> __device__ __forceinline__ void wave_priority() { __builtin_amdgcn_s_setprio(P); }
__device__ __forceinline__ uint32_t hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 4); }
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20); }
// wave_any(v): true in every lane iff v holds in some active lane of the wavefront (one s_cmp on the ballot)
__device__ __forceinline__ bool wave_any(bool v) { return __builtin_amdgcn_ballot_w64(v) != 0; }
// keep_sgpr(a, b, ...): the (wave-uniform) values must sit in scalar registers at this point — pins where
// their loads have to have been issued and waited for
__device__ __forceinline__ void keep_sgpr1(int v) { asm volatile("" ::"s"(v)); }
template <typename... Ts>
__device__ __forceinline__ void keep_sgpr(Ts... v) {
    (keep_sgpr1((int)v), ...);
}
// keep_vgpr(a, b): the values must have been computed at this point (profiling marks: pins work in front of a time stamp)
__device__ __forceinline__ void keep_vgpr(int a, int b) { asm volatile("" ::"v"(a), "v"(b)); }
// keep_sgpr_ptr(p, q, ...): the same for (wave-uniform) pointers
template <typename T>
__device__ __forceinline__ void keep_sgpr_ptr1(T p) { asm volatile("" ::"s"(p)); }
template <typename... Ts>
__device__ __forceinline__ void keep_sgpr_ptr(Ts... p) {
    (keep_sgpr_ptr1(p), ...);
}
// store_f4_nt(dst, v): a 16-byte store with the non-temporal hint (global_store_dwordx4 ... nt)
__device__ __forceinline__ void store_f4_nt(float4 *dst, float4 v) {
    typedef float f4_t __attribute__((ext_vector_type(4)));
    f4_t vv = {v.x, v.y, v.z, v.w};
    __builtin_nontemporal_store(vv, reinterpret_cast<f4_t *>(dst));
}
// opaque(x): the value, with everything the optimiser knew about its bits forgotten
__device__ __forceinline__ uint32_t opaque(uint32_t x) { asm("" : "+v"(x)); return x; }
// uniform(x): tells the compiler a value is wave-uniform (v_readfirstlane), so tests on it become scalar branches
__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
// wave_lds_order(): LDS writes issued before it land before LDS writes issued after it, for all lanes of the wavefront.
// The DS unit executes one wavefront's instructions in issue order, so on the GPU this only has to stop the compiler
// from reordering or merging the stores around it; no s_waitcnt, nothing is emitted.
__device__ __forceinline__ void wave_lds_order() {
    asm volatile("" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}
// env_gather<N>(v, lane_base, out): out[k] = v held by agent k of this lane's env, where the N agents of an env sit in N
// consecutive lanes starting at `lane_base` (a multiple of N).  All 64 lanes must be active (call it outside divergent
// control flow).  N = 4 / 2: DPP quad_perm moves, no LDS hardware involved; other N: ds_bpermute_b32, one per agent,
// issued back to back (one wait for the batch).
template <int SEL0, int SEL1, int SEL2, int SEL3>
__device__ __forceinline__ int quad_perm(int v) {
    // (bound_ctrl set: the "old" operand is dead, so no register has to be zeroed in front of the move)
    return __builtin_amdgcn_mov_dpp(v, SEL0 | (SEL1 << 2) | (SEL2 << 4) | (SEL3 << 6), 0xf, 0xf, true);
}
template <int N>
__device__ __forceinline__ void env_gather(int v, int lane_base, int (&out)[N]) {
    if constexpr (N == 1) {
        out[0] = v;
    } else if constexpr (N == 2) {
        out[0] = quad_perm<0, 0, 2, 2>(v);
        out[1] = quad_perm<1, 1, 3, 3>(v);
    } else if constexpr (N == 4) {
        out[0] = quad_perm<0, 0, 0, 0>(v);
        out[1] = quad_perm<1, 1, 1, 1>(v);
        out[2] = quad_perm<2, 2, 2, 2>(v);
        out[3] = quad_perm<3, 3, 3, 3>(v);
    } else {
#pragma unroll
        for (int k = 0; k < N; ++k) out[k] = __builtin_amdgcn_ds_bpermute((lane_base + k) << 2, v);
    }
}
// env_or<N>(v, lane_base): bitwise OR of v over the N agents of this lane's env, delivered to each of them.
template <int N>
__device__ __forceinline__ int env_or(int v, int lane_base) {
    if constexpr (N == 1) {
        return v;
    } else if constexpr (N == 2) {
        return v | quad_perm<1, 0, 3, 2>(v);
    } else if constexpr (N == 4) {
        v |= quad_perm<1, 0, 3, 2>(v);
        return v | quad_perm<2, 3, 0, 1>(v);
    } else if constexpr (N == 8) {  // an env is half a DPP row: the two quads meet through row_half_mirror (lane i <-> 7 - i)
        v |= quad_perm<1, 0, 3, 2>(v);
        v |= quad_perm<2, 3, 0, 1>(v);
        return v | __builtin_amdgcn_mov_dpp(v, 0x141, 0xf, 0xf, true);
    } else {
        int g[N], r = 0;
        env_gather<N>(v, lane_base, g);
#pragma unroll
        for (int k = 0; k < N; ++k) r |= g[k];
        return r;
    }
}
// env_any<N>(v, lane_base): true in every lane of an env iff v holds in one of its N lanes (one ballot, one 64-bit shift)
template <int N>
__device__ __forceinline__ bool env_any(bool v, int lane_base) {
    const uint64_t b = __builtin_amdgcn_ballot_w64(v);
    return ((b >> lane_base) & ((1ull << N) - 1ull)) != 0ull;
}
__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace rw
