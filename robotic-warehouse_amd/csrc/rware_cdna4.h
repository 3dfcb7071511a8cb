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
//   sload16_issue / sload8_issue / sload_wait
//                                a scalar load whose ISSUE point is fixed by the source: hipcc sinks an
//                                ordinary `p->field` load down to its first use, which exposes the whole
//                                L2 round trip there.  The issue half returns SGPRs that are NOT valid
//                                until sload_wait() (s_waitcnt lgkmcnt(0)) has been called on them; do
//                                not read them in between.
#pragma once
#include <stdint.h>

namespace rw {

__device__ __forceinline__ void lds_dma_b128(const void *g_lane, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g_lane,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 16, 0, 0);
}
__device__ __forceinline__ void lds_dma_b32(const void *g_lane, void *lds_wave_base) {
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)g_lane,
                                     (__attribute__((address_space(3))) void *)lds_wave_base, 4, 0, 0);
}
__device__ __forceinline__ void lds_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}
typedef uint32_t sreg16 __attribute__((ext_vector_type(16)));
typedef uint32_t sreg8 __attribute__((ext_vector_type(8)));
template <int kByteOffset>
__device__ __forceinline__ sreg16 sload16_issue(uint64_t uniform_base) {
    sreg16 r;
    asm volatile("s_load_dwordx16 %0, %1, %2" : "=&s"(r) : "s"(uniform_base), "n"(kByteOffset));
    return r;
}
template <int kByteOffset>
__device__ __forceinline__ sreg8 sload8_issue(uint64_t uniform_base) {
    sreg8 r;
    asm volatile("s_load_dwordx8 %0, %1, %2" : "=&s"(r) : "s"(uniform_base), "n"(kByteOffset));
    return r;
}
__device__ __forceinline__ void sload_wait(sreg16 &a, sreg8 &b) {
    asm volatile("s_waitcnt lgkmcnt(0)" : "+s"(a), "+s"(b));
}
template <typename P, typename V>
__device__ __forceinline__ P *sreg_ptr(const V &r, int k) {  // pointer #k of a block of 64-bit pointers
    return reinterpret_cast<P *>((uint64_t)r[2 * k] | ((uint64_t)r[2 * k + 1] << 32));
}
// HW_ID (hwreg 4): wave[3:0] simd[5:4] pipe[7:6] cu[11:8] sh[12] se[15:13]; XCC_ID (hwreg 20): xcc[3:0]
__device__ __forceinline__ uint32_t hw_id() { return __builtin_amdgcn_s_getreg((31 << 11) | 4); }
__device__ __forceinline__ uint32_t xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20); }
// uniform(x): tells the compiler a value is wave-uniform (v_readfirstlane), so tests on it become scalar branches
__device__ __forceinline__ int uniform(int x) { return __builtin_amdgcn_readfirstlane(x); }
__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace rw
