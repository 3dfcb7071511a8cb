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
__device__ __forceinline__ void wave_sync() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_wave_barrier();
}

}  // namespace rw
