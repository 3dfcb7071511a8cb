// rware_static_table.h — the specialised kernel builds (rw::StaticCfg): which shapes get a build of their own.
//
//   {H, W, N, Q, S, R}  ->  kernel with those shapes (and the launch geometry E, T) folded in at compile time.
//
// The table is cut into groups; rware_static.hip is compiled once per group (-DRW_STATIC_GROUP=g, see the Makefile) so
// the ~100 kernel instantiations build in parallel.  rware_capi.hip walks the groups in order, exact-shape entries
// (N != 0) before size-static ones (N == 0), first match wins — so WITHIN the exact-shape entries order matters only
// between entries of the same shape (geometry / batch-size variants): keep those in one group, most specific first.
#pragma once
#include "rware_kernel_table.h"

namespace rw_tab {

struct StaticEntry {
    int H, W, N, Q, S, R, E, T;
    int max_B;  // with the default geometry: chosen only for batches up to this size (0 = any); first match wins
    int image;  // 1: IMAGE / IMAGE_DICT observations (any layer list), 0: FLATTENED
    int M;      // communication bits the build was made for
    int NL;     // IMAGE builds: > 0 = the layer list baked in (`layers`: 4 bits per id, first layer lowest) with `directional`
    uint32_t layers;
    int directional;
    step_kernel_t fn, fn_rollout;
    step_kernel_t fn_nt;  // the per-step kernel with non-temporal observation stores (nullptr: `fn` switches at run time)
    int pipe;             // 1: a chunk-pipelined persistent build (rw::StaticCfg PIPE_ = 1; OP_STEP launches only, beside a classic entry)
};

enum : int { kStaticGroups = 19 };
const StaticEntry *static_group(int group, int *n);   // rware_capi.hip's view: dispatches to the per-group tables below
const StaticEntry *static_group_0(int *n);
const StaticEntry *static_group_1(int *n);
const StaticEntry *static_group_2(int *n);
const StaticEntry *static_group_3(int *n);
const StaticEntry *static_group_4(int *n);
const StaticEntry *static_group_5(int *n);
const StaticEntry *static_group_6(int *n);
const StaticEntry *static_group_7(int *n);
const StaticEntry *static_group_8(int *n);
const StaticEntry *static_group_9(int *n);
const StaticEntry *static_group_10(int *n);
const StaticEntry *static_group_11(int *n);
const StaticEntry *static_group_12(int *n);
const StaticEntry *static_group_13(int *n);
const StaticEntry *static_group_14(int *n);
const StaticEntry *static_group_15(int *n);
const StaticEntry *static_group_16(int *n);
const StaticEntry *static_group_17(int *n);
const StaticEntry *static_group_18(int *n);

}  // namespace rw_tab

#ifdef RW_STATIC_GROUP  // ------------------------------------------------------------ only inside rware_static.hip
namespace rw_tab {
namespace {
// (fn_nt: the same per-step kernel with NT_ = 1; size-static entries, N == 0, switch at run time and carry nullptr)
#define RW_NT_OR_NULL(N, ...) ((N) != 0 ? (step_kernel_t)__VA_ARGS__ : (step_kernel_t) nullptr)
#define RW_STATIC(H, W, N, Q, S, R, E, T, MAXB)                                                                   \
    {H, W, N, Q, S, R, E, T, MAXB, 0, 0, 0, 0u, -1, (step_kernel_t)rw::rware_step_kernel<R, uint8_t, rw::StaticCfg<H, W, N, Q, S, E, T>, false>, \
     (step_kernel_t)rw::rware_step_kernel<R, uint8_t, rw::StaticCfg<H, W, N, Q, S, E, T>, true>,                   \
     RW_NT_OR_NULL(N, rw::rware_step_kernel<R, uint8_t, rw::StaticCfg<H, W, N, Q, S, E, T, 0, 0, 0u, -1, 1>, false>)}
#define RW_STATIC_IMAGE(H, W, N, Q, S, R, E, T, MAXB)                                                             \
    {H, W, N, Q, S, R, E, T, MAXB, 1, 0, 0, 0u, -1, (step_kernel_t)rw::rware_step_kernel<R, uint8_t, rw::StaticCfg<H, W, N, Q, S, E, T>, false, rw::OBS_IMAGE>, \
     (step_kernel_t)rw::rware_step_kernel<R, uint8_t, rw::StaticCfg<H, W, N, Q, S, E, T>, true, rw::OBS_IMAGE>,   \
     (step_kernel_t)rw::rware_step_kernel<R, uint8_t, rw::StaticCfg<H, W, N, Q, S, E, T, 0, 0, 0u, -1, 1>, false, rw::OBS_IMAGE>}
// ... with the layer list and the directional switch baked in (the gather's per-layer selects fold away)
#define RW_STATIC_IMAGE_LAYERS(H, W, N, Q, S, R, E, T, MAXB, NL, LAYERS, DIR)                                      \
    {H, W, N, Q, S, R, E, T, MAXB, 1, 0, NL, LAYERS, DIR,                                                           \
     (step_kernel_t)rw::rware_step_kernel<R, uint8_t, rw::StaticCfg<H, W, N, Q, S, E, T, 0, NL, LAYERS, DIR>, false, rw::OBS_IMAGE>, \
     (step_kernel_t)rw::rware_step_kernel<R, uint8_t, rw::StaticCfg<H, W, N, Q, S, E, T, 0, NL, LAYERS, DIR>, true, rw::OBS_IMAGE>, \
     (step_kernel_t)rw::rware_step_kernel<R, uint8_t, rw::StaticCfg<H, W, N, Q, S, E, T, 0, NL, LAYERS, DIR, 1>, false, rw::OBS_IMAGE>}
#define RW_STATIC_MSG(H, W, N, Q, S, R, E, T, MAXB, M)                                                            \
    {H, W, N, Q, S, R, E, T, MAXB, 0, M, 0, 0u, -1, (step_kernel_t)rw::rware_step_kernel<R, uint8_t, rw::StaticCfg<H, W, N, Q, S, E, T, M>, false, rw::OBS_FLATTENED_MSG>, \
     (step_kernel_t)rw::rware_step_kernel<R, uint8_t, rw::StaticCfg<H, W, N, Q, S, E, T, M>, true, rw::OBS_FLATTENED_MSG>, \
     (step_kernel_t)rw::rware_step_kernel<R, uint8_t, rw::StaticCfg<H, W, N, Q, S, E, T, M, 0, 0u, -1, 1>, false, rw::OBS_FLATTENED_MSG>}
// the chunk-pipelined persistent build of a shape (rware_kernels.h, "PIPE"): per-step only, cached / non-temporal stores
#define RW_PIPE(H, W, N, Q, S, R, E)                                                                               \
    {H, W, N, Q, S, R, E, 256, 0, 0, 0, 0, 0u, -1,                                                                   \
     (step_kernel_t)rw::rware_step_kernel<R, uint8_t, rw::StaticCfg<H, W, N, Q, S, E, 256, 0, 0, 0u, -1, 0, 1>, false>, (step_kernel_t) nullptr, \
     (step_kernel_t)rw::rware_step_kernel<R, uint8_t, rw::StaticCfg<H, W, N, Q, S, E, 256, 0, 0, 0u, -1, 1, 1>, false>, 1}
// the three registered warehouse sizes of the RWARE papers (rware/__init__.py:7-12): grid, shelves
#define RW_TINY(N, Q) RW_STATIC(11, 10, N, Q, 32, 1, 16, 256, 0)
#define RW_SMALL(N, Q) RW_STATIC(20, 10, N, Q, 80, 1, 16, 256, 0)
#define RW_MEDIUM(N, Q) RW_STATIC(20, 16, N, Q, 144, 1, 16, 256, 0)
// 6 and 8 agents: half-size workgroups (8 envs = 48 / 64 agents, one agent wavefront) for the smaller batches, listed in front
// of the 16-env build of the same shape (first match wins).  Measured, round 3, same box (us per step, E = 8 vs 16):
//   small-8ag   B = 8192 8.05 vs 9.14 | 16384 10.33 vs 11.60 | 32768 19.9 vs 17.6      medium-8ag  7.98 vs 9.06 | 10.23 vs 11.70 | 20.1 vs 21.4
//   small-6ag   B = 8192 7.42 vs 8.42 | 16384  9.37 vs  9.85 | 32768 16.2 vs 14.5      medium-6ag-hard 7.25 vs 8.27 | 10.83 vs 9.84 | 17.2 vs 14.7
//   tiny-6ag-hard        7.53 vs 8.48 |       11.25 vs  9.92 |       17.1 vs 14.4
// -> 8 agents: up to 16384 envs; 6 agents: up to 8192 envs.
// Round 6, on today's kernel (non-temporal stores, the register budget of 8 wavefronts, wavefront priority; profiles/r06_geom3.txt, E = 8 vs 16):
//   small-8ag   B = 32768 15.8 vs 17.5 | 65536 26.8 vs 30.1 | 131072 55.4 vs 56.5    tiny-8ag 16.0 vs 17.7 | 26.8 vs 30.6 | 54.8 vs 57.0
//   medium-8ag  15.8 vs 17.5 | 28.3 vs 29.4 | 56.6 vs 57.0        6 agents stay: medium-6ag-hard 13.7 vs 13.6 | 23.0 vs 22.7 | 43.5 vs 45.1
// -> 8 agents: 8 envs per workgroup at EVERY batch (the 16-env build stays for explicit geometries).
#define RW_E8_MAXB(N) ((N) >= 8 ? 0 : 8192)
#define RW_TINY_E8(N, Q) RW_STATIC(11, 10, N, Q, 32, 1, 8, 256, RW_E8_MAXB(N)), RW_TINY(N, Q)
#define RW_SMALL_E8(N, Q) RW_STATIC(20, 10, N, Q, 80, 1, 8, 256, RW_E8_MAXB(N)), RW_SMALL(N, Q)
#define RW_MEDIUM_E8(N, Q) RW_STATIC(20, 16, N, Q, 144, 1, 8, 256, RW_E8_MAXB(N)), RW_MEDIUM(N, Q)

// 2 agents: double-size workgroups (32 envs = 64 agents, one full agent wavefront) from 16384 envs up — the 16-env build in
// front of it serves the smaller batches, and again behind it the batches that are no multiple of 32.  Measured, round 3,
// same box (E = 16 vs 32): small-2ag B = 4096 4.38 vs 4.41 | 16384 5.87 vs 5.70 | 65536 13.8 vs 10.6; tiny-2ag 4.45 vs 4.44 |
// 5.96 vs 5.79 | 12.8 vs 10.3.  (The pattern across N = 2, 4, 6, 8: about 64 agents per workgroup.)
#define RW_TINY_E32(N, Q) RW_STATIC(11, 10, N, Q, 32, 1, 16, 256, 16383), RW_STATIC(11, 10, N, Q, 32, 1, 32, 256, 0), RW_TINY(N, Q)
#define RW_SMALL_E32(N, Q) RW_STATIC(20, 10, N, Q, 80, 1, 16, 256, 16383), RW_STATIC(20, 10, N, Q, 80, 1, 32, 256, 0), RW_SMALL(N, Q)
#define RW_MEDIUM_E32(N, Q) RW_STATIC(20, 16, N, Q, 144, 1, 16, 256, 16383), RW_STATIC(20, 16, N, Q, 144, 1, 32, 256, 0), RW_MEDIUM(N, Q)

// agent-count-static entries (Q == -1) with the geometry variants of their agent count
#define RW_QRT_12(H, W, S, N) RW_STATIC(H, W, N, -1, S, 1, 16, 256, 16383), RW_STATIC(H, W, N, -1, S, 1, 32, 256, 0), RW_STATIC(H, W, N, -1, S, 1, 16, 256, 0)
#define RW_QRT_34(H, W, S, N) RW_STATIC(H, W, N, -1, S, 1, 16, 256, 0)
// (these builds hold the run-time queue length and what depends on it in scalar registers — 104 SGPRs, 7 wavefronts per SIMD
//  instead of 8 — so their 8-env variant serves batches of up to 7 workgroups per CU: 14336 envs, not 16384.  Forcing the
//  register budget of 8 wavefronts (amdgpu_waves_per_eu) was measured: small-8ag B = 16384 12.73 -> 11.14 us, but +0.1 .. +0.4 us
//  wherever the batch fits anyway — the spilled scalars cost more than they buy.)
#define RW_QRT_58(H, W, S, N) RW_STATIC(H, W, N, -1, S, 1, 8, 256, ((N) >= 7 ? 14336 : 8192)), RW_STATIC(H, W, N, -1, S, 1, 16, 256, 0)
// 9 .. 19 agents (rware/__init__.py:16 registers every count up to 19): 8 envs per workgroup, with the register budget of 8
// wavefronts per SIMD (rw::want_occ8) and the per-cell agent phases (kCell).  Measured, round 4, B = 16384, same box, us per step:
//   all-gather agent phases, E = 8 vs 16:  small-9ag 13.35 / 13.05 | 10ag 13.80 / 14.23 | 12ag 15.67 / 16.52 | 13ag 18.48 / 18.45 |
//       14ag 20.75 / 19.79 | 16ag 22.81 / 21.60 | large-16ag 22.38 / 21.18 (E = 4: 20.8)   (from 14 agents on the budget spilled)
//   per-cell agent phases (54 .. 62 VGPRs at every agent count), E = 8 vs 16:  10ag 13.0 / 13.9 | 14ag 16.9 / 17.5 | 16ag 18.0 / 18.8 |
//       large-16ag 18.0 / 19.0 (E = 4: 17.4) | 17ag 19.6 | 19ag 21.3
//   round 6, two-wavefront workgroups (T = 128, 16 per CU, register budget of 8 wavefronts): E = 4 / 8 — small-10ag 14.4 / 14.7 against 13.1,
//       large-16ag 18.4 / 20.1 against 18.0, large-16ag r = 2 34.5 / 38.0 against 35.1 (profiles/r06_t128_stagger.txt): not kept
// -> 8 envs per workgroup for every count; 16 (up to 16 agents: three agents' envs of 17 .. 19 fit a wavefront, 12 envs at most per
//    4-wavefront workgroup) and 4 as explicit geometries and for batches that are no multiple of 8.
#define RW_QRT_WIDE(H, W, S, N) RW_QRT_WIDE_##N(H, W, S, N)
#define RW_QRT_WIDE_LO(H, W, S, N) RW_STATIC(H, W, N, -1, S, 1, 8, 256, 0), RW_STATIC(H, W, N, -1, S, 1, 16, 256, 0), RW_STATIC(H, W, N, -1, S, 1, 4, 256, 0)
#define RW_QRT_WIDE_HI(H, W, S, N) RW_STATIC(H, W, N, -1, S, 1, 8, 256, 0), RW_STATIC(H, W, N, -1, S, 1, 4, 256, 0)
#define RW_QRT_WIDE_9 RW_QRT_WIDE_LO
#define RW_QRT_WIDE_10 RW_QRT_WIDE_LO
#define RW_QRT_WIDE_11 RW_QRT_WIDE_LO
#define RW_QRT_WIDE_12 RW_QRT_WIDE_LO
#define RW_QRT_WIDE_13 RW_QRT_WIDE_LO
#define RW_QRT_WIDE_14 RW_QRT_WIDE_LO
#define RW_QRT_WIDE_15 RW_QRT_WIDE_LO
#define RW_QRT_WIDE_16 RW_QRT_WIDE_LO
#define RW_QRT_WIDE_17 RW_QRT_WIDE_HI
#define RW_QRT_WIDE_18 RW_QRT_WIDE_HI
#define RW_QRT_WIDE_19 RW_QRT_WIDE_HI
#define RW_QRT_WIDE_A(H, W, S) RW_QRT_WIDE(H, W, S, 9), RW_QRT_WIDE(H, W, S, 10), RW_QRT_WIDE(H, W, S, 11), RW_QRT_WIDE(H, W, S, 12), \
                               RW_QRT_WIDE(H, W, S, 13), RW_QRT_WIDE(H, W, S, 14)
#define RW_QRT_WIDE_B(H, W, S) RW_QRT_WIDE(H, W, S, 15), RW_QRT_WIDE(H, W, S, 16), RW_QRT_WIDE(H, W, S, 17), RW_QRT_WIDE(H, W, S, 18), \
                               RW_QRT_WIDE(H, W, S, 19)
// (the tiny warehouse: 110 cells — a 4-env shelf chunk is no whole number of 16-byte DMA pieces, so 8 envs only)
#define RW_QRT_WIDE8(H, W, S, N) RW_STATIC(H, W, N, -1, S, 1, 8, 256, 0)
#define RW_QRT_WIDE8_A(H, W, S) RW_QRT_WIDE8(H, W, S, 9), RW_QRT_WIDE8(H, W, S, 10), RW_QRT_WIDE8(H, W, S, 11), RW_QRT_WIDE8(H, W, S, 12), \
                                RW_QRT_WIDE8(H, W, S, 13), RW_QRT_WIDE8(H, W, S, 14)
#define RW_QRT_WIDE8_B(H, W, S) RW_QRT_WIDE8(H, W, S, 15), RW_QRT_WIDE8(H, W, S, 16), RW_QRT_WIDE8(H, W, S, 17), RW_QRT_WIDE8(H, W, S, 18), \
                                RW_QRT_WIDE8(H, W, S, 19)
#define RW_QRT_SIZE(H, W, S) RW_QRT_12(H, W, S, 1), RW_QRT_12(H, W, S, 2), RW_QRT_34(H, W, S, 3), RW_QRT_34(H, W, S, 4), \
                             RW_QRT_58(H, W, S, 5), RW_QRT_58(H, W, S, 6), RW_QRT_58(H, W, S, 7), RW_QRT_58(H, W, S, 8)

const StaticEntry kEntries[] = {
#if RW_STATIC_GROUP == 0
    // ---- the BASELINE.json tasks (+ their batch-size / geometry variants)
    // half-size workgroups for batches that leave the CUs short of workgroups at E = 16 (measured, round 2:
    // medium-6ag-hard B=8192 9.16 -> 7.98 us, B=4096 7.86 -> 6.71; B=16384 11.2 vs 13.1 the other way round)
    RW_STATIC(20, 16, 6, 3, 144, 1, 8, 256, 8192),
    RW_TINY_E32(2, 2),                             // rware-tiny-2ag (BASELINE config 2 runs the 16-env build: B = 4096)
    RW_STATIC(20, 10, 4, 4, 80, 1, 16, 256, 0),    // rware-small-4ag (headline)
    // (small-4ag with 8 envs per workgroup: since the agent phases run in registers the 16-env build wins at every batch
    //  size — B=1024 4.77 vs 4.81 us, 4096 5.35 vs 5.63, 16384 7.87 vs 10.3 — so this one only serves batches that are
    //  a multiple of 8 but not of 16, or an explicit geometry)
    RW_STATIC(20, 10, 4, 4, 80, 1, 8, 256, 0),
    RW_STATIC(20, 16, 6, 3, 144, 1, 16, 256, 0),   // rware-medium-6ag-hard
    // (round 6: still smaller workgroups for the small BASELINE launches lose — medium-6ag-hard x 8192 with 4 envs 8.45 us against 6.28,
    //  x 4096 5.88 against 5.65; tiny-2ag x 4096 with 8 envs 4.55 against 4.42, x 2048 4.24 against 4.19: profiles/r06_small_geom.txt)
    // (large-16ag r=2.  Round 3, same box: E = 8 36.2 us at B = 16384 vs 38.4 with E = 4 and 37.1 with E = 16; B = 4096: 14.6 vs 13.8
    //  with E = 4.  Round 4, agents in registers, same box: B = 16384 37.27 (E = 8) vs 37.29 (E = 4), both cached; B = 32768, past
    //  the Infinity Cache, non-temporal: 79.0 vs 75.3 -> 4 envs per workgroup in front)
    RW_STATIC(29, 16, 16, 16, 224, 2, 4, 256, 0),  // rware-large-16ag, sensor_range = 2
    RW_STATIC(29, 16, 16, 16, 224, 2, 8, 256, 0),  // (explicit geometry)
#elif RW_STATIC_GROUP == 1
    // ---- the "next" observation kinds callers hit first: IMAGE / IMAGE_DICT (any layer list, directional
    // or not) and FLATTENED with 1 or 2 communication bits, on the two smallest BASELINE tasks
    // (first the reference's default layer list — SHELVES, REQUESTS, AGENTS, GOALS, ACCESSIBLE, directional — baked in,
    //  then the any-list builds)
    RW_STATIC_IMAGE_LAYERS(20, 10, 4, 4, 80, 1, 16, 256, 0, 5, 0x65210u, 1),
    RW_STATIC_IMAGE_LAYERS(11, 10, 2, 2, 32, 1, 16, 256, 0, 5, 0x65210u, 1),
    RW_STATIC_IMAGE(20, 10, 4, 4, 80, 1, 16, 256, 0),
    RW_STATIC_IMAGE(11, 10, 2, 2, 32, 1, 16, 256, 0),
    RW_STATIC_MSG(20, 10, 4, 4, 80, 1, 16, 256, 0, 1),
    RW_STATIC_MSG(20, 10, 4, 4, 80, 1, 16, 256, 0, 2),
    RW_STATIC_MSG(11, 10, 2, 2, 32, 1, 16, 256, 0, 2),
#elif RW_STATIC_GROUP == 2
    // ---- the task grid of the RWARE benchmark papers (Papoudakis et al. 2021; Christianos et al. 2020): tiny / small /
    // medium x 2, 4, 6, 8 agents x easy / normal / hard.  request_queue_size = int(n_agents * {2, 1, 0.5})
    // (rware/__init__.py:14-21).  Shapes already listed in group 0 are not repeated.
    RW_TINY_E32(2, 4), RW_TINY(4, 8), RW_TINY_E8(6, 12), RW_TINY_E8(8, 16),    // -easy
    RW_TINY(4, 4), RW_TINY_E8(6, 6), RW_TINY_E8(8, 8),                     // normal (tiny-2ag: group 0)
    RW_TINY_E32(2, 1), RW_TINY(4, 2), RW_TINY_E8(6, 3), RW_TINY_E8(8, 4),      // -hard
#elif RW_STATIC_GROUP == 3
    RW_SMALL_E32(2, 4), RW_SMALL(4, 8), RW_SMALL_E8(6, 12), RW_SMALL_E8(8, 16),  // -easy
    RW_SMALL_E32(2, 2), RW_SMALL_E8(6, 6), RW_SMALL_E8(8, 8),                  // normal (small-4ag: group 0)
    RW_SMALL_E32(2, 1), RW_SMALL(4, 2), RW_SMALL_E8(6, 3), RW_SMALL_E8(8, 4),    // -hard
#elif RW_STATIC_GROUP == 4
    RW_MEDIUM_E32(2, 4), RW_MEDIUM(4, 8), RW_MEDIUM_E8(6, 12), RW_MEDIUM_E8(8, 16),  // -easy
    RW_MEDIUM_E32(2, 2), RW_MEDIUM(4, 4), RW_MEDIUM_E8(6, 6), RW_MEDIUM_E8(8, 8),  // normal
    RW_MEDIUM_E32(2, 1), RW_MEDIUM(4, 2), RW_MEDIUM_E8(8, 4),                   // -hard (medium-6ag-hard: group 0)
#elif RW_STATIC_GROUP == 5
    // ---- size-static builds (N == 0: any agent count / queue length): every other registered id, sensor_range 1
    RW_STATIC(11, 10, 0, 0, 32, 1, 16, 256, 0),    // rware-tiny-*
    RW_STATIC(20, 10, 0, 0, 80, 1, 16, 256, 0),    // rware-small-*
    RW_STATIC(20, 16, 0, 0, 144, 1, 16, 256, 0),   // rware-medium-*
    RW_STATIC(29, 16, 0, 0, 224, 1, 16, 256, 0),   // rware-large-*
    // ... and with sensor_range = 2 (the 5 x 5 window of BASELINE config 5, on every size)
    RW_STATIC(11, 10, 0, 0, 32, 2, 16, 256, 0), RW_STATIC(20, 10, 0, 0, 80, 2, 16, 256, 0),
    RW_STATIC(20, 16, 0, 0, 144, 2, 16, 256, 0), RW_STATIC(29, 16, 0, 0, 224, 2, 16, 256, 0),
#elif RW_STATIC_GROUP == 6
    // ---- agent-count-static builds (Q == -1: request-queue length read at run time, any Q <= 2 N): the easy / normal / hard
    // variants of a task and custom queue sizes share one build.  Geometry by the 64-agents-per-workgroup rule.
    RW_QRT_SIZE(20, 10, 80),     // small
#elif RW_STATIC_GROUP == 7
    RW_QRT_SIZE(11, 10, 32),     // tiny
#elif RW_STATIC_GROUP == 8
    RW_QRT_SIZE(20, 16, 144),    // medium
#elif RW_STATIC_GROUP == 9
    RW_QRT_SIZE(29, 16, 224),    // large
#elif RW_STATIC_GROUP == 10
    // ---- agent-count-static builds for 9 .. 19 agents (agent phases in registers, chain links in 64 / 128 bits)
    RW_QRT_WIDE_A(20, 10, 80),   // small
#elif RW_STATIC_GROUP == 11
    RW_QRT_WIDE_B(20, 10, 80),
#elif RW_STATIC_GROUP == 12
    RW_QRT_WIDE8_A(11, 10, 32),  // tiny
#elif RW_STATIC_GROUP == 13
    RW_QRT_WIDE8_B(11, 10, 32),
#elif RW_STATIC_GROUP == 14
    RW_QRT_WIDE_A(20, 16, 144),  // medium
#elif RW_STATIC_GROUP == 15
    RW_QRT_WIDE_B(20, 16, 144),
#elif RW_STATIC_GROUP == 16
    RW_QRT_WIDE_A(29, 16, 224),  // large
#elif RW_STATIC_GROUP == 17
    RW_QRT_WIDE_B(29, 16, 224),
#elif RW_STATIC_GROUP == 18
    // ---- chunk-pipelined persistent builds (pipe == 1): the agent phases of a chunk on one wavefront (E * N <= 64), first match wins
    RW_PIPE(20, 10, 4, 4, 80, 1, 16),      // rware-small-4ag
    RW_PIPE(20, 16, 6, 3, 144, 1, 8),      // rware-medium-6ag-hard
    RW_PIPE(29, 16, 16, 16, 224, 2, 4),    // rware-large-16ag, sensor_range = 2
    RW_PIPE(11, 10, 2, 2, 32, 1, 32),      // rware-tiny-2ag
    RW_PIPE(20, 10, 10, -1, 80, 1, 4),     // small, 10 agents (any queue length): the per-cell agent phases
    RW_PIPE(29, 16, 16, -1, 224, 1, 4),    // rware-large-16ag
#else
#error "RW_STATIC_GROUP out of range"
#endif
};
#undef RW_STATIC
#undef RW_PIPE
#undef RW_NT_OR_NULL
#undef RW_STATIC_IMAGE
#undef RW_STATIC_IMAGE_LAYERS
#undef RW_STATIC_MSG
#undef RW_TINY
#undef RW_TINY_E8
#undef RW_QRT_12
#undef RW_QRT_SIZE
#undef RW_QRT_WIDE
#undef RW_QRT_WIDE_A
#undef RW_QRT_WIDE_B
#undef RW_QRT_WIDE8
#undef RW_QRT_WIDE8_A
#undef RW_QRT_WIDE8_B
#undef RW_QRT_34
#undef RW_QRT_58
#undef RW_TINY_E32
#undef RW_SMALL_E32
#undef RW_MEDIUM_E32
#undef RW_SMALL_E8
#undef RW_MEDIUM_E8
#undef RW_SMALL
#undef RW_MEDIUM
}  // namespace
}  // namespace rw_tab
#endif  // RW_STATIC_GROUP
