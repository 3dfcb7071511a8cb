// rware_kernels.h — CDNA4 (gfx950) device code of the vectorised RWARE step engine.
//
// One fused kernel advances `envs_per_wg` independent warehouses per workgroup through every
// phase of rware.warehouse.Warehouse.step (rware/warehouse.py:804-946) and the FLATTENED
// observation gather (:598-674), with the per-env working set staged in LDS:
//
//   P0  HBM -> LDS by LDS-DMA (global_load_lds_dwordx4): the packed agent records (one dword per
//       agent, rec_pack below), the request queue, the actions and the env's SHELF layer.  The
//       kernel reads the shelf layer from a compact shadow (uint8/uint16 per cell,
//       `shelf_shadow`) rather than from the exported int32 grid: the grid is 93 % of a step's
//       read bytes, the shadow is 1/8 of it.  The agent layer is not read at all — it is rebuilt
//       in LDS from the agent coordinates.  The int32 grid [B][2][H][W] and the five int32 agent
//       arrays callers see are DERIVED VIEWS, rebuilt on demand by small host-path kernels.
//   AG  the per-agent phases, one lane per (env, agent), all agents of an env inside ONE
//       wavefront, ordered by wave-local LDS syncs (no workgroup barrier); written branch-free
//       (a wavefront holds every action and heading at once); a wave-uniform ballot skips the
//       chain bookkeeping when nobody steps onto an occupied cell:
//         P1  move intent + shelf-block cancel              (:825-846, Agent.req_location :102-116)
//         P2  collision resolution in closed form            (:848-876; notes below)
//         P3  apply (move / turn / load / unload) + incremental grid update instead of
//             _recalc_grid                                   (:880-901, :749-755)
//         P5  goals, request replacement (numpy-exact PCG64 draw), rewards, termination (:903-942)
//   RS  on-device reset for autoreset / rw_reset, numpy-exact draws (rare path)   (:757-802)
//   WB  state write-back in three roles: per-env counters/flags + queue; agent records + rewards
//       (coalesced); the shadow patch of the <= 2N shelf cells that changed.
//   OS  the self part of the observation (own coordinates, load, heading, on-highway).
//   P7  observation (:598-674): per (agent, window row) the 7-bit cell codes are OR-ed into ONE
//       contiguous bit string per workgroup (bit g == obs element g of the chunk), so float4 #q is
//       nibble #q; dwordx4 stores, the two coordinate floats per agent in a small second pass.
//       IMAGE / IMAGE_DICT observations (:527-596) use the same bit string (kImage).
//   Order after AG.  With 4 wavefronts and a small observation chunk, wavefront 3 is a service
//   wave: OS while wavefronts 0..2 gather P7, then all of WB while they expand and store, so the
//   observation stream — what the step ends with — starts as early as possible.  Otherwise
//   WB -> OS/P7 -> stores, one role per wavefront (the barriers after P0 wait on LDS only, so
//   the WB stores drain underneath P7).
//   The rollout variant (kRollout) wraps AG..P7 in a step loop: the chunk stays in LDS for T steps.
//
// Roofline: integer/indexing work, no MFMA; bound by HBM bytes.  Algorithmic bytes per
// env-step A = 8HW + 4N + 40N + 4Q + 16 + 4NL + 4N + 4; the shadow makes the
// real traffic smaller than A.
//
// Collision resolution, closed form.  The reference builds a digraph on cells with one
// out-edge per agent (start -> target) and, per weakly connected component, commits either
// the agents on its cycle (but nobody if the cycle is a 2-swap) or the agents on the longest
// path.  Because every node has out-degree <= 1, each component is an in-tree draining into
// one sink or one cycle, so with nxt(i) = the agent standing on i's target cell:
//   - i stationary (target == start, incl. wall-clamped FORWARD and cancelled moves): commits.
//   - depth(i) = longest chain of movers following i (atomicMax walk, <= N hops).
//   - win(i)   = i holds the largest (depth, then LOWEST id) among movers with the same target
//                (a scan over the env's N agents; contiguous LDS reads).
//   - walk i -> nxt(i) -> ...: reaches an empty cell  => commit iff every agent on the walk wins;
//                              reaches a stationary agent => fail;
//                              returns to i after len hops => commit iff len >= 3 (cycle);
//                              N hops without either       => i feeds a cycle => fail.
// The tie rule (lowest agent id among equal depths) is the pinned rule of DESIGN.md §tie-break.
#pragma once
#ifndef __HIPCC_RTC__
#include <stddef.h>
#include <stdint.h>

#include <type_traits>
#endif

#include <rware_cdna4.h>

#include "rware_pcg64.h"

// RW_RARE(c): c, marked unlikely — the compiler lays the guarded block out of line, so the common path stays one
// sequential run of code (every launch starts with a cold instruction cache: a taken branch over a big rare block is a
// fetch miss for the first wavefront that gets there)
#define RW_RARE(c) __builtin_expect(!!(c), 0)
// RW_INLINE: a lambda whose body goes into its caller before any optimisation runs (the pipelined flow's helpers)
#define RW_INLINE __attribute__((always_inline))

namespace rw {

// (the two type-level helpers the kernel needs, spelled out: the run-time compiler — hipRTC, rware_jit.cpp — has no <type_traits>)
template <bool C, typename A, typename B> struct pick_type { using type = A; };
template <typename A, typename B> struct pick_type<false, A, B> { using type = B; };
struct yes_t { static constexpr bool value = true; };
struct no_t { static constexpr bool value = false; };

enum : int { OP_STEP = 0, OP_RESET = 1, OP_OBS = 2 };
enum : int { ACT_NOOP = 0, ACT_FORWARD = 1, ACT_LEFT = 2, ACT_RIGHT = 3, ACT_TOGGLE = 4 };
enum : int { DIR_UP = 0, DIR_DOWN = 1, DIR_LEFT = 2, DIR_RIGHT = 3 };
enum : int { REW_GLOBAL = 0, REW_INDIVIDUAL = 1, REW_TWO_STAGE = 2 };
enum : int { AR_DISABLED = 0, AR_NEXT_STEP = 1, AR_SAME_STEP = 2 };
enum : int { STATUS_INVALID_ACTION = 1,
             // an AGENT_DIRECTION / AGENT_LOAD image layer of the reference would have raised IndexError (:552, :558)
             STATUS_IMAGE_INDEX = 2 };
enum : int { MAX_GOALS = 16, MAX_IMAGE_LAYERS = 8 };
enum : int { OBS_FLATTENED = 0, OBS_IMAGE = 1, OBS_FLATTENED_MSG = 2, OBS_IMAGE_MSG = 3 };  // _MSG: msg_bits > 0
// ImageLayer values of the reference (rware/warehouse.py:59-70); 3 and 4 are rejected by the host
enum : int { LAYER_SHELVES = 0, LAYER_REQUESTS = 1, LAYER_AGENTS = 2, LAYER_AGENT_DIRECTION = 3, LAYER_AGENT_LOAD = 4,
             LAYER_GOALS = 5, LAYER_ACCESSIBLE = 6 };

struct Params {
    // The twelve pointers the P0 stage-in needs come first, contiguous and cache-line aligned: the
    // compiler fetches them with two wide scalar loads.
    void *shelf_shadow;            // CellT [B][HW] (+ padding): compact copy of grid layer 1, the kernel's read path
    uint32_t *arec;                // [B][N] packed agent records (rec_pack below): the agents' state as the kernels keep it
    int32_t *queue;
    const uint32_t *highway_bits;  // [HWW] bit c == highways[c]                  (static per config)
    int32_t *counters;             // [B][2] the per-env counter record the kernels keep: {steps | need_reset << 31, inactive} — ONE
                                   // 8-byte load and ONE 8-byte store per env-step (a whole 128-byte line per 16-env workgroup)
                                   // where three arrays were read and two or three written; RW_BUF_STEPS / _INACTIVE / _NEED_RESET
                                   // are derived views (rware_unpack_counters_kernel), like the agent arrays
    int32_t *steps, *inactive;     // the exported views (host paths only)
    uint8_t *need_reset;           // [B]
    uint64_t *rng;                 // [6][B]
    // config
    int32_t B, H, W, HW, N, Q, S, SW;  // SW = dwords of the requested-shelf bitmap = (S+32)/32
    int32_t HWW;                       // dwords of the highway bitmap = (HW+31)/32
    int32_t n_goals, max_inactivity, max_steps, reward_type, autoreset, normalised;
    int32_t envs_per_wg;
    int32_t nt_obs;                    // 1: the observation stream is stored with the non-temporal hint (see ST)
    int32_t groups_per_wave;           // envs whose agents share one wavefront = 64 / N
    uint32_t magic_n;                  // ceil(2^18 / N): x / N == (x * magic) >> 18 for x * N < 2^18
    int32_t goal_cells[MAX_GOALS];     // cell index y*W+x per goal, list order
    const int32_t *shelf_init;     // [HW] shelf layer right after reset: ids 1..S row-major on non-highway cells
    // state (device)
    // exported int32 views RW_BUF_AGENT_X .. _DELIVERED: derived from `arec` on demand (rware_unpack_agents_kernel), never
    // touched by the step kernels — five store streams and five load streams per step that the step does not issue
    int32_t *ax, *ay, *adir, *acarry, *adeliv;
    int32_t *grid;
    uint8_t *truncated;   // [B]
    int32_t *status;      // [1] sticky error bits
    // IMAGE / IMAGE_DICT observations (rware/warehouse.py:527-596); unused by the FLATTENED kernels
    int32_t n_layers, directional;
    int32_t transposed_layers;  // bit 0: AGENT_DIRECTION requested, bit 1: AGENT_LOAD requested
    int32_t layers[MAX_IMAGE_LAYERS];
    float *features;      // [B][N][6] one-hot direction, on_highway, carrying (IMAGE_DICT), or nullptr
    // communication bits (rware/warehouse.py:255-259, 660-667, 810-812); only the OBS_FLATTENED_MSG kernels
    int32_t msg_bits;     // M: an action is [Action, bit_0 .. bit_{M-1}] per agent, L = 8 + (7 + M)(2r+1)^2
    int32_t *amsg;        // [B][N] bit k == message[k]
    // SAME_STEP autoreset: where the terminal observation of an env goes when the step that ends its episode also resets it
    // (RW_BUF_FINAL_OBS, [B][N][L] — the image for the IMAGE types); nullptr otherwise
    float *final_obs;
    float *final_features;  // ... and its IMAGE_DICT feature vectors (RW_BUF_FINAL_FEATURES, [B][N][6]), or nullptr
};

// What changes from launch to launch.  The kernel-argument segment is rewritten by the host for every
// launch and is therefore cold in the scalar cache and in L2 (~0.3 us per dependent fetch, measured);
// so it carries only this small block, fetched once, while the large constant `Params` block lives in
// device memory, stays warm in L2 across launches and is read through a pointer.
struct LaunchArgs {
    const int32_t *actions;     // [B][N] (or the [T][B][N] tape of a fused rollout)   (OP_STEP)
    int32_t op;                 // OP_STEP / OP_RESET / OP_OBS, | OP_FLAG_TIMELINE
    // fused rollout (rw_step_many_device): n_steps consecutive steps in ONE launch; the env chunk stays
    // in LDS between steps, only actions are read and obs/rewards/terminated written per step.
    int32_t n_steps;
    float *obs;                 // [B][N][L]
    float *rewards;             // [B][N]
    uint8_t *terminated;        // [B]
    const uint8_t *reset_mask;  // [B]                                                  (OP_RESET)
    uint64_t *timeline;         // nullptr, or [n_wg][TL_MARKS] wall-clock stamps (rw_debug_timeline)
    // Strides are in elements per step (0 == every step writes the same buffer).
    int64_t act_stride, obs_stride, rew_stride, term_stride;
};
// The kernel takes these fields as separate scalar arguments after `cp`: with
// -amdgpu-kernarg-preload-count=16 the first 14 dwords — cp and everything up to reset_mask — arrive in
// SGPRs at wave launch, so nothing on the stage-in path waits on the (cold) kernel-argument segment.
#define RW_LAUNCH_PARAMS                                                                                   \
    const int32_t *la_actions, const int32_t la_op, const int32_t la_n_steps, float *la_obs,               \
        float *la_rewards, uint8_t *la_terminated, const uint8_t *la_reset_mask, uint64_t *la_timeline,    \
        const int64_t la_act_stride, const int64_t la_obs_stride, const int64_t la_rew_stride,             \
        const int64_t la_term_stride
#define RW_LAUNCH_PARAMS_TYPES                                                                              \
    const rw::Params *, const int32_t *, const int32_t, const int32_t, float *, float *, uint8_t *, const uint8_t *,  \
        uint64_t *, const int64_t, const int64_t, const int64_t, const int64_t
#define RW_LAUNCH_ARGS(la)                                                                                  \
    (la).actions, (la).op, (la).n_steps, (la).obs, (la).rewards, (la).terminated, (la).reset_mask,         \
        (la).timeline, (la).act_stride, (la).obs_stride, (la).rew_stride, (la).term_stride
enum : int { OP_FLAG_TIMELINE = 0x100 };
enum : int { TL_START = 0, TL_ZEROED, TL_DMA_ISSUED, TL_ENV_LOADED, TL_LOADED, TL_AGENTS, TL_RESET, TL_OBS_BITS,
             TL_OBS_STORED, TL_END,  // 10, 11: where the wavefronts ran
             TL_AG_RECORD = 12, TL_AG_CELLS, TL_AG_WINNERS, TL_AG_APPLIED, TL_AG_GOALS,  // inside the agent phases (wavefront 0)
             // the pipelined build, chunk it < 4 of the workgroup, slot TL_PIPE + 8 * it + ...: behind barrier A | wavefront 0 done gathering |
             // wavefront 3 done with self bits + write-back | behind barrier B | wavefront 0 done with the next chunk's agent phases |
             // wavefront 1 done expanding | wavefront 3 has issued the stage-in of chunk it + 2 | ... and has seen it land (next stage)
             TL_PIPE = 16, TL_MARKS = 48 };

// LDS carve-up, in dwords.  Every sub-array starts on a 16-byte boundary.
struct LdsLayout {
    // DMA destinations, contiguous in exactly this order (the static builds fill them with ONE linear
    // LDS-DMA stream): shelf layer, agent arrays (the packed records land in `ax` and are unpacked in place; in the
    // kDirect builds the agent lanes publish them), actions, queue, highway bitmap, per-env counter records, reset mask
    int gs, ax, ay, dir, carry, deliv, act, queue, hw, dcnt, dflag, dma_end;
    int ga, zero_end;  // cleared every launch
    int tgt, nxt, depth, win, rew, mv, msg, fx, fy, req, obits, envi, misc, total;
};
enum : int { ENVI_STEPS = 0, ENVI_INACTIVE = 1, ENVI_RESET = 2, ENVI_DONE = 3, ENVI_SKIP = 4,
             ENVI_QDIRTY = 5,  // a request was replaced since the chunk was staged: the queue has to be written back
             ENVI_W = 8 };

RW_HD int rw_up4(int x) { return (x + 3) & ~3; }
RW_HD uint32_t rw_magic18(int d) { return d > 0 ? (uint32_t)(((1u << 18) + (uint32_t)d - 1u) / (uint32_t)d) : 0u; }
RW_HD int rw_div18(int x, uint32_t magic) { return (int)(((uint32_t)x * magic) >> 18); }

RW_HD LdsLayout make_lds_layout(int E, int N, int Q, int HW, int SW, int OW, int cell_bytes, int act_words = 1) {
    LdsLayout l;
    int o = 0;
    const int en = rw_up4(E * N);
    l.gs = o;
    o += rw_up4((E * HW * cell_bytes + 3) / 4);  // shelf layer, CellT per cell
    l.ax = o;     o += en;
    l.ay = o;     o += en;
    l.dir = o;    o += en;
    l.carry = o;  o += en;
    l.deliv = o;  o += en;
    l.act = o;    o += rw_up4(E * N * act_words);  // [Action, message bits...] per agent
    l.queue = o;  o += rw_up4(E * Q);
    l.hw = o;     o += rw_up4((HW + 31) / 32);
    l.dcnt = o;   o += rw_up4(2 * E);                          // counter records {steps | need_reset << 31, inactive}
    l.dflag = o;  o += rw_up4((E + 3) / 4);                    // bytes: the reset mask (OP_RESET only)
    l.dma_end = o;
    l.ga = o;     o += rw_up4((E * HW + 3) / 4);               // agent layer, 1 byte per cell: id | 0x80 if loaded
    l.zero_end = o;
    l.tgt = o;    o += en;
    l.nxt = o;    o += en;
    l.depth = o;  o += en;
    l.win = o;    o += en;
    l.rew = o;    o += en;
    l.mv = o;     o += en;
    l.msg = o;    o += en;
    l.fx = o;     o += en;
    l.fy = o;     o += en;
    l.req = o;    o += rw_up4(E * SW);
    l.obits = o;  o += rw_up4(E * N * OW + 4);  // one contiguous string of E*N*L bits (+ spill words)
    l.envi = o;   o += rw_up4(E * ENVI_W);
    l.misc = o;   o += 4;
    l.total = o;
    return l;
}

// Config policy of the kernel.  DynamicCfg reads every shape from Params at run time (any layout,
// any N/Q, any launch geometry).  StaticCfg bakes the shapes of one registered task and one launch
// geometry in as constants, so the LDS carve-up, the index arithmetic, the divisions and the loop
// trip counts all fold at compile time and the kernel needs a fraction of the scalar registers
// (no kernarg re-loads on the critical path).  A field == 0 means "take it from Params"; StaticCfg with
// N_ == 0 is a "size-static" build: grid shape and geometry folded in, agent count / queue length read
// at run time — one such build covers every registered id of a warehouse size.
struct DynamicCfg {
    static constexpr int kH = 0, kW = 0, kN = 0, kQ = 0, kS = 0, kE = 0, kT = 0, kM = 0;
    static constexpr bool kQrt = false;
    static constexpr int kQcap = 0;
    static constexpr int kNT = -1;   // observation stores: cached or non-temporal by Params::nt_obs, at run time
    static constexpr int kNL = 0, kDirectional = -1;
    static constexpr uint32_t kLayers = 0;
    static constexpr int kPipe = 0;
};
// M_: communication bits (the _MSG kernels).  NL_ / LAYERS_ / DIR_ (IMAGE kernels): a layer list baked in — NL_ layer ids,
// 4 bits each, first layer in the low nibble — and the `image_observation_directional` switch; NL_ == 0: any list, at run time.
// NT_: how the observation stream is stored — 0 cached, 1 with the non-temporal hint (two builds of the per-step kernel, picked
// by rw_create), -1 by Params::nt_obs at run time (the size-static builds: two copies of the expansion pass in one kernel
// cost 2 % at the headline batch and 9 % past the Infinity Cache, measured, so the exact builds do not do that).
// PIPE_: 1 = the chunk-pipelined persistent build of the per-step kernel ("PIPE" in rware_step_kernel): a workgroup walks the
// chunks blockIdx, blockIdx + gridDim, ... through two LDS chunk buffers — wavefront 0 runs the agent phases of chunk c + 1 while
// wavefronts 1 and 2 expand and store the observation of chunk c and wavefront 3 stages chunk c + 2 in.
template <int H_, int W_, int N_, int Q_, int S_, int E_, int T_, int M_ = 0, int NL_ = 0, uint32_t LAYERS_ = 0, int DIR_ = -1, int NT_ = 0, int PIPE_ = 0>
struct StaticCfg {
    static constexpr int kPipe = PIPE_;
    static constexpr int kNT = N_ == 0 ? -1 : NT_;
    // Q_ < 0 (with N_ != 0): an "agent-count-static" build — everything of an exact-shape build except the request-queue
    // length, which is read at run time (any Q <= 2 N_: the easy / normal / hard variants of a task and custom queue sizes
    // share ONE build).  The LDS carve-up reserves the 2 N_ slots, so every offset stays a compile-time constant.
    static constexpr bool kQrt = N_ != 0 && Q_ < 0;
    static constexpr int kQcap = kQrt ? 2 * N_ : Q_;
    static constexpr int kH = H_, kW = W_, kN = N_, kQ = kQrt ? 0 : Q_, kS = S_, kE = E_, kT = T_, kM = M_;
    static constexpr int kNL = NL_, kDirectional = DIR_;
    static constexpr uint32_t kLayers = LAYERS_;
};
constexpr int packed_transposed_layers(uint32_t packed, int n) {  // Params::transposed_layers of a packed list
    int t = 0;
    for (int l = 0; l < n; ++l) {
        const int id = (int)((packed >> (4 * l)) & 15u);
        t |= (id == 3 ? 1 : 0) | (id == 4 ? 2 : 0);  // LAYER_AGENT_DIRECTION, LAYER_AGENT_LOAD
    }
    return t;
}

// Asynchronous flat dword copy HBM -> LDS through the LDS-DMA path.  dwordx4 pieces (1 KiB per wave
// instruction) when the source is 16-byte aligned, dword pieces otherwise; `lds_dst` is 16-byte
// aligned.  Nothing is waited for here.
__device__ __forceinline__ void dma_in(int32_t *lds_dst, const RW_GLOBAL int32_t *src, int n, int tid, int T) {
    const int lane = tid & 63, wave = tid >> 6, nw = T >> 6;
    if ((((uintptr_t)src) & 15u) == 0) {  // wave-uniform
        const int n4 = n >> 2;
        for (int b = wave * 64; b < n4; b += nw * 64)
            if (b + lane < n4) lds_dma_b128(src + 4 * (b + lane), lds_dst + 4 * b);
        const int t0 = n4 << 2;
        if (wave == 0 && lane < n - t0) lds_dma_b32(src + t0 + lane, lds_dst + t0);
    } else {
        for (int b = wave * 64; b < n; b += nw * 64)
            if (b + lane < n) lds_dma_b32(src + b + lane, lds_dst + b);
    }
}

__device__ __forceinline__ void rng_load(Pcg64 &g, const uint64_t *rng, int B, int e) {
    g.state = (((u128)rng[0 * (size_t)B + e]) << 64) | rng[1 * (size_t)B + e];
    g.inc = (((u128)rng[2 * (size_t)B + e]) << 64) | rng[3 * (size_t)B + e];
    g.has_uint32 = (uint32_t)rng[4 * (size_t)B + e];
    g.uinteger = (uint32_t)rng[5 * (size_t)B + e];
}
__device__ __forceinline__ void rng_store(const Pcg64 &g, uint64_t *rng, int B, int e) {
    rng[0 * (size_t)B + e] = (uint64_t)(g.state >> 64);
    rng[1 * (size_t)B + e] = (uint64_t)g.state;
    rng[2 * (size_t)B + e] = (uint64_t)(g.inc >> 64);
    rng[3 * (size_t)B + e] = (uint64_t)g.inc;
    rng[4 * (size_t)B + e] = (uint64_t)g.has_uint32;
    rng[5 * (size_t)B + e] = (uint64_t)g.uinteger;
}

// Packed agent record — the HBM form of one agent (Agent.x/.y/.dir/.carrying_shelf/.has_delivered, rware/warehouse.py:82-93):
//   bits 0..13  cell = y * W + x      (rw_create limits H*W to 10000 cells)
//   bits 14..15 dir                    (rw_direction)
//   bit  16     has_delivered
//   bits 17..30 carried shelf id, 0 == none   (ids 1..S, S <= H*W)
// One dword per agent: a step loads ONE stream and stores ONE stream for the agents' state instead of five each (the state
// write-back is priced per store stream, DESIGN.md ablations).  The five exported int32 arrays are derived views.
RW_HD uint32_t rec_pack(int cell, int d, int deliv, int carry) {
    return (uint32_t)cell | ((uint32_t)d << 14) | ((uint32_t)(deliv ? 1 : 0) << 16) | ((uint32_t)carry << 17);
}
RW_HD int rec_cell(uint32_t r) { return (int)(r & 0x3fffu); }
RW_HD int rec_dir(uint32_t r) { return (int)((r >> 14) & 3u); }
RW_HD int rec_deliv(uint32_t r) { return (int)((r >> 16) & 1u); }
RW_HD int rec_carry(uint32_t r) { return (int)((r >> 17) & 0x3fffu); }

// exported views <-> records (host paths: rw_read / rw_get_buffer of an agent array; after rw_write of one)
template <typename Dummy = void>
__global__ void rware_unpack_agents_kernel(const uint32_t *rec, int32_t *ax, int32_t *ay, int32_t *adir, int32_t *acarry,
                                           int32_t *adeliv, size_t n, int W) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = rec[i];
        const int c = rec_cell(r), y = c / W;
        ax[i] = c - y * W; ay[i] = y; adir[i] = rec_dir(r); acarry[i] = rec_carry(r); adeliv[i] = rec_deliv(r);
    }
}
template <typename Dummy = void>
__global__ void rware_pack_agents_kernel(uint32_t *rec, const int32_t *ax, const int32_t *ay, const int32_t *adir,
                                         const int32_t *acarry, const int32_t *adeliv, size_t n, int W) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        rec[i] = rec_pack((ay[i] * W + ax[i]) & 0x3fff, adir[i] & 3, adeliv[i], acarry[i] & 0x3fff);
}

// exported counter views <-> records (host paths: rw_read / rw_get_buffer of RW_BUF_STEPS / _INACTIVE / _NEED_RESET; after rw_write)
template <typename Dummy = void>
__global__ void rware_unpack_counters_kernel(const int32_t *cnt, int32_t *steps, int32_t *inactive, uint8_t *need_reset, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int32_t x = cnt[2 * i];
        steps[i] = x & 0x7fffffff; inactive[i] = cnt[2 * i + 1]; need_reset[i] = (uint8_t)((uint32_t)x >> 31);
    }
}
template <typename Dummy = void>
__global__ void rware_pack_counters_kernel(int32_t *cnt, const int32_t *steps, const int32_t *inactive, const uint8_t *need_reset, size_t n) {
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        cnt[2 * i] = (steps[i] & 0x7fffffff) | (need_reset[i] ? (int32_t)0x80000000 : 0);
        cnt[2 * i + 1] = inactive[i];
    }
}

// Rebuilds the exported int32 grid [B][2][H][W] (rware/warehouse.py:749-755, _recalc_grid) from the state the kernels keep:
// layer 1 = the shelf shadow, layer 0 = agent ids at the agent coordinates.  Two launches: cells, then agents.
template <typename CellT>
__global__ void rware_grid_cells_kernel(const CellT *shadow, int32_t *grid, int B, int HW) {
    const size_t n = (size_t)B * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i / HW, c = i - e * HW;
        grid[e * 2 * HW + c] = 0;
        grid[e * 2 * HW + HW + c] = (int32_t)shadow[i];
    }
}
template <typename CellT>
__global__ void rware_grid_agents_kernel(const uint32_t *rec, int32_t *grid, int B, int HW, int N) {
    const size_t n = (size_t)B * N;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i / N;
        grid[e * 2 * HW + (size_t)rec_cell(rec[i])] = (int32_t)(i - e * N) + 1;
    }
}

// Rebuilds the shelf shadow from the int32 grid (after a host write of RW_BUF_GRID).
template <typename CellT>
__global__ void rware_shadow_kernel(const int32_t *grid, CellT *shadow, int B, int HW) {
    const size_t n = (size_t)B * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const size_t e = i / HW, c = i - e * HW;
        shadow[i] = (CellT)grid[e * 2 * HW + HW + c];
    }
}

#ifndef RW_AB_OCC8
#define RW_AB_OCC8 0  // A/B hook (profiles/tools/ab.sh): 1 = ask for the register budget of 8 wavefronts per SIMD in every per-step build
#endif
// Which per-step builds ask for the register budget of 8 wavefronts per SIMD (amdgpu_waves_per_eu): 8-env workgroups hold a
// 16384-env batch in ONE round only if 8 of them fit a CU, i.e. 8 wavefronts per SIMD —
//   exact builds, 7 / 8 agents   (one of them — 8 agents, 16 queue slots — came out at 66 VGPRs, 7 per CU: 11.7 instead of ~9.6 us)
//   9 .. 19 agents, agent-count-static and (run-time compiled) exact builds alike   (104 scalar registers otherwise — the
//       run-time queue length and what hangs on it — so 7 per CU; with the budget: 54 .. 62 VGPRs, no scratch.  Round 4,
//       B = 16384: small-10ag 16.0 -> 13.8 us, small-12ag 18.3 -> 15.8.  With the all-gather agent phases the budget cost spills
//       from 14 agents on — 8 .. 44 bytes per lane; the per-cell exchange, kCell, needs no register arrays of N entries.)
template <int R, typename Cfg, bool kRollout>
constexpr bool want_occ8() {
    if (kRollout || RW_AB_OCC8 < 0) return false;
    if (RW_AB_OCC8 > 0) return true;
    if (Cfg::kE != 8) return false;
    // (sensor_range >= 2: the window gather and the expansion hold more rows in registers — the budget would spill on the
    //  common path, e.g. 8 bytes per lane for large-16ag r = 2)
    return (!Cfg::kQrt && (Cfg::kN == 7 || Cfg::kN == 8)) || (R == 1 && Cfg::kN >= 9 && Cfg::kN <= 19);
}
template <int R, typename CellT, typename Cfg, bool kRollout, int kObs = OBS_FLATTENED>
__global__ void __launch_bounds__(256)
#if defined(__HIPCC__)
__attribute__((amdgpu_waves_per_eu(want_occ8<R, Cfg, kRollout>() ? 8 : 1, 8)))
#endif
rware_step_kernel(const Params *__restrict__ cp, RW_LAUNCH_PARAMS) {
    constexpr int WIN = 2 * R + 1, CELLS = WIN * WIN, L0 = 8 + 7 * CELLS, OW0 = (L0 + 31) / 32;
    constexpr bool kImage = (kObs == OBS_IMAGE || kObs == OBS_IMAGE_MSG);
    constexpr bool kMsg = (kObs == OBS_FLATTENED_MSG || kObs == OBS_IMAGE_MSG);  // actions are [Action, message bits...]
    // PIPE — the chunk-pipelined persistent build (OP_STEP launches of exact-shape / agent-count-static FLATTENED kernels).  The
    // phases below stay in program order; the step loop further down becomes a loop over the workgroup's chunks, and because
    // wavefront 0 has no part in the expansion at the bottom of the loop body it falls through to the top and runs the agent
    // phases of the NEXT chunk (other LDS buffer) while wavefronts 1 and 2 are still storing the observation of this one:
    //     wavefront 0     AG(c)            | A | gather(c)                  | B | AG(c + 1) ...
    //     wavefronts 1,2                   | A | gather(c)                  | B | expand + store(c)
    //     wavefront 3     wait for DMA     | A | self bits, write-back(c)   | B | clear scratch, stage chunk c + 2 in (LDS-DMA)
    // Two barriers per chunk (A, B), both LDS-only for everybody but the DMA issuer: nobody waits for observation stores.
    constexpr bool kPipe = Cfg::kPipe != 0;
    static_assert(!kPipe || (Cfg::kE != 0 && Cfg::kN >= 1 && Cfg::kN <= 19 && Cfg::kT == 256 && !kRollout && kObs == OBS_FLATTENED),
                  "the pipelined build: exact-shape or agent-count-static, FLATTENED, 256 threads, per-step");
    static_assert(!kPipe || Cfg::kE <= 64 / (Cfg::kN ? Cfg::kN : 1), "the pipelined build runs the agent phases of a chunk on ONE wavefront");
    const Params &p = *cp;  // constant per engine, device-resident, L2-warm
    const LaunchArgs la{la_actions, la_op, la_n_steps, la_obs, la_rewards, la_terminated, la_reset_mask, la_timeline,
                        la_act_stride, la_obs_stride, la_rew_stride, la_term_stride};
    const int op = la.op & 0xff;
    const bool tl_on = (la.op & OP_FLAG_TIMELINE) != 0;  // the flag is preloaded; la.timeline itself is fetched only when set
    // Start stagger (launches of two or more rounds of workgroups; rw_create decides, bits 16.. of `op`): the workgroups of a launch
    // start together and stay in lock-step — all stage in, all run their agent phases, all store — so the memory system and the
    // SIMDs take turns idling, and the second round inherits the rhythm.  The k-th of the first eight workgroups a CU receives
    // (k = blockIdx / CUs: profiles/tools/placement_probe.hip) waits k * stg ticks of the 100 MHz clock before it begins: their
    // phases no longer coincide and the rounds behind them keep the offsets.  small-4ag x 65536 envs 16.5 -> 15.0 us per step,
    // medium-6ag-hard x 65536 26.4 -> 24 (profiles/r04_stagger_sweep.txt); a single round only loses the delay: not staggered.
    {
        const int stg = (la.op >> 16) & 0xff;
        if (RW_RARE(stg != 0)) {
            const int k = (int)(blockIdx.x >> ((la.op >> 24) & 0xf));
            if (k > 0 && k < 8) {
                const uint64_t t0 = wall_clock64();
                while ((int64_t)(wall_clock64() - t0) < (int64_t)(k * stg)) nap();
            }
        }
    }
    // observation row length: a compile-time constant except with communication bits
    static_assert(kMsg || Cfg::kM == 0, "communication bits need a _MSG observation kind");
    const int M = kMsg ? (Cfg::kM ? Cfg::kM : p.msg_bits) : 0, AM = 1 + M, CW = 7 + M;
    const int L = kMsg ? 8 + CW * CELLS : L0;
    // words of the observation bit string per agent (the image string holds n_layers * CELLS bits per agent)
    const int OW = kImage ? max(OW0, ((Cfg::kNL > 0 ? Cfg::kNL : p.n_layers) * CELLS + 31) / 32) : kMsg ? (L + 31) / 32 : OW0;  // (LDS carve-up: needed first)
    extern __shared__ __align__(16) int32_t smem[];

    int tid = threadIdx.x;
    const int T = Cfg::kT ? Cfg::kT : (int)blockDim.x;
    int lane = tid & 63, wave = tid >> 6;
    const int nw = T >> 6;
    const int E = Cfg::kE ? Cfg::kE : p.envs_per_wg;
    int e0 = blockIdx.x * E;
    const int ne = Cfg::kE ? E : min(E, p.B - e0);  // the static kernels are only launched with B % E == 0
    if (ne <= 0) return;
    const int N = Cfg::kN ? Cfg::kN : p.N, Q = (Cfg::kN && !Cfg::kQrt) ? Cfg::kQ : p.Q;
    const int QL = Cfg::kQrt ? Cfg::kQcap : Q;  // queue slots the LDS carve-up reserves per env
    const int H = Cfg::kH ? Cfg::kH : p.H, W = Cfg::kW ? Cfg::kW : p.W, HW = H * W;
    const int S = Cfg::kS ? Cfg::kS : p.S, SW = (S + 32) / 32, B = p.B;
    const int nea = ne * N;
    // agent phases with cross-lane exchange in registers (see AG); kDirect: own record fetched straight into registers
    // (every registered agent count, rware/__init__.py:16: the chain links of an env as one word — 4 bits x 6 agents in 32 bits,
    //  5 bits x 12 in 64, 6 bits x 19 in 128)
    constexpr bool kRegAG = Cfg::kN >= 1 && Cfg::kN <= 19;
    constexpr bool kDirect = kRegAG && Cfg::kE != 0 && (!kMsg || (Cfg::kM >= 1 && Cfg::kM <= 4));  // (message words: one register each)
#ifndef RW_AB_CELL_AG
#define RW_AB_CELL_AG 1  // A/B hook (profiles/tools/ab.sh): 0 = all-gather exchange for every agent count
#endif
    // 9 .. 19 agents, per-step builds (kCell): the exchange goes through the per-cell agent layer in LDS instead of all-gathers — who
    // stands on my target cell is ONE byte read, who competes for it is a look at its four neighbours, follower depth and the
    // chain walk chase pointers — O(1) per agent where the gathers are O(N) moves + O(N) compares per agent.  These kernels are
    // bound by instruction issue: at 16 agents the gathers were most of the agent phases' ~800 instructions per
    // wavefront.  (Up to 8 agents the gathers are DPP moves or a handful of ds_bpermute and stay.)
    constexpr bool kCell = RW_AB_CELL_AG != 0 && kDirect && !kRollout && Cfg::kN >= 9;
    // ONE scalar batch, first thing in the kernel, for every field of the parameter block that the stage-in and the agent
    // phases read: left to itself hipcc fetches each field where it is first used — three dependent scalar-cache round
    // trips in the prologue (every launch starts with cold caches) and more inside the agent phases, which run on one
    // wavefront and cannot hide them.  keep_sgpr*() pins the values in scalar registers right here.
    // (as_global: these pointers were loaded from memory — hipcc would address them with flat_* instructions, see rware_cdna4.h)
    RW_GLOBAL CellT *const g_shadow = as_global(reinterpret_cast<CellT *>(p.shelf_shadow));
    RW_GLOBAL uint32_t *const q_rec = as_global(p.arec);
    RW_GLOBAL int32_t *const q_queue = as_global(p.queue);
    // (the record as one 64-bit word — low half steps | pending-reset bit, high half inactive: scalar types load and store
    //  through address-space pointers, class types like int2 do not)
    RW_GLOBAL uint64_t *const q_cnt = as_global(reinterpret_cast<uint64_t *>(p.counters));
    struct Cnt { int x, y; };
    auto cnt_load = [&](int ge_) -> Cnt { const uint64_t v = q_cnt[ge_]; return Cnt{(int)(uint32_t)v, (int)(uint32_t)(v >> 32)}; };
    auto cnt_store = [&](int ge_, int x, int y) { q_cnt[ge_] = (uint64_t)(uint32_t)x | ((uint64_t)(uint32_t)y << 32); };
    const RW_GLOBAL uint32_t *const q_hw = as_global(p.highway_bits);
    const int k_reward_type = p.reward_type, k_max_inactivity = p.max_inactivity, k_max_steps = p.max_steps;
    const int k_autoreset = p.autoreset, k_n_goals = p.n_goals, k_normalised = p.normalised, k_nt = p.nt_obs;
    const int k_goal0 = p.goal_cells[0], k_goal1 = p.goal_cells[1];
    keep_sgpr_ptr(g_shadow, q_rec, q_queue, q_cnt, q_hw);
    keep_sgpr(k_reward_type, k_max_inactivity, k_max_steps, k_autoreset, k_n_goals, k_goal0, k_goal1, k_normalised, k_nt);
    if constexpr (Cfg::kQrt) keep_sgpr(Q);  // (the stage-in of the queue needs it)
    // IMAGE kernels: the layer list and its switches belong to the same batch — the image gather used to fetch them where
    // it uses them (one scalar-cache round trip per layer, per goal cell and per switch, inside the phase that stands between
    // the agent phases and the first observation store: 2.5 us against 0.7 us for the FLATTENED gather, r02_timeline_image)
    int k_n_layers = 0, k_directional = 0, k_transposed = 0;
    int k_layer[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    RW_GLOBAL float *q_features = nullptr;
    if constexpr (kImage) {
        q_features = as_global(p.features);
        keep_sgpr_ptr(q_features);
        if constexpr (Cfg::kNL > 0) {  // the layer list is part of the build: every select on a layer id folds
            k_n_layers = Cfg::kNL;
            k_directional = Cfg::kDirectional;
            k_transposed = packed_transposed_layers(Cfg::kLayers, Cfg::kNL);
#pragma unroll
            for (int l = 0; l < 8; ++l) k_layer[l] = (int)((Cfg::kLayers >> (4 * l)) & 15u);
        } else {
            k_n_layers = p.n_layers; k_directional = p.directional; k_transposed = p.transposed_layers;
#pragma unroll
            for (int l = 0; l < 8; ++l) k_layer[l] = p.layers[l];
            keep_sgpr(k_n_layers, k_directional, k_transposed);
            keep_sgpr(k_layer[0], k_layer[1], k_layer[2], k_layer[3], k_layer[4], k_layer[5], k_layer[6], k_layer[7]);
        }
    }
    // wavefront 3 = service wave after the agent phases (see WB); pays off while the observation of a workgroup is
    // small enough that three wavefronts expand it as fast as the stores drain (measured: small-4ag 8.91 -> 8.79 us,
    // fused 4.93 -> 4.56; medium-6ag-hard 8.56 -> 8.40; large-16ag r=2 with 23 K floats per workgroup 43.6 -> 46.1, so not there)
    const bool split = kPipe || (nw == 4 && nea * (kImage ? k_n_layers * CELLS : L) <= 8192);
    const int TW = split ? T - 64 : T;        // threads that gather window rows and expand the observation
    const uint32_t mN = Cfg::kN ? rw_magic18(Cfg::kN) : p.magic_n;
    // optional per-workgroup phase stamps (100 MHz wall clock); one scalar branch per mark when off
#define RW_MARK(k) do { if (RW_RARE(tl_on) && tid == 0) la.timeline[(size_t)blockIdx.x * TL_MARKS + (k)] = wall_clock64(); } while (0)
    // (the pipelined build: stamp k of chunk `it`, taken by the first lane of wavefront `w`)
#define RW_PIPE_MARK(k, w) do { if (kPipe && RW_RARE(tl_on) && it < 4 && tid == 64 * (w)) la.timeline[(size_t)blockIdx.x * TL_MARKS + TL_PIPE + 8 * it + (k)] = wall_clock64(); } while (0)
    // marks INSIDE the agent phases: only in a -DRW_TL_AG_MARKS build (profiles/tools/timeline_probe.py says how) — even
    // switched off each one is a scalar test and a branch on the one wavefront every other wavefront is waiting for
#ifdef RW_TL_AG_MARKS
#define RW_AG_MARK(k, v0, v1) do { if (RW_RARE(tl_on)) { keep_vgpr((v0), (v1)); RW_MARK(k); } } while (0)
#else
#define RW_AG_MARK(k, v0, v1) do { } while (0)
#endif
    RW_MARK(TL_START);

    const LdsLayout lo = make_lds_layout(E, N, QL, HW, SW, OW, (int)sizeof(CellT), AM);
    // every per-chunk array hangs off one base (PIPE: two chunk buffers of lo.total dwords each, re-bound per chunk: bind_lds)
    int32_t *sm = smem;
    CellT *s_gs = reinterpret_cast<CellT *>(smem + lo.gs);
    uint8_t *s_ga = reinterpret_cast<uint8_t *>(smem + lo.ga);
    int32_t *s_ax = smem + lo.ax, *s_ay = smem + lo.ay, *s_dir = smem + lo.dir;
    int32_t *s_carry = smem + lo.carry, *s_deliv = smem + lo.deliv, *s_act = smem + lo.act;
    int32_t *s_tgt = smem + lo.tgt, *s_nxt = smem + lo.nxt, *s_depth = smem + lo.depth, *s_win = smem + lo.win;
    float *s_rew = reinterpret_cast<float *>(smem + lo.rew);
    int32_t *s_mv = smem + lo.mv, *s_msg = smem + lo.msg;
    int32_t *s_xy = smem + lo.tgt;  // per agent x | y << 8 for the observation expansion (aliases the agent phases' s_tgt scratch)
    float *s_fx = reinterpret_cast<float *>(smem + lo.fx), *s_fy = reinterpret_cast<float *>(smem + lo.fy);
    int32_t *s_queue = smem + lo.queue;
    uint32_t *s_req = reinterpret_cast<uint32_t *>(smem + lo.req);
    const uint32_t *s_hw = reinterpret_cast<const uint32_t *>(smem + lo.hw);
    uint32_t *s_obits = reinterpret_cast<uint32_t *>(smem + lo.obits);
    int32_t *s_envi = smem + lo.envi;
    int32_t *s_misc = smem + lo.misc;
    auto bind_lds = [&](int32_t *base) RW_INLINE {
        sm = base;
        s_gs = reinterpret_cast<CellT *>(base + lo.gs);
        s_ga = reinterpret_cast<uint8_t *>(base + lo.ga);
        s_ax = base + lo.ax; s_ay = base + lo.ay; s_dir = base + lo.dir;
        s_carry = base + lo.carry; s_deliv = base + lo.deliv; s_act = base + lo.act;
        s_tgt = base + lo.tgt; s_nxt = base + lo.nxt; s_depth = base + lo.depth; s_win = base + lo.win;
        s_rew = reinterpret_cast<float *>(base + lo.rew);
        s_mv = base + lo.mv; s_msg = base + lo.msg;
        s_xy = base + lo.tgt;
        s_fx = reinterpret_cast<float *>(base + lo.fx); s_fy = reinterpret_cast<float *>(base + lo.fy);
        s_queue = base + lo.queue;
        s_req = reinterpret_cast<uint32_t *>(base + lo.req);
        s_hw = reinterpret_cast<const uint32_t *>(base + lo.hw);
        s_obits = reinterpret_cast<uint32_t *>(base + lo.obits);
        s_envi = base + lo.envi;
        s_misc = base + lo.misc;
    };
    (void)sm; (void)bind_lds;
    auto on_highway = [&](int c) -> bool { return (s_hw[c >> 5] >> (c & 31)) & 1u; };
    auto coordf = [&](int k, int v) -> float {
        if (k_normalised) return (float)((double)v / (double)((k == 0 ? W : H) - 1));  // :636-638
        return (float)v;
    };

    // ---------------------------------------------------------------- P0: stage the env chunk
    auto clear_scratch = [&]() {  // agent layer, depth, obs bit string, request bitmap
        int4 *z = reinterpret_cast<int4 *>(smem + lo.ga);
        const int nz = (lo.zero_end - lo.ga) >> 2;
        for (int i = tid; i < nz; i += T) z[i] = int4{0, 0, 0, 0};
        for (int i = tid; i < nea; i += T) s_depth[i] = 0;
        for (int i = tid; i < nea * OW + 4; i += T) s_obits[i] = 0u;
        for (int i = tid; i < ne * SW; i += T) s_req[i] = 0u;
        if (tid == 0) s_misc[0] = 0;
    };
    // The same clear in whole 16-byte pieces (every sub-array starts on a 16-byte boundary and is padded to one: LdsLayout),
    // through stores hipcc does not order behind the stage-in DMA (lds_zero_b128_blind): kDmaFirst builds issue the DMA first and
    // clear underneath its round trip.  The caller waits for the stores (lds_wait) before the barrier.
    auto clear_scratch_blind = [&]() {
        const int reg[4][2] = {{lo.ga, lo.zero_end}, {lo.depth, lo.win}, {lo.req, lo.envi}, {lo.misc, lo.total}};  // (req and obits are neighbours)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            for (int i = tid; i < ((reg[r][1] - reg[r][0]) >> 2); i += T) lds_zero_b128_blind(smem + reg[r][0] + 4 * i);
    };
    // kDirect: every agent lane fetches ITS OWN record (and its env's flags and counters) from HBM straight into registers,
    // first thing in the kernel: the loads fly beside the clear and the stage-in DMA, are complete at the barrier that
    // drains the DMA, and the agent phases start without an LDS read.  The LDS copies the later phases read (window
    // gather, write-back) are written by the agent lanes together with their results.
    // which envs reset in this launch: OP_RESET — the caller's mask (all-ones when none was given); OP_STEP — the pending-reset
    // bit of the counter record (NEXT_STEP autoreset); OP_OBS — none
    const RW_GLOBAL uint8_t *const q_mask = as_global(la.reset_mask);
    auto flag_of = [&](int cnt_x, int mask_byte) -> int { return op == OP_OBS ? 0 : op == OP_RESET ? mask_byte : (int)((uint32_t)cnt_x >> 31); };
    int r_x = 0, r_y = 0, r_d = 0, r_carry = 0, r_deliv = 0, r_act = ACT_NOOP, r_flag = 0, r_steps = 0, r_inact = 0;
    int r_cx = 0, r_mask = 0;  // (the counter record's first word and the reset-mask byte as loaded: decoded in unpack_own)
    uint32_t r_rec = 0;
    auto unpack_own = [&]() {  // (W is a compile-time constant in the builds that use this)
        r_y = rec_cell(r_rec) / W; r_x = rec_cell(r_rec) - r_y * W;
        r_d = rec_dir(r_rec); r_carry = rec_carry(r_rec); r_deliv = rec_deliv(r_rec);
        r_flag = flag_of(r_cx, r_mask); r_steps = r_cx & 0x7fffffff;
    };
    constexpr int KMW = (kMsg && Cfg::kM) ? Cfg::kM : 1;
    int r_mw[KMW] = {};   // the action's message words (_MSG builds), r_msg: the agent's stored message
    int r_msg = 0;
    if constexpr (kDirect && !kPipe) {
        constexpr int KN = Cfg::kN, KG = 64 / KN;
        static_assert(Cfg::kE <= (Cfg::kT / 64) * KG, "every env of the chunk needs its own agent lanes");
        const int g = lane / KN, a_idx = lane - g * KN, le = wave * KG + g;
        if (g < KG && le < Cfg::kE) {
            const int ge = e0 + le;
            const size_t gi = (size_t)ge * KN + a_idx;
            r_rec = q_rec[gi];  // ONE load per agent: the packed record (unpacked where it is first needed — unpack_own —
                                // so that nothing up here waits for it: the clear and the DMA issue run under its latency)
            if (op == OP_STEP) r_act = la.actions[gi * AM];
            if constexpr (kMsg) {
                r_msg = as_global(p.amsg)[gi];
                if (op == OP_STEP)
#pragma unroll
                    for (int k = 0; k < KMW; ++k) r_mw[k] = la.actions[gi * AM + 1 + k];
            }
            const Cnt c = cnt_load(ge);  // ONE 8-byte load: steps, pending-reset bit, inactive — decoded where the record is
            r_cx = c.x;                  // (unpack_own), so that nothing up here waits for it
            r_inact = c.y;
            if (op == OP_RESET) r_mask = (int)q_mask[ge];
        }
    }
    // P1 of the agent phases as two pieces, so that the exact-shape per-step kernels can run them BEFORE the stage-in barrier
    // (kEarly, below): intent (:825-834, everything that needs only the agent's own record and action) and the occupant
    // of the target cell (a cross-lane exchange of the agents' positions).
    struct Intent { int a, st, tg0, tx0, ty0, occ_w; };
    constexpr int KNX = kRegAG ? Cfg::kN : 1;
    auto intent_of = [&](bool stepping, int a, int x, int y, int d) -> Intent {
        if (RW_RARE(stepping && (unsigned)a > 4u)) atomicOr(p.status, STATUS_INVALID_ACTION);  // Action(a) raises (:814); runs as NOOP
        a = (stepping && (unsigned)a <= 4u) ? a : (int)ACT_NOOP;
        const int fwd = (a == ACT_FORWARD) ? 1 : 0;
        const int dx = fwd & ((d == DIR_RIGHT) ? 1 : 0), dxn = fwd & ((d == DIR_LEFT) ? 1 : 0);
        const int dy = fwd & ((d == DIR_DOWN) ? 1 : 0), dyn = fwd & ((d == DIR_UP) ? 1 : 0);
        const int tx0 = min(max(x + dx - dxn, 0), W - 1);  // clamped at the walls (:105-112)
        const int ty0 = min(max(y + dy - dyn, 0), H - 1);
        return Intent{a, y * W + x, ty0 * W + tx0, tx0, ty0, -1};
    };
    // who stands on my target cell, and is it loaded: every agent announces (cell | loaded << 16 | index << 20)
    auto occupant_of = [&](const Intent &in, int carry, int a_idx, int lane_base) -> int {
        int pkv[KNX];
        env_gather<KNX>(in.st | (carry ? 0x10000 : 0) | (a_idx << 20), lane_base, pkv);
        int occ_w = -1;
#pragma unroll
        for (int k = 0; k < KNX; ++k) occ_w = ((pkv[k] & 0xffff) == in.tg0) ? pkv[k] : occ_w;
        return occ_w;
    };
    // kEarly (exact-shape per-step kernels): the agent wavefront does not take part in the stage-in DMA; its record loads
    // were the first thing it issued, so they are back while the other wavefronts' DMA is still in flight — it computes
    // intent and occupant in that window, before the barrier that everything else of the agent phases has to wait for.
    constexpr bool kEarly = kDirect && !kRollout && Cfg::kT >= 128 && !kPipe;
#ifndef RW_AB_DMA_FIRST
#define RW_AB_DMA_FIRST 0  // A/B hook (profiles/tools/ab.sh)
#endif
    // exact-shape per-step builds: stage-in DMA first, scratch clear underneath it (see clear_scratch_blind)
    constexpr bool kDmaFirst = RW_AB_DMA_FIRST != 0 && Cfg::kE != 0 && Cfg::kN != 0;
    Intent early{ACT_NOOP, 0, 0, 0, 0, -1};
    // builds that stage the agents through LDS: the DMA put the chunk's packed records into the `ax` slot; every thread
    // unpacks its agents in place (reads its own slot before it overwrites it) into the five per-agent LDS arrays
    auto unpack_records = [&]() {
        for (int i = tid; i < nea; i += T) {
            const uint32_t r = (uint32_t)s_ax[i];
            const int c = rec_cell(r), y = c / W;
            s_ax[i] = c - y * W; s_ay[i] = y; s_dir[i] = rec_dir(r); s_carry[i] = rec_carry(r); s_deliv[i] = rec_deliv(r);
        }
    };
    if constexpr (!kPipe) {
    if constexpr (!kDmaFirst) clear_scratch();  // before the DMA: hipcc orders any later LDS write behind an in-flight LDS-DMA (vmcnt)
    RW_MARK(TL_ZEROED);
    if constexpr (Cfg::kE != 0) {
        // Static build: every DMA destination is contiguous in LDS and every source chunk is a whole
        // number of 16-byte pieces, so the chunk is ONE linear stream — thread t moves LDS piece t;
        // its HBM source is picked from the segment table (all pointers fetched in one scalar batch).
        static_assert((Cfg::kE * Cfg::kN) % 4 == 0 && (Cfg::kE * Cfg::kQcap) % 4 == 0 && Cfg::kE % 4 == 0, "chunk not 16-byte granular");
        static_assert(!kMsg || Cfg::kN == 0 || Cfg::kM != 0, "an exact-shape _MSG build needs its communication bits at compile time");
        static_assert((Cfg::kE * Cfg::kH * Cfg::kW * (int)sizeof(CellT)) % 16 == 0, "shelf chunk not 16-byte granular");
        const RW_GLOBAL char *src[12] = {
            as_bytes(g_shadow + (size_t)e0 * HW),
            // the chunk's packed records go to the `ax` slot and are unpacked in place behind the barrier (unpack_records);
            // the ay / dir / carry / deliv slots have no DMA source any more (entries 2..5 are skipped below)
            as_bytes(q_rec + (size_t)e0 * N), nullptr, nullptr, nullptr, nullptr,
            op == OP_STEP ? as_bytes(as_global(la.actions) + (size_t)e0 * N * AM) : as_bytes(q_rec + (size_t)e0 * N),
            as_bytes(q_queue + (size_t)e0 * Q), as_bytes(q_hw),
            as_bytes(q_cnt + e0), nullptr,   // (entry 10: the old second counter array — the record is one stream)
            as_bytes(q_mask + e0)};
        const int seg[13] = {lo.gs, lo.ax, lo.ay, lo.dir, lo.carry, lo.deliv, lo.act, lo.queue, lo.hw,
                             lo.dcnt, lo.dflag, lo.dflag, lo.dma_end};  // dword offsets, all multiples of 4 (entry 10 is empty)
        if constexpr (Cfg::kN != 0) {
            // One DMA instruction moves up to 64 pieces of ONE segment (LDS base + lane * 16), so the source
            // pick is scalar; the (compile-time) list of such instructions is dealt round-robin to the waves.
            // Per wave that is 3-4 instructions of ~3 VALU ops each — the phase is VALU-issue bound otherwise.
            // (kEarly: dealt to wavefronts 1.. only — wavefront 0 must not have a DMA of its own to wait for)
            const int wave_s = uniform(wave) - (kEarly ? 1 : 0), dma_w = nw - (kEarly ? 1 : 0);
            int job = 0;
#pragma unroll
            for (int k = 0; k < 12; ++k) {  // (fully unrolled: the shapes are compile-time constants — keep every `continue` a compile-time one)
                if ((k >= 2 && k <= 5) || k == 10) continue;  // (filled by unpack_records, not by DMA; no source)
                if (kDirect && ((k >= 1 && k <= 6) || k >= 9)) continue;  // agent records, actions, counters, flags: in registers
                const int pieces = (seg[k + 1] - seg[k]) >> 2;
                // (run-time queue length: the slot holds 2 N entries per env, the chunk in HBM is [E][Q] — contiguous, E * Q / 4 pieces)
                const int have = (Cfg::kQrt && k == 7) ? (E * Q) >> 2 : pieces;
                const bool wanted = k != 11 || op == OP_RESET;  // (scalar: the reset mask only matters to OP_RESET)
#pragma unroll
                for (int c = 0; c < pieces; c += 64, ++job)
                    if (job % dma_w == wave_s && c + lane < have && wanted)
                        lds_dma_b128(src[k] + (size_t)(c + lane) * 16, smem + seg[k] + 4 * c);
            }
            if constexpr (kMsg && !kDirect) {  // the agents' stored messages: a 13th array, outside the contiguous block
                const int pieces = (Cfg::kE * Cfg::kN) >> 2;
                for (int c = 0; c < pieces; c += 64, ++job)
                    if (job % dma_w == wave_s && c + lane < pieces)
                        lds_dma_b128(as_bytes(as_global(p.amsg) + (size_t)e0 * N) + (size_t)(c + lane) * 16, smem + lo.msg + 4 * c);
            }
            if constexpr (kDmaFirst) clear_scratch_blind();  // (under the DMA's round trip)
            if constexpr (kEarly) {
                constexpr int KN = Cfg::kN, KG = 64 / KN;
                if (uniform(wave) * KG < Cfg::kE) {  // wave-uniform: a wavefront that runs agent phases
                    const int g = lane / KN, a_idx = lane - g * KN;
                    const bool mine = (g < KG) && (wave * KG + g < Cfg::kE);
                    unpack_own();
                    early = intent_of((op == OP_STEP) && mine && !r_flag, r_act, r_x, r_y, r_d);
                    if constexpr (!kCell) early.occ_w = occupant_of(early, r_carry, a_idx, (g < KG ? g : KG - 1) * KN);
                    keep_vgpr(early.tg0, early.occ_w);  // (materialised here, not sunk below the barrier)
                }
            }
        } else {  // N, Q are run-time values: thread t moves LDS piece t, its source picked per lane
            const int pieces = (lo.dma_end - lo.gs) >> 2;
            for (int b = wave * 64; b < pieces; b += nw * 64) {  // wave-uniform
                const int t = b + lane;
                const RW_GLOBAL char *g = src[0] + (size_t)t * 16;
#pragma unroll
                for (int k = 1; k < 12; ++k)
                    if (t >= ((seg[k] - seg[0]) >> 2)) g = src[k] + (size_t)(t - ((seg[k] - seg[0]) >> 2)) * 16;
                const bool no_src = (t >= ((seg[2] - seg[0]) >> 2) && t < ((seg[6] - seg[0]) >> 2)) ||   // ay .. deliv slots: unpack_records
                                    (t >= ((seg[11] - seg[0]) >> 2) && op != OP_RESET);                  // the reset mask: OP_RESET only
                if (t < pieces && !no_src) lds_dma_b128(g, smem + lo.gs + 4 * b);
            }
        }
        // (rounding pieces at the tail of hw / dcnt / dflag read a few bytes past the logical end of their source:
        //  the bitmap is allocated rounded up to 16 bytes, the records sit in the padded slab, the mask buffer has +64 bytes)
        RW_MARK(TL_DMA_ISSUED);
        RW_MARK(TL_ENV_LOADED);
        if constexpr (kDmaFirst) lds_wait();  // (the blind clear: hipcc does not count those stores)
        dma_wait();       // the stage-in DMA this wavefront issued has landed (explicit: see rware_cdna4.h) ...
        __syncthreads();  // ... and everybody else's: the one full barrier
        if constexpr (!kDirect) {  // (kDirect: the leader lane of each env publishes these from its registers, in AG)
            const uint8_t *s_dflag = reinterpret_cast<const uint8_t *>(smem + lo.dflag);
            for (int e = tid; e < ne; e += T) {
                int32_t *ev = s_envi + e * ENVI_W;
                const int cx = smem[lo.dcnt + 2 * e];
                const int rs = flag_of(cx, op == OP_RESET ? (int)s_dflag[e] : 0);
                ev[ENVI_STEPS] = cx & 0x7fffffff;
                ev[ENVI_INACTIVE] = smem[lo.dcnt + 2 * e + 1];
                ev[ENVI_RESET] = rs;
                ev[ENVI_SKIP] = rs;
                ev[ENVI_DONE] = 0; ev[ENVI_QDIRTY] = 0;
                if (rs) atomicOr(&s_misc[0], 1);
            }
            unpack_records();
            lds_barrier();
        }
    } else {
        dma_in(smem + lo.gs, (const RW_GLOBAL int32_t *)(g_shadow + (size_t)e0 * HW), (ne * HW * (int)sizeof(CellT) + 3) >> 2, tid, T);
        dma_in(s_ax, (const RW_GLOBAL int32_t *)(q_rec + (size_t)e0 * N), nea, tid, T);  // packed records -> the `ax` slot
        dma_in(s_queue, q_queue + (size_t)e0 * Q, ne * Q, tid, T);
        if (op == OP_STEP) dma_in(s_act, as_global(la.actions) + (size_t)e0 * N * AM, nea * AM, tid, T);
        if (kMsg) dma_in(s_msg, as_global(p.amsg) + (size_t)e0 * N, nea, tid, T);
        dma_in(smem + lo.hw, (const RW_GLOBAL int32_t *)q_hw, (HW + 31) / 32, tid, T);
        RW_MARK(TL_DMA_ISSUED);
        lds_barrier();  // orders the s_misc clear above before the flag writes below
        for (int e = tid; e < ne; e += T) {
            int32_t *ev = s_envi + e * ENVI_W;
            const Cnt c = cnt_load(e0 + e);
            const int rs = flag_of(c.x, op == OP_RESET ? (int)q_mask[e0 + e] : 0);
            ev[ENVI_STEPS] = c.x & 0x7fffffff;
            ev[ENVI_INACTIVE] = c.y;
            ev[ENVI_RESET] = rs;
            ev[ENVI_SKIP] = rs;  // an env that resets in this call does not step
            ev[ENVI_DONE] = 0; ev[ENVI_QDIRTY] = 0;
            if (rs) atomicOr(&s_misc[0], 1);
        }
        RW_MARK(TL_ENV_LOADED);
        dma_wait();
        __syncthreads();  // the one full barrier (the DMA has been waited for)
        unpack_records();
        lds_barrier();
    }
    keep_sgpr(k_reward_type, k_max_inactivity, k_max_steps, k_autoreset, k_n_goals, k_goal0, k_goal1, k_normalised, k_nt);
    if constexpr (Cfg::kQrt) keep_sgpr(Q);
    RW_MARK(TL_LOADED);
    }

    // kRollout == false is the single-step kernel (rw_step / rw_reset / rw_refresh_obs): no loop at all.
    const int n_steps = (kRollout && op == OP_STEP) ? la.n_steps : 1;
    // Rollout: when every agent lane owns exactly one (env, agent) for the whole launch, the NEXT step's
    // action is fetched into a register one step ahead, so its HBM latency hides under the current step.
    const bool act_prefetch = kRollout && !kMsg && (ne <= nw * (Cfg::kN ? 64 / (Cfg::kN ? Cfg::kN : 1) : p.groups_per_wave));
    int a_pref = ACT_NOOP;
    // ---------------------------------------------------------------- PIPE: the workgroup's chunks, the two buffers, the stage-in
    // chunk `it` of this workgroup is chunk blockIdx + it * gridDim of the batch and lives in LDS buffer it & 1
    const int pipe_wave = kPipe ? uniform(tid >> 6) : 0;  // (scalar: the roles below are scalar branches)
    const int pipe_chunks = kPipe ? (B / E - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    auto pipe_e0 = [&](int it_) RW_INLINE -> int { return ((int)blockIdx.x + it_ * (int)gridDim.x) * E; };
    // stage the chunk whose first env is ce0 into the buffer at `base`: shelf layer, packed records, actions, queue, highway bitmap,
    // counter records — ONE wavefront issues the whole list (LDS-DMA, 1 KiB per instruction), nothing is waited for here
    auto pipe_stage_in = [&](int ce0, int32_t *base) RW_INLINE {
        if constexpr (kPipe) {
            const RW_GLOBAL char *src[6] = {as_bytes(g_shadow + (size_t)ce0 * HW), as_bytes(q_rec + (size_t)ce0 * N),
                                            as_bytes(as_global(la.actions) + (size_t)ce0 * N * AM), as_bytes(q_queue + (size_t)ce0 * Q),
                                            as_bytes(q_hw), as_bytes(q_cnt + ce0)};
            const int seg[6] = {lo.gs, lo.ax, lo.act, lo.queue, lo.hw, lo.dcnt}, end[6] = {lo.ax, lo.ay, lo.queue, lo.hw, lo.dcnt, lo.dflag};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int pieces = (end[k] - seg[k]) >> 2;
                const int have = (Cfg::kQrt && k == 3) ? (E * Q) >> 2 : pieces;  // (run-time queue length: [E][Q] in HBM, contiguous)
#pragma unroll
                for (int c = 0; c < pieces; c += 64)
                    if (c + lane < have) lds_dma_b128(src[k] + (size_t)(c + lane) * 16, base + seg[k] + 4 * c);
            }
        }
    };
    // zero [lo_, hi_) dwords of a buffer, one wavefront, 16 bytes per lane
    auto pipe_zero = [&](int32_t *base, int lo_, int hi_) RW_INLINE {
        for (int i = lane; i < ((hi_ - lo_) >> 2); i += 64) reinterpret_cast<int4 *>(base + lo_)[i] = int4{0, 0, 0, 0};
    };
    // the agent lanes' own record, action and counter record out of the staged chunk (what the classic flow fetches from HBM
    // at the top of the kernel): the rest of the kDirect path is shared
    auto load_own_lds = [&]() RW_INLINE {
        if constexpr (kPipe) {
            constexpr int KN = Cfg::kN, KG = 64 / KN;
            const int g = lane / KN, a_idx = lane - g * KN;
            if (g < KG && g < Cfg::kE) {
                const int i = g * KN + a_idx;
                r_rec = (uint32_t)s_ax[i];
                if (op == OP_STEP) r_act = s_act[i * AM];
                r_cx = sm[lo.dcnt + 2 * g];
                r_inact = sm[lo.dcnt + 2 * g + 1];
            }
        }
    };
    if constexpr (kPipe) {
        int4 *z = reinterpret_cast<int4 *>(smem);
        for (int i = tid; i < (2 * lo.total) >> 2; i += T) z[i] = int4{0, 0, 0, 0};
        lds_barrier();  // (the clear is another wavefront's: in front of the DMA into the same buffers)
        if (pipe_wave == 3) {
            pipe_stage_in(pipe_e0(0), smem);
            if (pipe_chunks > 1) pipe_stage_in(pipe_e0(1), smem + lo.total);
            dma_wait();
        }
        lds_barrier();
        RW_MARK(TL_LOADED);
    }
    const int n_iter = kPipe ? pipe_chunks : n_steps;
    // fused rollout: one iteration per env step, t; PIPE: one per chunk, `it` (t stays 0: every chunk takes its first and only step)
    for (int t = 0, it = 0; (kPipe ? it : t) < n_iter; kPipe ? ++it : ++t) {
    if constexpr (kPipe) {  // this wavefront moves on to its next chunk (wavefront 0 ahead of the others)
        bind_lds(smem + (it & 1) * lo.total);
        e0 = pipe_e0(it);
    }
    if (kRollout || kPipe) {
        // Re-derive the thread coordinates inside the loop from an opaque copy: otherwise LICM hoists every
        // tid-derived address of the unrolled phases out of the step loop and keeps them live across it
        // (200+ VGPRs, half the occupancy).
        asm volatile("" : "+v"(tid));
        lane = tid & 63;
        wave = tid >> 6;
    }
    const bool worker = !split || wave < 3;
    // who expands and stores the observation: the gatherers — except PIPE, where wavefronts 1 and 2 do (0 runs the next chunk's
    // agent phases meanwhile, 3 stages the chunk after that in)
    const bool x_worker = kPipe ? (pipe_wave == 1 || pipe_wave == 2) : worker;
    const int x_tid = kPipe ? tid - 64 : tid, x_TW = kPipe ? 128 : TW;
    const int32_t *act_t = la.actions + (size_t)t * la.act_stride;
    float *obs_t = la.obs + (size_t)t * la.obs_stride;
    float *rew_t = la.rewards + (size_t)t * la.rew_stride;
    uint8_t *term_t = la.terminated + (size_t)t * la.term_stride;
    if (t > 0) {  // the chunk is already in LDS: recycle the scratch, roll the autoreset flags forward
        lds_barrier();  // the previous step's expansion has finished reading the bit string
        clear_scratch();
        lds_barrier();
        for (int e = tid; e < ne; e += T) {
            int32_t *ev = s_envi + e * ENVI_W;
            const int rs = (k_autoreset == AR_NEXT_STEP) ? ev[ENVI_DONE] : 0;
            ev[ENVI_RESET] = rs;
            ev[ENVI_SKIP] = rs;
            ev[ENVI_DONE] = 0;
            if (rs) atomicOr(&s_misc[0], 1);
        }
        lds_barrier();
    }
    // P5 of one env, run by its leader lane once the moves are applied (:903-942): goals in list order, request replacement
    // with the numpy-exact draw, rewards, the env's counters and termination.  ONE copy, used by both agent-phase
    // implementations below (each kernel instantiation has exactly one call site, so it is inlined there).
    auto goals_and_termination = [&](int e, int ge, int base, int32_t *ev, CellT *gS, uint8_t *gA) {
        int32_t *q = s_queue + e * Q;
        bool delivered = false;
        for (int gi = 0; gi < k_n_goals; ++gi) {  // in list order (:904)
            const int cell = gi == 0 ? k_goal0 : gi == 1 ? k_goal1 : p.goal_cells[gi];
            const int sid = gS[cell];
            if (!sid) continue;
            int slot = -1;  // first queue slot holding sid; all Q entries read in one LDS batch (no early exit)
            for (int k = Q - 1; k >= 0; --k) slot = (q[k] == sid) ? k : slot;
            if (slot < 0) continue;
            delivered = true;
            ev[ENVI_QDIRTY] = 1;
            // candidates = shelves not in the queue, id order; one bounded draw (:915-916)
            Pcg64 rg;
            rng_load(rg, p.rng, B, ge);
            const int idx = (int)pcg_bounded(rg, (uint32_t)(S - Q - 1));
            rng_store(rg, p.rng, B, ge);
            int cand = idx + 1;  // idx-th id (0-based) among ids 1..S that are not queued
            for (;;) {
                int c = 0;
                for (int k = 0; k < Q; ++k) c += (q[k] <= cand) ? 1 : 0;
                const int nc = idx + 1 + c;
                if (nc == cand) break;
                cand = nc;
            }
            q[slot] = cand;
            if (k_reward_type == REW_GLOBAL) {
                for (int k = 0; k < N; ++k) s_rew[base + k] += 1.0f;
            } else {
                const int aid = gA[cell] & 0x7f;
                const int ai = aid > 0 ? aid - 1 : N - 1;  // rewards[-1] when nobody stands there
                if (k_reward_type == REW_INDIVIDUAL) {
                    s_rew[base + ai] += 1.0f;
                } else {
                    s_deliv[base + ai] = 1;
                    s_rew[base + ai] += 0.5f;
                }
            }
        }
        ev[ENVI_INACTIVE] = delivered ? 0 : ev[ENVI_INACTIVE] + 1;
        ev[ENVI_STEPS] += 1;
        const int done = ((k_max_inactivity && ev[ENVI_INACTIVE] >= k_max_inactivity) ||
                          (k_max_steps && ev[ENVI_STEPS] >= k_max_steps)) ? 1 : 0;
        ev[ENVI_DONE] = done;
        if (done && k_autoreset == AR_SAME_STEP) {
            ev[ENVI_RESET] = 1;
            atomicOr(&s_misc[0], 1);
        }
    };
    // ---------------------------------------------------------------- AG: per-agent phases, wave-local
    // Lane -> (env group g, agent a): all N agents of an env sit in one wavefront.  Two implementations:
    //   kRegAG  (exact-shape builds, N <= 6)  the agents of an env exchange intent, chain links, follower depth and
    //           winners through cross-lane moves (env_gather: DPP quad_perm for N = 4 / 2, ds_bpermute otherwise) and
    //           everything else stays in registers; LDS is read twice (own record; the shelf cells the agent looks at)
    //           and written once (the results).  The common step has no LDS round trip after those two reads.
    //           With kDirect the own record does not come from LDS either: the lane fetched it from HBM into registers
    //           at the top of the kernel.
    //   else    the sub-phases exchange through LDS arrays under wave_sync() (any N up to 64, run-time shapes).
    if constexpr (kPipe) { if (pipe_wave == 0) load_own_lds(); }
    if constexpr (kRegAG) {
    constexpr int KN = Cfg::kN, KG = 64 / KN, QS = (Cfg::kQcap + KN - 1) / KN;
    const int KQ = Cfg::kQrt ? Q : Cfg::kQ;  // (a compile-time constant unless the build reads the queue length at run time)
    static_assert(Cfg::kH * Cfg::kW < 0x8000, "cell indices are packed into 16 bits");
    for (int eb = wave * KG; eb < ne; eb += nw * KG) {  // wave-uniform
        const int g = lane / KN, a_idx = lane - g * KN;
        const bool mine = (g < KG) && (eb + g < ne);
        const int lane_base = (g < KG ? g : KG - 1) * KN;  // (the 64 % N idle tail lanes gather from the last group)
        const int e = mine ? eb + g : eb;  // keep every address in range for idle lanes
        const int base = e * KN, i = base + (mine ? a_idx : 0);
        CellT *gS = s_gs + e * HW;
        uint8_t *gA = s_ga + e * HW;
        int32_t *ev = s_envi + e * ENVI_W;
        const int ge = e0 + e;  // global env index
        // ---- own record, env flags and counters: from registers (kDirect, first step of the launch), else LDS read batch 1
        int ev_skip, ev_reset, ev_steps, ev_inact, x, y, d, carry, deliv, a_lds;
        if (kDirect && t == 0) {
            if constexpr (!kEarly) unpack_own();  // (kEarly: done in front of the stage-in barrier)
            ev_skip = ev_reset = r_flag; ev_steps = r_steps; ev_inact = r_inact;
            x = r_x; y = r_y; d = r_d; carry = r_carry; deliv = r_deliv; a_lds = r_act;
        } else {
            ev_skip = ev[ENVI_SKIP]; ev_reset = ev[ENVI_RESET]; ev_steps = ev[ENVI_STEPS]; ev_inact = ev[ENVI_INACTIVE];
            x = s_ax[i]; y = s_ay[i]; d = s_dir[i]; carry = s_carry[i]; deliv = s_deliv[i];
            a_lds = (t == 0) ? s_act[i * AM] : (int)ACT_NOOP;
        }
        const bool stepping = (op == OP_STEP) && mine && !ev_skip;
        int a = ACT_NOOP;
        if (!kEarly && mine) {
            if (stepping) a = (t == 0) ? a_lds : (act_prefetch ? a_pref : act_t[((size_t)ge * KN + a_idx) * AM]);
            if (act_prefetch && t + 1 < n_steps) a_pref = (act_t + la.act_stride)[(size_t)ge * KN + a_idx];
        }
        if constexpr (kMsg && kDirect) {  // agent.message[:] = action[1:] — for every agent, whatever its move does (:812)
            if (mine) {
                int msg = r_msg;  // an env that does not step keeps its stored messages (first step of the launch: registers)
                if (stepping) {
                    msg = 0;
#pragma unroll
                    for (int k = 0; k < KMW; ++k) {
                        const int v = (t == 0) ? r_mw[k] : act_t[((size_t)ge * KN + a_idx) * AM + 1 + k];
                        if (RW_RARE((unsigned)v > 1u)) atomicOr(p.status, STATUS_INVALID_ACTION);  // MultiDiscrete([5, 2, 2, ...])
                        msg |= (v & 1) << k;
                    }
                }
                if (stepping || t == 0) s_msg[i] = msg;
            }
        } else if (kMsg && mine && stepping) {
            int msg = 0;
            for (int k = 0; k < M; ++k) {
                const int v = (t == 0) ? s_act[i * AM + 1 + k] : act_t[((size_t)ge * KN + a_idx) * AM + 1 + k];
                if ((unsigned)v > 1u) atomicOr(p.status, STATUS_INVALID_ACTION);  // MultiDiscrete([5, 2, 2, ...])
                msg |= (v & 1) << k;
            }
            s_msg[i] = msg;
        }
        // ------------------------------------------------------------ P1: intent (:825-846), branch-free
        Intent in = kEarly ? early : intent_of(stepping, a, x, y, d);
        a = in.a;
        const int st = in.st, tg0 = in.tg0, tx0 = in.tx0, ty0 = in.ty0;
        RW_AG_MARK(TL_AG_RECORD, st, a);
        if constexpr (kCell) {  // the start-of-step agent layer (id | 0x80 if loaded; zeroed by the clear): where everybody stands
            if (mine && !ev_reset) gA[st] = (uint8_t)((a_idx + 1) | (carry ? 0x80 : 0));
            wave_sync();
        }
        // ---- LDS read batch 2 (the only one of the common kDirect step): the shelf layer at the target, under the agent and
        // on the first two goal cells (start-of-step values), the highway word of the agent's cell
        const int sh_tg = gS[tg0], shelf_here = gS[st], sh_g0 = gS[k_goal0], sh_g1 = gS[k_goal1];
        const uint32_t hw_word = s_hw[st >> 5];
        int qv[QS > 0 ? QS : 1];  // the queue slots this lane publishes in the requested-shelf bitmap
        const int eq = Cfg::kQrt ? __mul24(e, KQ) : e * KQ;  // row of the env in the LDS queue
#pragma unroll
        for (int q = 0; q < QS; ++q) {
            if (Cfg::kQrt && q * KN >= KQ) { qv[q] = 0; continue; }  // (scalar test: a slot group beyond the run-time queue length)
            qv[q] = s_queue[eq + (Cfg::kQrt ? max(min(a_idx + q * KN, KQ - 1), 0) : min(a_idx + q * KN, KQ - 1))];
        }
        int occ_w;
        if constexpr (kCell) {  // (one byte of the agent layer, read in the same batch)
            const int ag_tg = gA[tg0];
            occ_w = (ag_tg & 0x7f) ? ((((ag_tg & 0x7f) - 1) << 20) | ((ag_tg & 0x80) << 9)) : -1;
        } else {
            occ_w = kEarly ? in.occ_w : occupant_of(in, carry, a_idx, lane_base);  // (issued beside the LDS reads above)
        }
        if (kDirect && t == 0 && mine && a_idx == 0) {  // the env's leader lane publishes the flags and counters the other
            // phases read (from its registers; LDS stores issued while the reads above are in flight)
            ev[ENVI_STEPS] = r_steps; ev[ENVI_INACTIVE] = r_inact; ev[ENVI_RESET] = r_flag; ev[ENVI_SKIP] = r_flag;
            ev[ENVI_DONE] = 0; ev[ENVI_QDIRTY] = 0;
            if (r_flag) atomicOr(&s_misc[0], 1);
        }
        const int occ = occ_w >> 20;  // -1: nobody there
        const int occ_loaded = (occ_w >> 16) & 1 & ~(occ_w >> 31);
        // a standing shelf blocks a loaded agent (:836-846)
        const bool blocked = (carry != 0) & (tg0 != st) & (sh_tg != 0) & (occ_loaded == 0);
        RW_AG_MARK(TL_AG_CELLS, (int)blocked, sh_g0 + sh_g1);
        a = blocked ? (int)ACT_NOOP : a;
        const int tg = blocked ? st : tg0, tx = blocked ? x : tx0, ty = blocked ? y : ty0;
        // successor on the chain: agent index on the target cell, -1 empty, -2 == this agent is stationary
        const int nxt = (tg == st) ? -2 : occ;
        // Chains (an agent stepping onto a cell another agent stands on) are rare; when the wavefront has none, every
        // follower depth is 0 and a mover commits iff it wins its cell.
        const bool chains = wave_any(nxt >= 0);  // wave-uniform
        int depth = 0, lose = 0, commit = 0;
        if constexpr (kCell) {
            // ---- through LDS, O(1) per agent: chain links and contested-cell keys are published, follower depth by walking the
            // links with atomicMax (a chain of movers is short), the winner test looks at the four neighbours of the target
            // cell — whoever else wants that cell stands on one of them —, the chain walk chases pointers.
            if (mine) { s_nxt[i] = nxt; s_tgt[i] = (nxt != -2) ? tg : -1; }  // (s_depth was zeroed by the clear)
            wave_sync();
            if (chains) {
                if (nxt >= 0) {
                    int j = nxt, dd = 1;
                    while (j >= 0 && j != a_idx && dd <= KN && s_nxt[base + j] != -2) {
                        atomicMax(&s_depth[base + j], dd);
                        j = s_nxt[base + j];
                        ++dd;
                    }
                }
                wave_sync();
                depth = s_depth[i];
            }
            // winner of a contested cell: larger follower depth, then the LOWER agent id; only movers compete.  The start-of-step
            // agent layer says who stands on the four neighbours of my target; their published keys say whether they want it.
            {
                const bool okn[4] = {ty > 0, ty < H - 1, tx > 0, tx < W - 1};
                const int nbc[4] = {tg - W, tg + W, tg - 1, tg + 1};
                int kk[4];
                bool val[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {  // (read batch 1: an off-grid neighbour reads my own cell and is masked)
                    const int ida = gA[okn[q] ? nbc[q] : st] & 0x7f;
                    kk[q] = ida - 1;
                    val[q] = okn[q] & (ida != 0) & (kk[q] != a_idx);
                }
                int tk[4], dk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {  // (read batch 2)
                    const int jq = base + (val[q] ? kk[q] : 0);
                    tk[q] = s_tgt[jq];
                    dk[q] = s_depth[jq];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    lose |= (val[q] & (tk[q] == tg) & ((dk[q] > depth) | ((dk[q] == depth) & (kk[q] < a_idx)))) ? 1 : 0;
                lose = (nxt != -2) ? lose : 0;
            }
            // ---- commit (:871-876)
            commit = (nxt == -1) ? (lose ^ 1) : ((nxt == -2) ? 1 : 0);
            if (chains) {  // walk the chain ahead: i -> nxt(i) -> ...
                if (mine) s_win[i] = lose ^ 1;
                wave_sync();
                if (nxt >= 0) {
                    int j = a_idx, hops = 0, ok = 1, cm = 0;
                    for (;;) {
                        ok &= s_win[base + j];
                        const int nj = s_nxt[base + j];
                        ++hops;
                        if (nj == -1) { cm = ok; break; }                  // drains into an empty cell
                        if (nj == a_idx) { cm = (hops >= 3) ? 1 : 0; break; }  // a cycle through me; the 2-swap is refused
                        if (s_nxt[base + nj] == -2) break;                 // blocked by a stationary agent
                        if (hops >= KN) break;                             // feeds a cycle it is not part of
                        j = nj;
                    }
                    commit = cm;
                }
            }
            wave_lds_order();  // every lane's reads of the start-of-step agent layer come before the first lane's update of it (P3)
        } else {
            // ------------------------------------------------------------ P2a: follower depth (longest chain of movers behind me)
            int nxv[KN];
            if (chains) {
                env_gather<KN>(nxt, lane_base, nxv);
                for (int it = 1; it < KN; ++it) {  // relaxation; a chain of movers has at most N - 1 links
                    int dv[KN];
                    env_gather<KN>(depth, lane_base, dv);
                    int nd = 0;
#pragma unroll
                    for (int k = 0; k < KN; ++k) nd = max(nd, (nxv[k] == a_idx) ? dv[k] + 1 : 0);
                    nd = (nxt != -2) ? nd : 0;  // only movers carry a depth
                    const bool changed = nd != depth;
                    depth = nd;
                    if (!wave_any(changed)) break;  // wave-uniform (agents on a cycle never settle: their depth is not used)
                }
            }
            // ------------------------------------------------------------ P2b: winner per contested cell
            // larger follower depth wins, then the LOWER agent id; only movers compete.  One word per agent, target cell above
            // the priority (depth << IB | 2^IB - 1 - index; IB = 4 bits up to 16 agents, 5 beyond): agent k beats me iff the cell
            // fields agree and its word is the larger one.  A stationary agent announces a cell nobody can target (0x1fff00 | index)
            // and so neither beats nor is beaten.
            constexpr int IB = KN <= 16 ? 4 : 5, PW = 2 * IB;  // (a depth is at most N - 1: the same width)
            const uint32_t vme = (nxt != -2) ? ((uint32_t)tg << PW) | ((uint32_t)depth << IB) | (uint32_t)((1 << IB) - 1 - a_idx)
                                             : (0x1fff00u | (uint32_t)a_idx) << PW;
            int kv[KN];
            env_gather<KN>((int)vme, lane_base, kv);
            // (A subtract-and-running-minimum form of this test — one compare at the end — passed the host emulation and failed a
            //  golden trace on the GPU: hipcc folds the DPP move into `v_subrev_u32_dpp` and the result came out with the
            //  operands swapped; profiles/tools/dpp_subrev_probe.hip.  Keep the exchange results in registers of their own.)
#pragma unroll
            for (int k = 0; k < KN; ++k)  // (bitwise on purpose: no short-circuit branches)
                lose |= ((((uint32_t)kv[k] ^ vme) < (1u << PW)) & ((uint32_t)kv[k] > vme)) ? 1 : 0;
            // ------------------------------------------------------------ P2c: commit (:871-876)
            commit = (nxt == -1) ? (lose ^ 1) : ((nxt == -2) ? 1 : 0);
            if (chains) {  // walk the chain ahead: i -> nxt(i) -> ... on the gathered links
                // every agent's (nxt + 2 | win << LB) as one field of a word every lane of the env holds: following a link is
                // a shift and a mask (a register array indexed by a run-time agent index would live in scratch memory).
                // N <= 6: 3 + 1 bits per agent in 32 bits; 7 <= N <= 12: 4 + 1 bits per agent in 64 bits (two OR-reductions);
                // 13 <= N <= 19: 5 + 1 bits per agent in 128 bits — ONE gather of every agent's field (N cross-lane moves), the word
                // assembled in each lane with compile-time shifts (four OR-reductions would be 4 N moves).
                constexpr int LB = KN <= 6 ? 3 : KN <= 12 ? 4 : 5, FW = LB + 1;
                constexpr uint32_t LM = (1u << LB) - 1u;
                using links_t = typename pick_type<KN <= 6, uint32_t, typename pick_type<KN <= 12, uint64_t, u128>::type>::type;
                links_t links;
                {
                    const uint32_t own_field = (uint32_t)((nxt + 2) | ((lose ^ 1) << LB));
                    if constexpr (KN <= 6) {
                        links = (links_t)(uint32_t)env_or<KN>((int)(own_field << (FW * a_idx)), lane_base);
                    } else if constexpr (KN <= 12) {
                        const uint64_t own = (uint64_t)own_field << (FW * a_idx);
                        const uint32_t lo = (uint32_t)env_or<KN>((int)(uint32_t)own, lane_base);
                        const uint32_t hi = (uint32_t)env_or<KN>((int)(uint32_t)(own >> 32), lane_base);
                        links = (links_t)(((uint64_t)hi << 32) | lo);
                    } else {
                        int fv[KN];
                        env_gather<KN>((int)own_field, lane_base, fv);
                        links = 0;
#pragma unroll
                        for (int k = 0; k < KN; ++k) links |= (links_t)(uint32_t)fv[k] << (FW * k);
                    }
                }
                int j = a_idx, hops = 0, ok = 1, cm = 0;
                bool done = nxt < 0;
#pragma unroll
                for (int h = 0; h < KN; ++h) {
                    const uint32_t ent = (uint32_t)(links >> (FW * j)) & ((1u << FW) - 1u);
                    const int nj = (int)(ent & LM) - 2;
                    ok &= (int)(ent >> LB);
                    ++hops;
                    const int nnj = (int)((uint32_t)(links >> (FW * (nj & (KN <= 6 ? 7 : (nj < 0 ? 0 : 31))))) & LM) - 2;  // nxt of the successor (not used when nj < 0)
                    const bool to_empty = nj == -1;                     // drains into an empty cell
                    const bool back = nj == a_idx;                      // a cycle through me; the 2-swap is refused
                    const bool stuck = (nj >= 0) & (nnj == -2);         // blocked by a stationary agent
                    cm = (!done & to_empty) ? ok : cm;
                    cm = (!done & !to_empty & back) ? ((hops >= 3) ? 1 : 0) : cm;
                    done = done | to_empty | back | stuck | (hops >= KN);  // hops == N: feeds a cycle it is not part of
                    j = (nj >= 0) ? nj : j;
                    if (h + 1 < KN && !wave_any(!done)) break;  // wave-uniform: the usual chain is one or two links long
                }
                commit = (nxt >= 0) ? cm : commit;
            }
        }
        // ------------------------------------------------------------ P3: apply (:878-899)
        RW_AG_MARK(TL_AG_WINNERS, commit, lose);
        a = commit ? a : (int)ACT_NOOP;  // a failed mover does nothing (:875)
        const bool moved = (a == ACT_FORWARD) & (tg != st);
        x = moved ? tx : x;
        y = moved ? ty : y;
        // wraplist [UP, RIGHT, DOWN, LEFT] (:119): RIGHT 0->3->1->2->0, LEFT 0->2->1->3->0
        const int right = (0x1023 >> (4 * d)) & 0xF;  // d: 0->3, 1->2, 2->0, 3->1
        const int left = (0x0132 >> (4 * d)) & 0xF;   // d: 0->2, 1->3, 2->1, 3->0
        d = (a == ACT_RIGHT) ? right : ((a == ACT_LEFT) ? left : d);
        // TOGGLE_LOAD (:886-899): pick up the shelf under the agent, or put the carried one down off the highways
        const bool toggle = (a == ACT_TOGGLE);
        const bool drop = toggle & (carry != 0) & (((hw_word >> (st & 31)) & 1u) == 0u);
        const bool pick = toggle & (carry == 0) & (shelf_here != 0);
        const float rew = (drop & (deliv != 0) & (k_reward_type == REW_TWO_STAGE)) ? 0.5f : 0.0f;
        const bool mcar = moved & (carry != 0);  // a loaded mover drags its shelf along the shelf layer
        deliv = drop ? 0 : deliv;
        carry = drop ? 0 : (pick ? shelf_here : carry);
        // ---- results to LDS (stores only; nothing below waits for them on the common path)
        if (kDirect ? mine : stepping) { s_ax[i] = x; s_ay[i] = y; s_dir[i] = d; s_carry[i] = carry; s_deliv[i] = deliv; }
        if (mine) s_rew[i] = rew;  // every agent of the chunk gets its reward slot
        if (mine) s_mv[i] = moved ? (st | (tg << 16)) : -1;  // which two cells changed (write-back hand-off)
        if (mcar) gS[st] = 0;  // incremental _recalc_grid (:749-755): clear phase ...
        if constexpr (kCell) { if (moved) gA[st] = 0; }  // (kCell: the layer holds the start-of-step marks — a mover's goes first)
        wave_lds_order();
        if (mcar) gS[tg] = (CellT)carry;  // ... then set phase, for the whole wavefront in this order
        // the agent layer was zeroed at the start of the step: final position only (id | 0x80 if loaded)
        if (mine && !ev_reset) gA[moved ? tg : st] = (uint8_t)((a_idx + 1) | (carry ? 0x80 : 0));
        // ------------------------------------------------------------ P5: goals, rewards, termination (:903-942)
        // Is there a shelf on a goal cell after the moves?  From registers: a loaded mover that arrived there, or the
        // start-of-step shelf unless a loaded mover took it away.  (More than two goal cells: always take the LDS path.)
        int gflags;
        if constexpr (kCell) {  // (four ballots instead of N cross-lane moves)
            gflags = (env_any<KN>(mcar & (tg == k_goal0), lane_base) ? 1 : 0) | (env_any<KN>(mcar & (tg == k_goal1), lane_base) ? 2 : 0) |
                     (env_any<KN>(mcar & (st == k_goal0), lane_base) ? 4 : 0) | (env_any<KN>(mcar & (st == k_goal1), lane_base) ? 8 : 0);
        } else {
            gflags = env_or<KN>(mcar ? ((tg == k_goal0 ? 1 : 0) | (tg == k_goal1 ? 2 : 0) | (st == k_goal0 ? 4 : 0) |
                                         (st == k_goal1 ? 8 : 0)) : 0, lane_base);
        }
        // bit g of `on_goal`: a shelf stands on goal g after the moves — a loaded mover arrived (gflags bits 0, 1), or the
        // start-of-step shelf is still there (bits 2, 3 say a loaded mover took it away).  Integer arithmetic on purpose.
        const int had = min(sh_g0, 1) | (min(sh_g1, 1) << 1);
        const int on_goal = (gflags | (had & ~(gflags >> 2))) & (k_n_goals > 1 ? 3 : 1);
        const bool goal_hit = (on_goal != 0) | (k_n_goals > 2);
        const bool leader = stepping && a_idx == 0;
        RW_AG_MARK(TL_AG_APPLIED, (int)goal_hit, (int)moved);
        if (RW_RARE(wave_any(leader && goal_hit))) {  // wave-uniform; a delivery may be due: the LDS path
            wave_sync();
            if (leader) {
                goals_and_termination(e, ge, base, ev, gS, gA);
            }
            wave_sync();
#pragma unroll
            for (int q = 0; q < QS; ++q) {  // a request may have been replaced
                if (Cfg::kQrt && q * KN >= KQ) continue;
                qv[q] = s_queue[eq + (Cfg::kQrt ? max(min(a_idx + q * KN, KQ - 1), 0) : min(a_idx + q * KN, KQ - 1))];
            }
        } else if (leader) {  // nothing on a goal: counters and termination from registers
            const int inact = ev_inact + 1, steps = ev_steps + 1;
            const int done = ((k_max_inactivity && inact >= k_max_inactivity) || (k_max_steps && steps >= k_max_steps)) ? 1 : 0;
            ev[ENVI_INACTIVE] = inact;
            ev[ENVI_STEPS] = steps;
            ev[ENVI_DONE] = done;
            if (done && k_autoreset == AR_SAME_STEP) {
                ev[ENVI_RESET] = 1;
                atomicOr(&s_misc[0], 1);
            }
        }
        RW_AG_MARK(TL_AG_GOALS, 0, 0);
        // requested-shelf bitmap of the (post-step) queue.  Envs that reset in this launch are included: RS clears and
        // rebuilds their bitmap.
        if (mine) {
#pragma unroll
            for (int q = 0; q < QS; ++q) {
                if (Cfg::kQrt && q * KN >= KQ) continue;  // (scalar)
                if (a_idx + q * KN < KQ) atomicOr(&s_req[e * SW + (qv[q] >> 5)], 1u << (qv[q] & 31));
            }
        }
    }
    } else {
    const int G = Cfg::kN ? 64 / (Cfg::kN ? Cfg::kN : 1) : p.groups_per_wave;
    for (int eb = wave * G; eb < ne; eb += nw * G) {  // wave-uniform
        const int g = rw_div18(lane, mN), a_idx = lane - g * N;
        const bool mine = (g < G) && (eb + g < ne);
        const int e = mine ? eb + g : eb;  // keep every address in range for idle lanes
        const int base = e * N, i = base + (mine ? a_idx : 0);
        CellT *gS = s_gs + e * HW;
        uint8_t *gA = s_ga + e * HW;
        int32_t *ev = s_envi + e * ENVI_W;
        const int ge = e0 + e;  // global env index
        // ---- R1: own record into registers; rebuild the agent layer (id | 0x80 if loaded)
        // (all LDS reads are issued as one batch: idle lanes read a valid slot and ignore it)
        const int ev_skip = ev[ENVI_SKIP], ev_reset = ev[ENVI_RESET];
        int x = s_ax[i], y = s_ay[i], d = s_dir[i], carry = s_carry[i], deliv = s_deliv[i];
        const int a_lds = (t == 0) ? s_act[i * AM] : (int)ACT_NOOP;
        const bool stepping = (op == OP_STEP) && mine && !ev_skip;
        int a = ACT_NOOP;
        if (mine) {
            if (stepping) a = (t == 0) ? a_lds : (act_prefetch ? a_pref : act_t[((size_t)ge * N + a_idx) * AM]);
            if (kMsg && stepping) {  // agent.message[:] = action[1:] — for every agent, whatever its move does (:812)
                int msg = 0;
                for (int k = 0; k < M; ++k) {
                    const int v = (t == 0) ? s_act[i * AM + 1 + k] : act_t[((size_t)ge * N + a_idx) * AM + 1 + k];
                    if ((unsigned)v > 1u) atomicOr(p.status, STATUS_INVALID_ACTION);  // MultiDiscrete([5, 2, 2, ...])
                    msg |= (v & 1) << k;
                }
                s_msg[i] = msg;
            }
            if (act_prefetch && t + 1 < n_steps) a_pref = (act_t + la.act_stride)[(size_t)ge * N + a_idx];
        }
        const int st = y * W + x;
        if (mine && !ev_reset) gA[st] = (uint8_t)((a_idx + 1) | (carry ? 0x80 : 0));
        wave_sync();
        // ------------------------------------------------------------ P1: intent (:825-846)
        int tg = st, nxt = -2, shelf_here = 0, tx = x, ty = y;
        if (stepping) {
            if ((unsigned)a > 4u) {  // Action(a) raises in the reference (:814); flagged, runs as NOOP
                atomicOr(p.status, STATUS_INVALID_ACTION);
                a = ACT_NOOP;
            }
            // branch-free: the wave holds every action / heading at once, so a 4-way branch costs all 4 arms
            const int fwd = (a == ACT_FORWARD) ? 1 : 0;
            const int dx = fwd & ((d == DIR_RIGHT) ? 1 : 0), dxn = fwd & ((d == DIR_LEFT) ? 1 : 0);
            const int dy = fwd & ((d == DIR_DOWN) ? 1 : 0), dyn = fwd & ((d == DIR_UP) ? 1 : 0);
            tx = min(max(x + dx - dxn, 0), W - 1);  // clamped at the walls (:105-112)
            ty = min(max(y + dy - dyn, 0), H - 1);
            tg = ty * W + tx;
            const int sh_tg = gS[tg], ag_tg = gA[tg];
            shelf_here = gS[st];
            // a standing shelf blocks a loaded agent (:836-846)
            const bool blocked = (carry != 0) & (tg != st) & (sh_tg != 0) & ((ag_tg & 0x80) == 0);
            a = blocked ? (int)ACT_NOOP : a;
            tg = blocked ? st : tg;
            tx = blocked ? x : tx;
            ty = blocked ? y : ty;
            // successor on the chain: agent index on the target cell, -1 empty, -2 == i is stationary
            nxt = (tg == st) ? -2 : ((ag_tg & 0x7f) - 1);
            s_tgt[i] = (nxt == -2) ? -1 : tg;  // contested-cell key: only movers compete
            s_nxt[i] = nxt;
        }
        wave_sync();
        // Chains (an agent stepping onto a cell another agent stands on) are rare; when the wavefront has
        // none, every follower depth is 0, a mover commits iff it wins its cell, and P2a, the depth reads
        // and the s_win exchange (two LDS round trips) drop out.
        const bool chains = wave_any(stepping && nxt >= 0);  // wave-uniform
        // ------------------------------------------------------------ P2a: follower depth
        if (chains) {
            if (stepping && nxt >= 0) {
                int j = nxt, dd = 1;
                while (j >= 0 && j != a_idx && dd <= N && s_nxt[base + j] != -2) {
                    atomicMax(&s_depth[base + j], dd);
                    j = s_nxt[base + j];
                    ++dd;
                }
            }
            wave_sync();
        }
        // ------------------------------------------------------------ P2b: winner per contested cell
        int lose = 0;  // larger follower depth wins, then the LOWER agent id
        if (stepping && nxt != -2) {
            if (chains) {
                const int dme = s_depth[i];
                for (int k = 0; k < N; ++k) {  // (bitwise on purpose: no short-circuit branches)
                    const int tk = s_tgt[base + k], dk = s_depth[base + k];
                    lose |= ((tk == tg) & (k != a_idx) & ((dk > dme) | ((dk == dme) & (k < a_idx)))) ? 1 : 0;
                }
            } else {
                for (int k = 0; k < N; ++k) lose |= ((s_tgt[base + k] == tg) & (k < a_idx)) ? 1 : 0;
            }
        }
        if (chains) {
            if (stepping) s_win[i] = lose ^ 1;
            wave_sync();
        }
        // ------------------------------------------------------------ P2c + P3: commit, apply (:871-899)
        bool moved = false;
        float rew = 0.0f;
        if (stepping) {
            if (nxt == -1) {  // drains into an empty cell: commits iff it won the cell
                if (lose) a = ACT_NOOP;
            } else if (nxt >= 0) {  // walk the chain ahead
                int j = a_idx, hops = 0, ok = 1, commit = 0;
                for (;;) {
                    ok &= s_win[base + j];
                    const int nj = s_nxt[base + j];
                    ++hops;
                    if (nj == -1) { commit = ok; break; }              // drains into an empty cell
                    if (nj == a_idx) { commit = (hops >= 3); break; }  // a cycle through me; 2-swap refused
                    if (s_nxt[base + nj] == -2) break;                 // blocked by a stationary agent
                    if (hops >= N) break;                              // feeds a cycle it is not part of
                    j = nj;
                }
                if (!commit) a = ACT_NOOP;
            }
            moved = (a == ACT_FORWARD) & (tg != st);
            x = moved ? tx : x;
            y = moved ? ty : y;
            if (moved) {
                gA[st] = 0;  // clear phase of the incremental _recalc_grid
                if (carry) gS[st] = 0;
            }
            // wraplist [UP, RIGHT, DOWN, LEFT] (:119): RIGHT 0->3->1->2->0, LEFT 0->2->1->3->0
            const int right = (0x1023 >> (4 * d)) & 0xF;  // d: 0->3, 1->2, 2->0, 3->1
            const int left = (0x0132 >> (4 * d)) & 0xF;   // d: 0->2, 1->3, 2->1, 3->0
            d = (a == ACT_RIGHT) ? right : ((a == ACT_LEFT) ? left : d);
            // TOGGLE_LOAD (:886-899): pick up the shelf under the agent, or put the carried one down off the highways
            const bool toggle = (a == ACT_TOGGLE);
            const bool drop = toggle & (carry != 0) & !on_highway(st);
            const bool pick = toggle & (carry == 0) & (shelf_here != 0);
            rew = (drop & (deliv != 0) & (k_reward_type == REW_TWO_STAGE)) ? 0.5f : 0.0f;
            deliv = drop ? 0 : deliv;
            carry = drop ? 0 : (pick ? shelf_here : carry);
            s_ax[i] = x; s_ay[i] = y; s_dir[i] = d; s_carry[i] = carry; s_deliv[i] = deliv;
        }
        if (mine) s_rew[i] = rew;  // every agent of the chunk gets its reward slot
        wave_sync();
        if (stepping) {  // set phase (also refreshes the loaded flag after a pick-up / drop)
            gA[moved ? tg : st] = (uint8_t)((a_idx + 1) | (carry ? 0x80 : 0));
            if (moved && carry) gS[tg] = (CellT)carry;
        }
        wave_sync();
        // ------------------------------------------------------------ P5: goals, rewards, termination
        if (stepping && a_idx == 0) {
            goals_and_termination(e, ge, base, ev, gS, gA);
        }
        wave_sync();
        // ------------------------------------------------------------ hand-off to the write-back roles
        if (mine) s_mv[i] = (stepping && moved) ? (st | (tg << 16)) : -1;  // which two cells changed
        if (mine && !ev[ENVI_RESET])  // requested-shelf bitmap of the (post-step) queue; RS builds it for reset envs
            for (int k = a_idx; k < Q; k += N) {
                const int sid = s_queue[e * Q + k];
                atomicOr(&s_req[e * SW + (sid >> 5)], 1u << (sid & 31));
            }
    }
    }
    RW_PIPE_MARK(4 - 8, 0);  // (the agent phases of this chunk ran one stage ago: slot 4 of chunk it - 1)
    lds_barrier();  // (PIPE: barrier A)
    RW_MARK(TL_AGENTS);
    RW_PIPE_MARK(0, 0);
    // PIPE: the stage-in wavefront 3 issued one stage ago (chunk it + 1, for the agent phases that start behind barrier B) is waited
    // for HERE, behind barrier A and on wavefront 3 only: it has had a whole stage to land, and the gather does not wait for it
    if constexpr (kPipe) { if (pipe_wave == 3) { dma_wait(); RW_PIPE_MARK(7 - 8, 3); } }

    // ---------------------------------------------------------------- RS: reset flagged envs (:757-802)
    if (RW_RARE(s_misc[0] != 0)) {  // workgroup-uniform; rare
        if constexpr (!kImage) {
            // SAME_STEP autoreset: the observation of the terminating step itself — what Warehouse.step returns together with
            // done = True (rware/warehouse.py:929-946, _make_obs :722-744) — goes to RW_BUF_FINAL_OBS before the env is reset
            // (Gymnasium's info["final_obs"]).  Rare (every max_steps steps), so it is written straight from the definition
            // (:598-674), one thread per float, no bit string: compact code off the common path.
            float *fin = p.final_obs;
            if (op == OP_STEP && k_autoreset == AR_SAME_STEP && fin != nullptr) {
                for (int g = tid; g < nea * L; g += T) {
                    const int i = g / L, k = g - i * L, e = rw_div18(i, mN);
                    if (!s_envi[e * ENVI_W + ENVI_DONE]) continue;  // (only envs this step terminated: ENVI_RESET may also be a mask)
                    const int ax = s_ax[i], ay = s_ay[i];
                    float v;
                    if (k < 8) {  // self part (:643-647)
                        v = k == 0 ? coordf(0, ax) : k == 1 ? coordf(1, ay) : k == 2 ? (s_carry[i] ? 1.0f : 0.0f)
                          : k < 7 ? (s_dir[i] == k - 3 ? 1.0f : 0.0f) : (on_highway(ay * W + ax) ? 1.0f : 0.0f);
                    } else {      // window cell c, row-major, dy outer (:628-629), CW values per cell (:655-673)
                        const int c = (k - 8) / CW, b = (k - 8) - c * CW;
                        const int x = ax + c % WIN - R, y = ay + c / WIN - R;
                        const bool ok = (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H;
                        const int cell = e * HW + (ok ? y * W + x : 0);
                        const int ida = ok ? (s_ga[cell] & 0x7f) : 0, ids = ok ? (int)s_gs[cell] : 0;
                        const int j = e * N + (ida ? ida - 1 : 0);
                        if (b == 0) v = ida ? 1.0f : 0.0f;
                        else if (b < 5) v = (ida ? s_dir[j] : 0) == b - 1 ? 1.0f : 0.0f;  // empty / off-map: [1, 0, 0, 0] (:659)
                        else if (b < 5 + M) v = (ida && ((s_msg[j] >> (b - 5)) & 1)) ? 1.0f : 0.0f;
                        else if (b == 5 + M) v = ids ? 1.0f : 0.0f;
                        else {  // requested: straight from the queue (the bitmap of an env that resets in this launch is not built
                            int rq = 0;  // by every agent-phase implementation — RS rebuilds it after the reset)
                            for (int q = 0; q < Q; ++q) rq |= (s_queue[e * Q + q] == ids) ? 1 : 0;
                            v = (ids && rq) ? 1.0f : 0.0f;
                        }
                    }
                    as_global(fin)[((size_t)e0 * N) * L + g] = v;
                }
                lds_barrier();  // (the reset below overwrites the arrays this read)
            }
        } else {
            // ... and for the IMAGE types (rware/warehouse.py:527-596, 722-744): the terminating step's image — every requested layer of
            // the (rotated) window, one thread per float, from the definition — and, for IMAGE_DICT, its feature vectors
            float *fin = p.final_obs;
            if (op == OP_STEP && k_autoreset == AR_SAME_STEP && fin != nullptr) {
                const int Limg = k_n_layers * CELLS;
                for (int g = tid; g < nea * Limg; g += T) {
                    const int i = g / Limg, rest = g - i * Limg, e = rw_div18(i, mN);
                    if (!s_envi[e * ENVI_W + ENVI_DONE]) continue;  // (only envs this step terminated)
                    const int l = rest / CELLS, rc = rest - l * CELLS, r = rc / WIN, cc = rc - r * WIN;
                    // (the layer id by arithmetic on the packed list or a load from the parameter block: not from the register copy,
                    //  which a run-time index would push into scratch memory)
                    const int layer = Cfg::kNL > 0 ? (int)((Cfg::kLayers >> (4 * l)) & 15u) : p.layers[l];
                    const int ax = s_ax[i], ay = s_ay[i], d = k_directional ? s_dir[i] : (int)DIR_UP;
                    int wr = r, wc = cc;  // (r, cc) indexes the rotated image, (wr, wc) the north-up window (:584-595)
                    if (d == DIR_DOWN) { wr = WIN - 1 - r; wc = WIN - 1 - cc; }
                    else if (d == DIR_LEFT) { wr = WIN - 1 - cc; wc = r; }
                    else if (d == DIR_RIGHT) { wr = cc; wc = WIN - 1 - r; }
                    const int y = ay - R + wr, x = ax - R + wc;
                    const bool ok = (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H;  // outside: np.pad zeros (:573)
                    const int cell = ok ? y * W + x : 0;
                    const int ida = ok ? (s_ga[e * HW + cell] & 0x7f) : 0, ids = ok ? (int)s_gs[e * HW + cell] : 0;
                    const bool tok = ok && x < H && y < W;   // layer[ag.x, ag.y]: the agent standing at (x', y') = (y, x)  (:552, :558)
                    const int gt = tok ? (int)s_ga[e * HW + x * W + y] : 0;
                    float v = 0.0f;
                    if (layer == LAYER_SHELVES) v = ids ? 1.0f : 0.0f;
                    else if (layer == LAYER_REQUESTS) {  // straight from the queue (see the FLATTENED path above)
                        int rq = 0;
                        for (int q = 0; q < Q; ++q) rq |= (s_queue[e * Q + q] == ids) ? 1 : 0;
                        v = (ids && rq) ? 1.0f : 0.0f;
                    } else if (layer == LAYER_AGENTS) v = ida ? 1.0f : 0.0f;
                    else if (layer == LAYER_GOALS) {
                        int gl = 0;
                        for (int q = 0; q < k_n_goals; ++q) gl |= (ok && p.goal_cells[q] == cell) ? 1 : 0;
                        v = gl ? 1.0f : 0.0f;
                    } else if (layer == LAYER_ACCESSIBLE) v = (ok && !ida) ? 1.0f : 0.0f;
                    else if (layer == LAYER_AGENT_DIRECTION) v = (gt & 0x7f) ? (float)(s_dir[e * N + (gt & 0x7f) - 1] + 1) : 0.0f;
                    else if (layer == LAYER_AGENT_LOAD) v = (gt & 0x80) ? 1.0f : 0.0f;
                    as_global(fin)[((size_t)e0 * N) * Limg + g] = v;
                    if (rest == 0 && k_transposed) {  // the reference's IndexError of those two layers (:552, :558), at the terminating step too
                        const bool counted = (k_transposed & 1) || s_carry[i];
                        if (counted && (ax >= H || ay >= W)) atomicOr(p.status, STATUS_IMAGE_INDEX);
                    }
                }
                if (p.final_features)
                    for (int i = tid; i < nea; i += T) {
                        if (!s_envi[rw_div18(i, mN) * ENVI_W + ENVI_DONE]) continue;
                        RW_GLOBAL float *f = as_global(p.final_features) + ((size_t)e0 * N + i) * 6;
                        const int d = s_dir[i];
                        f[0] = d == 0 ? 1.0f : 0.0f; f[1] = d == 1 ? 1.0f : 0.0f; f[2] = d == 2 ? 1.0f : 0.0f; f[3] = d == 3 ? 1.0f : 0.0f;
                        f[4] = on_highway(s_ay[i] * W + s_ax[i]) ? 1.0f : 0.0f;
                        f[5] = s_carry[i] ? 1.0f : 0.0f;
                    }
                lds_barrier();  // (the reset below overwrites the arrays this read)
            }
        }
        for (int c = tid; c < ne * HW; c += T) {
            const int e = c / HW;
            if (!s_envi[e * ENVI_W + ENVI_RESET]) continue;
            s_ga[c] = 0;
            s_gs[c] = (CellT)as_global(p.shelf_init)[c - e * HW];
        }
        __syncthreads();
        for (int e = tid; e < ne; e += T) {
            int32_t *ev = s_envi + e * ENVI_W;
            if (!ev[ENVI_RESET]) continue;
            Pcg64 rg;
            rng_load(rg, p.rng, B, e0 + e);
            int32_t *cells = s_tgt + e * N;  // scratch
            pcg_choice_no_replace(rg, HW, N, cells);  // agent cells (:781-786)
            for (int k = 0; k < N; ++k) {
                const int c = cells[k];
                s_ax[e * N + k] = c % W;
                s_ay[e * N + k] = c / W;
                s_ga[e * HW + c] = (uint8_t)(k + 1);
            }
            for (int k = 0; k < N; ++k) {  // directions (:788)
                s_dir[e * N + k] = (int)pcg_bounded(rg, 3u);
                s_carry[e * N + k] = 0;
                s_deliv[e * N + k] = 0;
            }  // s_rew keeps the terminating step's rewards (SAME_STEP); it is still 0 for envs that did not step
            int32_t *q = s_queue + e * Q;  // request queue (:796-800)
            pcg_choice_no_replace(rg, S, Q, q);
            for (int k = 0; k < SW; ++k) s_req[e * SW + k] = 0u;
            for (int k = 0; k < Q; ++k) {
                q[k] += 1;
                s_req[e * SW + (q[k] >> 5)] |= 1u << (q[k] & 31);
            }
            rng_store(rg, p.rng, B, e0 + e);
            ev[ENVI_STEPS] = 0;
            ev[ENVI_INACTIVE] = 0;
        }
        __syncthreads();
        // write the reset envs back: shelf shadow (the exported int32 grid is derived from it on demand), agent records, queue,
        // counters, self bits
        for (int c = tid; c < ne * HW; c += T) {
            const int e = c / HW;
            if (!s_envi[e * ENVI_W + ENVI_RESET]) continue;
            g_shadow[(size_t)(e0 + e) * HW + (c - e * HW)] = s_gs[c];
        }
        for (int i = tid; i < nea; i += T) {
            const int e = rw_div18(i, mN);
            if (!s_envi[e * ENVI_W + ENVI_RESET]) continue;
            const size_t gi = (size_t)e0 * N + i;
            q_rec[gi] = rec_pack(s_ay[i] * W + s_ax[i], s_dir[i], 0, 0);
            if (kMsg) { s_msg[i] = 0; as_global(p.amsg)[gi] = 0; }  // fresh Agent objects: message = zeros (:89)
            rew_t[gi] = s_rew[i];
            if (!kImage) {
                s_fx[i] = coordf(0, s_ax[i]);
                s_fy[i] = coordf(1, s_ay[i]);
                s_xy[i] = s_ax[i] | (s_ay[i] << 8);
                const uint32_t self = (2u << s_dir[i]) | (on_highway(s_ay[i] * W + s_ax[i]) ? 32u : 0u);
                const int bit = i * L + 2, wd = bit >> 5, sh = bit & 31;
                atomicOr(&s_obits[wd], self << sh);
                if (sh > 26) atomicOr(&s_obits[wd + 1], self >> (32 - sh));
            } else if (p.features) {
                RW_GLOBAL float *f = as_global(p.features) + gi * 6;
                const int d = s_dir[i];
                f[0] = d == 0 ? 1.0f : 0.0f; f[1] = d == 1 ? 1.0f : 0.0f; f[2] = d == 2 ? 1.0f : 0.0f; f[3] = d == 3 ? 1.0f : 0.0f;
                f[4] = on_highway(s_ay[i] * W + s_ax[i]) ? 1.0f : 0.0f;
                f[5] = 0.0f;
            }
        }
        for (int e = tid; e < ne; e += T) {
            const int32_t *ev = s_envi + e * ENVI_W;
            if (!ev[ENVI_RESET]) continue;
            for (int k = 0; k < Q; ++k) q_queue[(size_t)(e0 + e) * Q + k] = s_queue[e * Q + k];
            cnt_store(e0 + e, 0, 0);  // steps 0, nothing pending, inactive 0
            term_t[e0 + e] = (uint8_t)ev[ENVI_DONE];
            as_global(p.truncated)[e0 + e] = 0;
        }
        // fused rollout: a later step's write-back (another wavefront) may store need_reset = 1 for the same env — this
        // path's stores are made visible first (vmcnt drained before the barrier; the path is rare, the wait is free)
        if (kRollout) { dma_wait(); __syncthreads(); } else lds_barrier();
    }
    RW_MARK(TL_RESET);

    // ---------------------------------------------------------------- WB: state write-back, one role per wavefront
    // (envs flagged for reset were written by RS).  Where it runs is a measured choice.  Without the split below
    // (fewer than 4 wavefronts, or a large observation chunk), one role per wavefront:
    //   single step    before the observation: its small stores then drain underneath P7; issued after the
    //                  18 MB observation stream they queue behind it and hold every wavefront ~0.8 us longer
    //   fused rollout  after the observation stores: the next step's compute hides them, and the stream
    //                  starts 0.4 us earlier (5.62 -> 5.44 us per step)
    // With 4 wavefronts the workgroup splits after the agent phases ("split"): wavefront 3 is the service wave — it
    // writes the self bits while wavefronts 0..2 gather the window rows, and after the barrier it does ALL the state
    // write-back while wavefronts 0..2 expand and store the observation.  The observation stream — what the step
    // ends with — then starts one write-back earlier, and the small state stores go out beside its head instead of
    // behind its tail.
    auto write_back = [&](int first_role, int role_step) {
    for (int role = first_role; role < 3; role += role_step) {  // wave-uniform
        if (role == 0) {  // per-env counters and flags, request queue
            if (op == OP_STEP)
                for (int e = lane; e < ne; e += 64) {
                    const int32_t *ev = s_envi + e * ENVI_W;
                    if (ev[ENVI_RESET]) continue;
                    const int ge = e0 + e;
                    // (fused rollout: only the launch's last step stores the pending-reset bit — the reset at the top of the
                    //  following step consumes it from LDS)
                    const int pend = (ev[ENVI_DONE] && k_autoreset == AR_NEXT_STEP && (!kRollout || t + 1 == n_steps)) ? (int)0x80000000 : 0;
                    cnt_store(ge, ev[ENVI_STEPS] | pend, ev[ENVI_INACTIVE]);  // ONE 8-byte store: the env's counter record
                    term_t[ge] = (uint8_t)ev[ENVI_DONE];
                    // Only what changed: RW_BUF_TRUNCATED is zero for the engine's lifetime (the reference never truncates,
                    // :942); the queue changes only on a delivery.  Every store stream a step does not issue is ~0.1 us of it
                    // (DESIGN.md ablations).
                    if (ev[ENVI_QDIRTY])
                        for (int k = 0; k < Q; ++k) q_queue[(size_t)ge * Q + k] = s_queue[e * Q + k];
                }
        } else if (role == 1) {  // agent records and rewards: the chunk is contiguous in both [B][N] arrays
            if (op == OP_STEP)
                for (int i = lane; i < nea; i += 64) {
                    if (s_envi[rw_div18(i, mN) * ENVI_W + ENVI_RESET]) continue;
                    const size_t gi = (size_t)e0 * N + i;
                    q_rec[gi] = rec_pack(s_ay[i] * W + s_ax[i], s_dir[i], s_deliv[i], s_carry[i]);  // one store stream, not five
                    rew_t[gi] = s_rew[i];
                    if (kMsg) as_global(p.amsg)[gi] = s_msg[i];
                }
        } else if (role == 2) {  // patch the shelf shadow at the two cells a LOADED mover changed
            // (The exported int32 grid, RW_BUF_GRID, is NOT patched here any more: it is rebuilt from the shadow and the agent
            //  coordinates when somebody asks for it — rw_refresh_grid.  Its scattered 4-byte patches were partial-line
            //  writes; once a batch outgrows the Infinity Cache each of them is a read-modify-write in HBM: 15 % of the
            //  step at B = 262144, measured by ablation.)
            if (op == OP_STEP)
                for (int i = lane; i < nea; i += 64) {
                    const int mv = s_mv[i], carry = s_carry[i];
                    if (mv < 0 || !carry) continue;
                    const int e = rw_div18(i, mN);
                    if (s_envi[e * ENVI_W + ENVI_RESET]) continue;
                    const int st = mv & 0xffff, tg = mv >> 16;
                    const size_t ge = (size_t)(e0 + e);
                    // the cell it left: cleared unless a loaded follower stepped onto it — the follower then writes that
                    // cell itself (as its `tg`), so every shadow cell has exactly one writer
                    if (s_gs[e * HW + st] == 0) g_shadow[ge * HW + st] = 0;
                    g_shadow[ge * HW + tg] = (CellT)carry;
                }
        }
    }
    };
    if (!split && !kRollout) write_back(wave, nw);

    // ---------------------------------------------------------------- OS: self part of the observation
    // Runs on the LAST role slot (wavefront 3 of 4), side by side with the window rows below, which fill
    // wavefronts 0..2 first.
    for (int role = wave; role < 4; role += nw) {  // wave-uniform
        if (role != 3) continue;
        if (!kImage) {  // self part of the observation, k = 2..7 (:643-647), and the float coordinates k = 0,1
            for (int i = lane; i < nea; i += 64) {
                if (s_envi[rw_div18(i, mN) * ENVI_W + ENVI_RESET]) continue;
                const int x = s_ax[i], y = s_ay[i];
                s_fx[i] = coordf(0, x);
                s_fy[i] = coordf(1, y);
                s_xy[i] = x | (y << 8);  // (aliases the s_tgt scratch of the agent phases, free by now)
                const uint32_t self = (s_carry[i] ? 1u : 0u) | (2u << s_dir[i]) | (on_highway(y * W + x) ? 32u : 0u);
                const int bit = i * L + 2, wd = bit >> 5, sh = bit & 31;
                atomicOr(&s_obits[wd], self << sh);
                if (sh > 26) atomicOr(&s_obits[wd + 1], self >> (32 - sh));
            }
        } else {
            if (q_features)  // IMAGE_DICT feature vector: one-hot direction, on_highway, carrying (:730-738)
                for (int i = lane; i < nea; i += 64) {
                    if (s_envi[rw_div18(i, mN) * ENVI_W + ENVI_RESET]) continue;
                    RW_GLOBAL float *f = q_features + ((size_t)e0 * N + i) * 6;
                    const int d = s_dir[i];
                    f[0] = d == 0 ? 1.0f : 0.0f; f[1] = d == 1 ? 1.0f : 0.0f; f[2] = d == 2 ? 1.0f : 0.0f; f[3] = d == 3 ? 1.0f : 0.0f;
                    f[4] = on_highway(s_ay[i] * W + s_ax[i]) ? 1.0f : 0.0f;
                    f[5] = s_carry[i] ? 1.0f : 0.0f;
                }
            if (k_transposed)  // layer[ag.x, ag.y] on an (H, W) array (:552, :558): IndexError when out of bounds
                for (int i = lane; i < nea; i += 64) {  // (envs reset in this launch included: nobody is loaded there)
                    const bool loaded = s_carry[i] && !s_envi[rw_div18(i, mN) * ENVI_W + ENVI_RESET];
                    const bool counted = (k_transposed & 1) || loaded;
                    if (counted && (s_ax[i] >= H || s_ay[i] >= W)) atomicOr(p.status, STATUS_IMAGE_INDEX);
                }
        }
    }
    // ---------------------------------------------------------------- P7: observation bits (:598-674)
    // One contiguous bit string per workgroup: bit (i*L + k) == obs[agent i][k] for k >= 2; the two
    // coordinate slots k = 0,1 stay 0 here and are filled in as floats during expansion.
    if constexpr (kMsg && !kImage) {
        // with communication bits a cell code is 7 + M bits wide: [has_agent, dir x4, message x M, has_shelf,
        // requested] (:655-673); gathered per (agent, cell)
        if (worker)
        for (int w = tid; w < nea * CELLS; w += TW) {
            const int i = w / CELLS, cidx = w - i * CELLS;
            const int e = rw_div18(i, mN);
            const int ax = s_ax[i], ay = s_ay[i];
            const int x = ax + cidx % WIN - R, y = ay + cidx / WIN - R;
            // (two unconditional LDS read batches; an off-map cell reads the agent's own cell and is masked)
            const bool ok = (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H;
            const int c = e * HW + (ok ? y * W + x : ay * W + ax);
            const int ida = ok ? (s_ga[c] & 0x7f) : 0, ids = ok ? (int)s_gs[c] : 0;
            const int j = e * N + (ida ? ida - 1 : 0);
            const int dj = s_dir[j], mj = s_msg[j];
            const uint32_t rq = s_req[e * SW + (ids >> 5)];
            // empty / off-map: direction one-hot [1,0,0,0], message skipped (zeros)
            uint32_t code = ida ? (1u | (2u << dj) | ((uint32_t)mj << 5)) : 2u;
            code |= ids ? ((1u << (5 + M)) | (((rq >> (ids & 31)) & 1u) << (6 + M))) : 0u;
            const int bit = i * L + 8 + CW * cidx;
            const int wd = bit >> 5, sh = bit & 31;
            atomicOr(&s_obits[wd], code << sh);
            if (sh + CW > 32) atomicOr(&s_obits[wd + 1], code >> (32 - sh));
        }
    } else if constexpr (kObs == OBS_FLATTENED) {
    // one thread per (agent, window row): the agent's position is read once, the row's WIN cells are
    // gathered with independent LDS reads, and the row's 7*WIN bits go out in one or two LDS atomics
    if (worker)
    for (int w = tid; w < nea * WIN; w += TW) {
        const int i = w / WIN, row = w - i * WIN;
        const int e = rw_div18(i, mN);
        const int ax = s_ax[i], ay = s_ay[i], y = ay + row - R;
        const bool row_ok = (unsigned)y < (unsigned)H;
        const int rowbase = e * HW + y * W, own = e * HW + ay * W + ax;
        // Two LDS read batches, no read inside a branch (hipcc waits for each predicated read on its own, which
        // costs a full LDS round trip per cell): out-of-map cells read the agent's own cell and are masked after.
        int ida[WIN], ids[WIN];
        bool ok[WIN];
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            const int x = ax + k - R;
            ok[k] = row_ok && (unsigned)x < (unsigned)W;
            const int c = ok[k] ? rowbase + x : own;
            ida[k] = s_ga[c] & 0x7f;
            ids[k] = (int)s_gs[c];
        }
        int dirv[WIN];
        uint32_t reqw[WIN];
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            ida[k] = ok[k] ? ida[k] : 0;
            ids[k] = ok[k] ? ids[k] : 0;
            dirv[k] = s_dir[e * N + (ida[k] ? ida[k] - 1 : 0)];
            reqw[k] = s_req[e * SW + (ids[k] >> 5)];
        }
        uint64_t bits = 0;  // 7 * WIN <= 77 bits for R <= 5: R <= 4 fits 64; R == 5 handled by the split below
        uint32_t hi = 0;    // bits 64.. of the row (only R == 5)
#pragma unroll
        for (int k = 0; k < WIN; ++k) {
            // empty / off-map cell: has_agent 0, direction one-hot [1,0,0,0] (:659)
            uint32_t code = ida[k] ? (1u | (2u << dirv[k])) : 2u;
            code |= ids[k] ? (32u | (((reqw[k] >> (ids[k] & 31)) & 1u) << 6)) : 0u;
            if (7 * k < 64) bits |= (uint64_t)code << (7 * k);
            if (7 * k + 7 > 64) hi |= (7 * k >= 64) ? (code << (7 * k - 64)) : (code >> (64 - 7 * k));
        }
        const int bit = i * L + 8 + 7 * WIN * row;
        const int wd = bit >> 5, sh = bit & 31;
        // the row occupies bits [sh, sh + 7*WIN) of the window starting at word wd
        const uint32_t lo32 = (uint32_t)bits, mid32 = (uint32_t)(bits >> 32);
        atomicOr(&s_obits[wd], lo32 << sh);
        if (sh + 7 * WIN > 32) {
            const uint32_t w1 = (sh ? (lo32 >> (32 - sh)) : 0u) | (mid32 << sh);
            atomicOr(&s_obits[wd + 1], w1);
        }
        if (sh + 7 * WIN > 64) {
            const uint32_t w2 = (sh ? (mid32 >> (32 - sh)) : 0u) | (hi << sh);
            atomicOr(&s_obits[wd + 2], w2);
        }
        if (7 * WIN > 64 && sh + 7 * WIN > 96) {
            const uint32_t w3 = sh ? (hi >> (32 - sh)) : 0u;
            atomicOr(&s_obits[wd + 3], w3);
        }
    }
    } else {
        // IMAGE observation: per agent n_layers x WIN x WIN binary values, optionally rotated into the
        // agent's heading (np.rot90 of the north-up window, :584-595).  Same contiguous bit string; one
        // thread per (agent, layer, image row).
        // thread per (agent, image row): the row's WIN cells are read once and give one WIN-bit mask per
        // property; every requested layer is then one of those masks.
        const int Limg = k_n_layers * CELLS;
        if (worker)
        for (int w = tid; w < nea * WIN; w += TW) {
            const int i = w / WIN, r = w - i * WIN;
            const int e = rw_div18(i, mN);
            const int ax = s_ax[i], ay = s_ay[i], d = k_directional ? s_dir[i] : DIR_UP;
            uint32_t m_shelf = 0, m_req = 0, m_agent = 0, m_goal = 0, m_map = 0;
            uint32_t m_tagent = 0, m_tload = 0;  // the transposed layers: an agent / a loaded agent with (x, y) == (row, col)
            // LDS reads in unconditional batches (an off-map cell reads the agent's own cell and is masked): a
            // predicated read costs a full LDS round trip of its own
            const int own = e * HW + ay * W + ax;
            int cellv[WIN], gav[WIN], gsv[WIN], gtv[WIN];
            bool okv[WIN];
#pragma unroll
            for (int cc = 0; cc < WIN; ++cc) {
                int wr = r, wc = cc;  // (r, cc) indexes the rotated image, (wr, wc) the north-up window
                if (d == DIR_DOWN) { wr = WIN - 1 - r; wc = WIN - 1 - cc; }   // k = 2
                else if (d == DIR_LEFT) { wr = WIN - 1 - cc; wc = r; }        // k = 3
                else if (d == DIR_RIGHT) { wr = cc; wc = WIN - 1 - r; }       // k = 1
                const int y = ay - R + wr, x = ax - R + wc;
                okv[cc] = (unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H;  // outside: np.pad zeros (:573)
                cellv[cc] = y * W + x;
                const int c = okv[cc] ? e * HW + cellv[cc] : own;
                gav[cc] = s_ga[c];
                gsv[cc] = (int)s_gs[c];
                gtv[cc] = 0;
                if (k_transposed) {  // layer[ag.x, ag.y]: the agent standing at (x', y') = (y, x)
                    const bool tok = okv[cc] && x < H && y < W;
                    gtv[cc] = tok ? (int)s_ga[tok ? e * HW + x * W + y : own] : 0;
                }
            }
#pragma unroll
            for (int cc = 0; cc < WIN; ++cc) {
                const int ida = okv[cc] ? (gav[cc] & 0x7f) : 0, ids = okv[cc] ? gsv[cc] : 0;
                const uint32_t rq = s_req[e * SW + (ids >> 5)];
                m_map |= (okv[cc] ? 1u : 0u) << cc;
                m_agent |= (ida ? 1u : 0u) << cc;
                m_shelf |= (ids ? 1u : 0u) << cc;
                m_req |= (ids ? ((rq >> (ids & 31)) & 1u) : 0u) << cc;
                m_goal |= ((okv[cc] && ((k_n_goals > 0 && k_goal0 == cellv[cc]) || (k_n_goals > 1 && k_goal1 == cellv[cc]))) ? 1u : 0u) << cc;
                for (int g = 2; g < k_n_goals; ++g)  // (more than two goal cells: custom layouts)
                    if (okv[cc] && p.goal_cells[g] == cellv[cc]) m_goal |= 1u << cc;
                m_tagent |= ((gtv[cc] & 0x7f) ? 1u : 0u) << cc;
                m_tload |= ((gtv[cc] & 0x80) ? 1u : 0u) << cc;
            }
#pragma unroll
            for (int l = 0; l < 8; ++l) {  // (unrolled over the register copy of the layer list)
                if (l >= k_n_layers) break;
                const int layer = k_layer[l];
                // (AGENT_DIRECTION holds dir + 1 in 1..4: its bit marks the cell, the value is patched in after the
                //  expansion, see below)
                const uint32_t bits = layer == LAYER_SHELVES ? m_shelf : layer == LAYER_REQUESTS ? m_req
                                    : layer == LAYER_AGENTS ? m_agent : layer == LAYER_GOALS ? m_goal
                                    : layer == LAYER_AGENT_DIRECTION ? m_tagent : layer == LAYER_AGENT_LOAD ? m_tload
                                    : (m_map & ~m_agent);  // LAYER_ACCESSIBLE
                const int bit = i * Limg + (l * WIN + r) * WIN;
                const int wd = bit >> 5, sh = bit & 31;
                atomicOr(&s_obits[wd], bits << sh);
                if (sh + WIN > 32) atomicOr(&s_obits[wd + 1], bits >> (32 - sh));
            }
        }
    }
    if constexpr (kPipe) { if (pipe_wave == 3) write_back(0, 1); }  // (before barrier B: behind it the buffer's staging slots are refilled)
    RW_PIPE_MARK(1, 0);
    RW_PIPE_MARK(2, 3);
    lds_barrier();
    RW_MARK(TL_OBS_BITS);
    RW_PIPE_MARK(3, 0);
    if constexpr (!kPipe) {
        if (split && wave == 3) write_back(0, 1);  // all three roles, beside the head of the observation stream
    } else if (pipe_wave == 3) {
        // the service wavefront, beside the expansion of this chunk and the agent phases of the next: the scratch of THIS buffer
        // (its readers — gather, write-back — ran in front of the barrier) and the bit string of the OTHER one (expanded one stage
        // ago) are zeroed for the chunks that come next, and chunk it + 2 is staged into this buffer's slots
        pipe_zero(sm, lo.ga, lo.zero_end);
        pipe_zero(sm, lo.depth, lo.win);
        pipe_zero(sm, lo.req, lo.obits);
        pipe_zero(sm, lo.misc, lo.total);
        pipe_zero(smem + ((it + 1) & 1) * lo.total, lo.obits, lo.envi);
        if (it + 2 < n_iter) pipe_stage_in(pipe_e0(it + 2), sm);
        RW_PIPE_MARK(6, 3);
    }

    // ---------------------------------------------------------------- ST: obs, float4 #q == nibble #q
    if constexpr (!kImage) {
        const int nf = nea * L;
        const int nf4 = nf >> 2;
        float *out = obs_t + (size_t)e0 * N * L;  // 16-byte aligned: e0 is a multiple of 4
        float4 *out4 = reinterpret_cast<float4 *>(out);
        auto spread = [&](uint32_t nib) -> float4 {  // 4 bits -> 4 floats
            // one multiply spreads the bits into 4 bytes (0 or 1 each); hidden from the optimiser so that each
            // byte converts with ONE v_cvt_f32_ubyteN instead of a shift/and/convert chain
            const uint32_t b = opaque((nib * 0x00204081u) & 0x01010101u);
            float4 v;
            v.x = (float)(b & 0xFFu);
            v.y = (float)((b >> 8) & 0xFFu);
            v.z = (float)((b >> 16) & 0xFFu);
            v.w = (float)(b >> 24);
            return v;
        };
        // 16-byte store at (uniform) out + a per-lane byte offset the optimiser cannot take apart: it then keeps
        // the scalar-base form of the store instead of rebuilding a 64-bit per-lane address for every pass.
        // (Not inline asm: hipcc must see the store to respect the write-data hazard of 128-bit stores.)
        // `nt` (a std::bool_constant tag): store with the non-temporal hint.  Whole 128-byte lines written exactly once are a
        // pure stream; with the hint they no longer displace the state the next launch reads back, and at the headline batch
        // the step goes 7.13 -> 6.17 us, past the Infinity Cache 70.2 -> 60.2 us (round 3; round 1 measured the opposite on
        // the old two-pass expansion, whose second pass re-touched lines).  Per engine, by Params::nt_obs (rw_create's rule).
        auto store4 = [&](auto nt, uint32_t byte_off, float4 v) {
            float4 *dst = reinterpret_cast<float4 *>(reinterpret_cast<char *>(out) + (size_t)opaque(byte_off));
            if constexpr (decltype(nt)::value) store_f4_nt(dst, v); else *dst = v;
        };
        auto expand = [&](int q4) -> float4 {  // float4 #q4 of the chunk == nibble #q4 of the bit string
            return spread((s_obits[q4 >> 3] >> ((q4 & 7) << 2)) & 0xFu);
        };
        // ONE pass over the chunk's float4s when the coordinates are plain cell indices (not normalised): thread t takes
        // float4 t, t + T, ...: its nibble position inside the word (t & 7) and its word column (t >> 3) never change
        // (T % 8 == 0), and m = (4 q + 3) mod L — the float4 holds a coordinate slot iff m < 5 — and the agent index
        // (4 q + 3) div L advance by constants.  The four bits become four BYTES (0 / 1) that convert with one
        // v_cvt_f32_ubyteN each; a coordinate is a small integer and converts the same way, so the agent's (x | y << 8)
        // word is simply OR-ed into the byte lanes of its slots (which are 0 in the bit string).  Every float4 is written
        // exactly once and in order: whole 128-byte lines, no second scattered pass (7.84 -> 7.5 us per step at the
        // headline batch, -7 % at the cache-exceeding batches).
        // (not in the fused rollout: its steps are bound by instruction issue, not by the store stream, and the single pass
        //  costs ~10 more VALU operations per float4: 4.16 -> 4.78 us per step there)
        // (the coordinates travel as bytes: layouts wider or taller than 256 cells take the two-pass form as well — a
        //  compile-time fact in the exact-shape and size-static builds)
        const bool xy_bytes = !kRollout && !k_normalised && W <= 256 && H <= 256;  // workgroup-uniform
        auto single_pass = [&](auto nt) {
            // (the thread index through an opaque copy: otherwise the address arithmetic of BOTH copies of the pass is hoisted in
            //  front of the branch that picks one — large-16ag r=2: 102 VGPRs instead of 60, 4 workgroups per CU instead of 7)
            // (only in the builds that hold both copies: with one copy the hoisting is wanted — the address arithmetic then runs
            //  while the workgroup waits at the bit-string barrier)
            int tq = x_tid;
            if constexpr (Cfg::kNT < 0) asm volatile("" : "+v"(tq));
            const int shift = (tq & 7) << 2, words_per_pass = x_TW >> 3;  // (x_TW % 8 == 0)
            const uint32_t *wp = s_obits + (tq >> 3);
            const int dm = (4 * x_TW) % L, di = (4 * x_TW) / L;
            int m = (4 * tq + 3) % L, ai = (4 * tq + 3) / L;
            const int passes = (nf4 + x_TW - 1) / x_TW;  // a compile-time constant in the specialised builds (full unroll)
            // in groups of 8 passes: first the LDS reads of all 8 in one unconditional batch (a read past the string still
            // lands inside the workgroup's LDS; the agent index is clamped), then the 8 expansions
            for (int k0 = 0; k0 < passes; k0 += 8) {
                uint32_t wv[8], xyv[8];
                int mv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    mv[j] = m;
                    wv[j] = (k0 + j < passes) ? wp[(k0 + j) * words_per_pass] : 0u;
                    xyv[j] = (k0 + j < passes) ? (uint32_t)s_xy[min(ai, nea - 1)] : 0u;
                    m += dm;
                    const bool wrap = m >= L;
                    m = wrap ? m - L : m;
                    ai += di + (wrap ? 1 : 0);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (k0 + j >= passes) break;
                    const int q4 = tq + (k0 + j) * x_TW;
                    if (q4 < nf4) {
                        const uint32_t bits = (((wv[j] >> shift) & 0xFu) * 0x00204081u) & 0x01010101u;
                        // x goes to byte 3 - m, y to byte 4 - m of this float4 (m == 4: x was the last float of the one before)
                        const uint32_t xy = (mv[j] <= 3) ? (xyv[j] << ((24 - 8 * mv[j]) & 31)) : ((mv[j] == 4) ? (xyv[j] >> 8) : 0u);
                        const uint32_t b = opaque(bits | xy);
                        float4 v;
                        v.x = (float)(b & 0xFFu);
                        v.y = (float)((b >> 8) & 0xFFu);
                        v.z = (float)((b >> 16) & 0xFFu);
                        v.w = (float)(b >> 24);
                        store4(nt, (uint32_t)q4 << 4, v);
                    }
                }
            }
        };
        if (x_worker && xy_bytes) {
            if constexpr (Cfg::kNT == 1) single_pass(yes_t{});
            else if constexpr (Cfg::kNT == 0) single_pass(no_t{});
            else {  // (two copies of the pass, one taken: a scalar branch on a workgroup-uniform flag)
                if (k_nt) single_pass(yes_t{}); else single_pass(no_t{});
            }
        }
        // normalised coordinates are fractions: bulk pass over every float4 that holds no coordinate slot (all but ~2 in
        // 18), then a second pass for the coordinate slots
        if (x_worker && !xy_bytes) {
            const int shift = (x_tid & 7) << 2, words_per_pass = x_TW >> 3;  // (x_TW % 8 == 0)
            const uint32_t *wp = s_obits + (x_tid >> 3);
            const int dm = (4 * x_TW) % L;
            int m = (4 * x_tid + 3) % L;
            const int passes = (nf4 + x_TW - 1) / x_TW;
            for (int k0 = 0; k0 < passes; k0 += 8) {
                uint32_t wv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[j] = (k0 + j < passes) ? wp[(k0 + j) * words_per_pass] : 0u;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (k0 + j >= passes) break;
                    const int q4 = x_tid + (k0 + j) * x_TW;
                    if (q4 < nf4 && m >= 5) store4(no_t{}, (uint32_t)q4 << 4, spread((wv[j] >> shift) & 0xFu));
                    m += dm;
                    m = (m >= L) ? m - L : m;
                }
            }
        }
        // coordinate pass: per agent, the one or two float4s that hold its x (element i*L) and y (i*L + 1)
        if (x_worker && !xy_bytes)
        for (int i = x_tid; i < nea; i += x_TW) {
            const int g = i * L, q4 = g >> 2, pos = g & 3;
            const float fx = s_fx[i], fy = s_fy[i];
            if (q4 < nf4) {
                float4 v = expand(q4);
                if (pos == 0) { v.x = fx; v.y = fy; }
                else if (pos == 1) { v.y = fx; v.z = fy; }
                else if (pos == 2) { v.z = fx; v.w = fy; }
                else { v.w = fx; }
                out4[q4] = v;
            }
            if (pos == 3 && q4 + 1 < nf4) {
                float4 v = expand(q4 + 1);
                v.x = fy;
                out4[q4 + 1] = v;
            }
        }
        if (x_worker)
        for (int g = (nf4 << 2) + x_tid; g < nf; g += x_TW) {  // < 4 leftover floats (partial last workgroup)
            const int i = g / L, k = g - i * L;
            out[g] = (k >= 2) ? (((s_obits[g >> 5] >> (g & 31)) & 1u) ? 1.0f : 0.0f)
                              : (k == 0 ? s_fx[i] : s_fy[i]);
        }
    }
    else {  // IMAGE: every element is a bit of the string; no coordinate slots
        const int Limg = k_n_layers * CELLS;
        const int nf = nea * Limg, nf4 = nf >> 2;
        float *out = obs_t + (size_t)e0 * N * Limg;  // 16-byte aligned: e0 is a multiple of 4
        if (worker) {
            const int shift = (tid & 7) << 2, words_per_pass = TW >> 3;  // (see the FLATTENED bulk pass)
            const uint32_t *wp = s_obits + (tid >> 3);
            const int passes = (nf4 + TW - 1) / TW;
            for (int k = 0; k < passes; ++k) {
                const int q4 = tid + k * TW;
                if (q4 >= nf4) break;
                const uint32_t b = opaque((((wp[k * words_per_pass] >> shift) & 0xFu) * 0x00204081u) & 0x01010101u);  // 4 bits -> 4 bytes
                float4 v;
                v.x = (float)(b & 0xFFu);
                v.y = (float)((b >> 8) & 0xFFu);
                v.z = (float)((b >> 16) & 0xFFu);
                v.w = (float)(b >> 24);
                float4 *dst = reinterpret_cast<float4 *>(reinterpret_cast<char *>(out) + (size_t)opaque((uint32_t)q4 << 4));
                // (AGENT_DIRECTION patches its cells afterwards: cached)
                const bool nt = Cfg::kNT == 1 ? true : Cfg::kNT == 0 ? false : (k_nt != 0);
                if (nt && !(k_transposed & 1)) store_f4_nt(dst, v); else *dst = v;
            }
        }
        if (worker)
        for (int g = (nf4 << 2) + tid; g < nf; g += TW) out[g] = ((s_obits[g >> 5] >> (g & 31)) & 1u) ? 1.0f : 0.0f;
        if (k_transposed & 1) {
            // AGENT_DIRECTION (:547-552): the marked cells hold dir + 1, not 1.  Patched after every 0/1 store of
            // the workgroup has completed (full barrier: vmcnt), one thread per (agent, image row) as in P7.
            dma_wait();
            __syncthreads();
            if (worker)
            for (int w = tid; w < nea * WIN; w += TW) {
                const int i = w / WIN, r = w - i * WIN;
                const int e = rw_div18(i, mN);
                const int ax = s_ax[i], ay = s_ay[i], d = k_directional ? s_dir[i] : DIR_UP;
                for (int cc = 0; cc < WIN; ++cc) {
                    int wr = r, wc = cc;
                    if (d == DIR_DOWN) { wr = WIN - 1 - r; wc = WIN - 1 - cc; }
                    else if (d == DIR_LEFT) { wr = WIN - 1 - cc; wc = r; }
                    else if (d == DIR_RIGHT) { wr = cc; wc = WIN - 1 - r; }
                    const int y = ay - R + wr, x = ax - R + wc;
                    if ((unsigned)x >= (unsigned)W || (unsigned)y >= (unsigned)H || x >= H || y >= W) continue;
                    const int ida = s_ga[e * HW + x * W + y] & 0x7f;
                    if (!ida) continue;
                    const float v = (float)(s_dir[e * N + ida - 1] + 1);
#pragma unroll
                    for (int l = 0; l < 8; ++l)
                        if (l < k_n_layers && k_layer[l] == LAYER_AGENT_DIRECTION) out[(size_t)i * Limg + (l * WIN + r) * WIN + cc] = v;
                }
            }
        }
    }
    RW_MARK(TL_OBS_STORED);
    RW_PIPE_MARK(5, 1);
    if (!split && kRollout) write_back(wave, nw);
    }  // fused-rollout step loop
    RW_MARK(TL_END);
    if (RW_RARE(tl_on) && lane == 0) {  // where each wavefront ran: slot 10 = 4 x 16 bits of HW_ID, slot 11 = XCC id
        atomicOr(reinterpret_cast<unsigned long long *>(la.timeline + (size_t)blockIdx.x * TL_MARKS + 10),
                 (unsigned long long)(hw_id() & 0xFFFFu) << (16 * (wave & 3)));
        if (wave == 0) la.timeline[(size_t)blockIdx.x * TL_MARKS + 11] = xcc_id();
    }
#undef RW_MARK
#undef RW_PIPE_MARK
#undef RW_AG_MARK
}

}  // namespace rw
