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

#ifndef RW_STATS_BUILD
#define RW_STATS_BUILD 0
#endif

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
// ImageLayer values of the reference (rware/warehouse.py:59-70); 3 and 4 are written with transposed indices there (:552, :558) and
// reproduced as is, including the IndexError condition (STATUS_IMAGE_INDEX; include/rware_hip.h)
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
    // event counters (RW_STATS_ON; launches carry OP_FLAG_STATS): per env, running totals since rw_create — never reset by the engine.
    // The reference keeps none (`info` is {}, rware/warehouse.py:746-747); read only behind the flag, nullptr otherwise
    int32_t *stat_deliveries;    // [B] shelf deliveries (:907-927)
    int32_t *stat_failed_moves;  // [B] FORWARD requests the step turned into NOOP: shelf-block cancel (:843-846) + failed movers (:871-876)
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
enum : int { OP_FLAG_TIMELINE = 0x100,
             OP_FLAG_STATS = 0x200,    // count deliveries / failed moves into Params::stat_* (see count_events, rware_phase_goals.h)
             OP_FLAG_PRIO = 0x400 };   // raise the wavefronts' priority until the agent phases are done (see the kernel's prologue)
enum : int { TL_START = 0, TL_ZEROED, TL_DMA_ISSUED, TL_ENV_LOADED, TL_LOADED, TL_AGENTS, TL_RESET, TL_OBS_BITS,
             TL_OBS_STORED, TL_END,  // 10, 11: where the wavefronts ran
             TL_AG_RECORD = 12, TL_AG_CELLS, TL_AG_WINNERS, TL_AG_APPLIED, TL_AG_GOALS,  // inside the agent phases (wavefront 0)
             // the pipelined build, chunk it < 4 of the workgroup, slot TL_PIPE + 8 * it + ...: behind barrier A | wavefront 0 done gathering |
             // wavefront 3 done with self bits + write-back | behind barrier B | wavefront 0 done with the next chunk's agent phases |
             // wavefront 1 done expanding | wavefront 3 has issued the stage-in of chunk it + 2 | ... and has seen it land (next stage)
             // (TL_PIPE sits behind the agent-phase slots: TL_AG_GOALS == 16.  Two stamps belong to the stage BEFORE chunk 0's — its agent
             //  phases ran in the prologue, the stage-in of chunk 1 was issued there — and get slots of their own)
             TL_PIPE_FIRST_AG = 17, TL_PIPE_FIRST_DMA = 18, TL_PIPE = 24, TL_MARKS = 56 };

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
             ENVI_NDELIV = 6,  // deliveries of the env's latest step — written by goals_and_termination only, i.e. valid iff that step
                               // delivered (ENVI_INACTIVE == 0 behind a step); read by count_events
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

// rw_debug_store_floor: writes `per_wg` floats per workgroup (one step's observation chunk) with the step kernel's store instruction and
// nothing else — the floor of any kernel that has to produce those observations (measurement aid)
template <bool kNT>
__global__ void __launch_bounds__(256) rware_store_floor_kernel(float *obs, int per_wg, size_t total) {
    const size_t base = (size_t)blockIdx.x * (size_t)per_wg;
    if (base >= total) return;
    const int n4 = (int)((total - base < (size_t)per_wg ? total - base : (size_t)per_wg) >> 2);
    float4 *o = reinterpret_cast<float4 *>(obs + base);
    for (int i = threadIdx.x; i < n4; i += blockDim.x) {
        const float4 v = float4{0.0f, 1.0f, 0.0f, 0.0f};
        if constexpr (kNT) store_f4_nt(o + i, v); else o[i] = v;
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

// Which per-step builds ask for the register budget of 8 wavefronts per SIMD (amdgpu_waves_per_eu): 8-env workgroups hold a
// 16384-env batch in ONE round only if 8 of them fit a CU, i.e. 8 wavefronts per SIMD —
//   exact builds, 7 / 8 agents   (one of them — 8 agents, 16 queue slots — came out at 66 VGPRs, 7 per CU: 11.7 instead of ~9.6 us)
//   9 .. 19 agents, agent-count-static and (run-time compiled) exact builds alike   (104 scalar registers otherwise — the
//       run-time queue length and what hangs on it — so 7 per CU; with the budget: 54 .. 62 VGPRs, no scratch.  Round 4,
//       B = 16384: small-10ag 16.0 -> 13.8 us, small-12ag 18.3 -> 15.8.  With the all-gather agent phases the budget cost spills
//       from 14 agents on — 8 .. 44 bytes per lane; the per-cell exchange, kCell, needs no register arrays of N entries.)
template <int R, typename Cfg, bool kRollout>
constexpr bool want_occ8() {
    if (kRollout) return false;
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
    // Event counters: compiled in only where RW_STATS_BUILD is set — the generic kernels and the run-time compiled builds rw_create asks for
    // when the caller wants counters (rware_jit.h).  The ahead-of-time exact-shape builds do not carry the code: even switched off and out
    // of line it moved 43 of them over a register step (BASELINE config 5's kernel 62 -> 66 VGPRs, 8 -> 7 workgroups per CU; round 6).
#if RW_STATS_BUILD
    const bool stats_on = (la.op & OP_FLAG_STATS) != 0;  // (the flag is preloaded: a scalar test where the counters are off)
#else
    constexpr bool stats_on = false;
#endif
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
    // Wavefront priority (round 6; rw_create decides, OP_FLAG_PRIO): everything in front of the first observation store — stage-in, agent
    // phases — is a chain of dependent steps on one or two wavefronts per workgroup, the stores behind it are bandwidth; a CU holds up to
    // eight workgroups in different places of that sequence.  The chain runs at priority 3 (s_setprio: the SIMD's issue arbiter picks
    // ready wavefronts of higher priority first) and drops back to 0 behind the agent-phase barrier, so a wavefront on the chain is
    // not queued behind the gather / expansion work of its neighbours: small-4ag x 32768 10.75 -> 9.06 us per step, medium-6ag-hard
    // x 65536 24.4 -> 22.7, x 8192 6.24 -> 5.94, small-12ag 15.1 -> 14.6 (profiles/r06_prio_*.txt; the start-staggered 13 .. 16-agent
    // launches lose with it and are left alone; the fused rollouts raise it again at the top of every step: small-4ag 3.95 -> 3.72 us per
    // step).  A hint to the scheduler, never a different result.
    const bool prio_on = !kPipe && (la.op & OP_FLAG_PRIO) != 0;
    if (prio_on) wave_priority<3>();
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
    // 9 .. 19 agents, per-step builds (kCell): the exchange goes through the per-cell agent layer in LDS instead of all-gathers — who
    // stands on my target cell is ONE byte read, who competes for it is a look at its four neighbours, follower depth and the
    // chain walk chase pointers — O(1) per agent where the gathers are O(N) moves + O(N) compares per agent.  These kernels are
    // bound by instruction issue: at 16 agents the gathers were most of the agent phases' ~800 instructions per
    // wavefront.  (Up to 8 agents the gathers are DPP moves or a handful of ds_bpermute and stay.)
    constexpr bool kCell = kDirect && !kRollout && Cfg::kN >= 9;
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
    // (... a stamp that belongs to the chunk BEFORE this one — work that ran one stage ago: slot k of chunk it - 1, for it = 1 .. 4; it == 0: `first`)
#define RW_PIPE_MARK_PREV(k, w, first) do { if (kPipe && RW_RARE(tl_on) && it <= 4 && tid == 64 * (w)) la.timeline[(size_t)blockIdx.x * TL_MARKS + (it == 0 ? (first) : TL_PIPE + 8 * (it - 1) + (k))] = wall_clock64(); } while (0)
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

    // The body from here on is assembled from phase files (rware_phase_*.h), included in program order — textual units, not functions:
    // see the note at the top of each.  What stays in this file is the skeleton: which phase runs where, the barriers between them,
    // the step loop of the fused rollout and the chunk loop of the pipelined flow.
#include "rware_phase_stage_in.h"
#include "rware_phase_pipe.h"
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
        if (kRollout && prio_on) wave_priority<3>();  // (fused rollout: the next step's chain)
    }
#include "rware_phase_goals.h"
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
#include "rware_phase_agents_reg.h"
    } else {
#include "rware_phase_agents_lds.h"
    }
    RW_PIPE_MARK_PREV(4, 0, TL_PIPE_FIRST_AG);  // (the agent phases of this chunk ran one stage ago: slot 4 of chunk it - 1)
    lds_barrier();  // (PIPE: barrier A)
    if (prio_on) wave_priority<0>();
    RW_MARK(TL_AGENTS);
    RW_PIPE_MARK(0, 0);
    // PIPE: the stage-in wavefront 3 issued one stage ago (chunk it + 1, for the agent phases that start behind barrier B) is waited
    // for HERE, behind barrier A and on wavefront 3 only: it has had a whole stage to land, and the gather does not wait for it
    if constexpr (kPipe) { if (pipe_wave == 3) { dma_wait(); RW_PIPE_MARK_PREV(7, 3, TL_PIPE_FIRST_DMA); } }

#include "rware_phase_reset.h"
    RW_MARK(TL_RESET);

#include "rware_phase_write_back.h"
    if (!split && !kRollout) write_back(wave, nw);

#include "rware_phase_gather.h"
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

#include "rware_phase_expand.h"
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
#undef RW_PIPE_MARK_PREV
#undef RW_AG_MARK
}

}  // namespace rw
