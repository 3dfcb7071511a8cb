// rware_capi.hip — host side of the C-ABI declared in include/rware_hip.h.
//
// Owns the HBM-resident batched state of `num_envs` warehouses on one HIP device and
// enqueues the fused step kernel (rware_kernels.h) on one stream.  No CPU fallback exists:
// without a usable HIP device rw_create fails with RW_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <unistd.h>
#include <hip/hip_ext.h>

#include <algorithm>
#include <atomic>
#include <condition_variable>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <mutex>
#include <string>
#include <thread>
#include <utility>
#include <vector>

#include "../../include/rware_hip.h"
#include "rware_hooks.h"
#include "rware_kernel_table.h"
#include "rware_kernels.h"
#include "rware_static_table.h"
#ifndef RW_NO_JIT
#include "rware_jit.h"
#endif

namespace {

thread_local std::string g_create_error;

struct Buf {
    void *ptr = nullptr;
    size_t bytes = 0;
};

}  // namespace

struct rw_engine {
    rw_config cfg{};
    rw::Params prm{};
    int S = 0, L = 0, OW = 0;
    int E = 0, T = 0, n_wg = 0;
    // the fused rollout's own launch geometry: the same as the per-step kernel's except where rw_create gives the per-step launches
    // smaller workgroups than the rollouts want (13 .. 16 agents, see there)
    int roll_n_wg = 0;
    size_t roll_lds_bytes = 0;
    int stagger_ticks = 0, stagger_shift = 0;  // start stagger of a CU's first eight workgroups (multi-round launches: see the kernel's prologue)
    bool prio_rollout = false; // ... the fused rollouts (every step of the launch raises it again)
    bool prio = false;         // per-step launches carry OP_FLAG_PRIO: raised wavefront priority up to the agent-phase barrier (rw_info::wave_priority)
    size_t lds_bytes = 0;
    hipStream_t stream = nullptr;
    bool own_stream = false;
    Buf buf[RW_BUF_KIND_COUNT];
    uint32_t *d_highway_bits = nullptr;
    int32_t *d_shelf_init = nullptr;
    uint8_t *d_mask = nullptr;
    int32_t *h_actions = nullptr;      // pinned staging buffer of rw_step (host actions)
    hipEvent_t h_actions_free = nullptr;  // recorded after the staging buffer's copy to the device
    void (*kernel)(const rw::Params *, const int32_t *, const int32_t, const int32_t, float *, float *, uint8_t *, const uint8_t *, uint64_t *, const int64_t, const int64_t, const int64_t, const int64_t) = nullptr;  // the step kernel instance this engine launches
    void (*kernel_rollout)(const rw::Params *, const int32_t *, const int32_t, const int32_t, float *, float *, uint8_t *, const uint8_t *, uint64_t *, const int64_t, const int64_t, const int64_t, const int64_t) = nullptr;  // its fused multi-step (rollout) sibling
    rw_tab::step_kernel_t kernel_nt = nullptr;  // the per-step kernel with non-temporal observation stores, where the build has one
    // the chunk-pipelined persistent build of this shape (rware_kernels.h "PIPE"), where the table has one and rw_create's rule (or
    // the caller: RW_PIPE_ON / RW_PIPE_OFF) wants it: OP_STEP launches go through it — `pipe_grid` persistent workgroups walk the
    // B / pipe_E chunks — everything else (reset, refresh, fused rollouts) through the classic kernels on the same state
    rw_tab::step_kernel_t pipe_kernel = nullptr;
    int pipe_E = 0, pipe_grid = 0;
    size_t pipe_lds = 0;
    rw::Params *d_prm = nullptr;  // device copy of `prm` (constant for the engine's lifetime)
    rw::LaunchArgs la{};          // per-launch defaults: the engine's own output buffers
    bool specialised = false;
    bool grid_stale = false;   // steps / resets have run since RW_BUF_GRID was last rebuilt (refresh_grid)
    bool agents_stale = false; // ... since RW_BUF_AGENT_X .. _DELIVERED were last unpacked from the records (refresh_agents)
    bool counters_stale = false; // ... since RW_BUF_STEPS / _INACTIVE / _NEED_RESET were last unpacked from the counter records
    int32_t *d_cnt = nullptr;  // [B][2] counter records {steps | need_reset << 31, inactive}: what the step kernels read and write
    size_t cnt_off = 0;
    // run-time specialised build (rware_jit.cpp): the code object's module and its two kernels; launched instead of `kernel` /
    // `kernel_rollout` when present
    hipModule_t jit_module = nullptr;
    hipFunction_t jit_step = nullptr, jit_rollout = nullptr;
    int jit_state = 0;         // rw_info::jit
    int jit_nt = 0;            // the run-time build's observation-store mode
    std::string jit_log;
    bool captured = false;     // a launch of this engine was recorded into a HIP graph: replays run without any host code, so the
                               // two flags above can no longer be trusted — the derived views are rebuilt whenever asked for
    size_t rec_off = 0;
    uint32_t *d_rec = nullptr; // [B][N] packed agent records (rw::rec_pack): the agents' state as the step kernels keep it
    void *slab = nullptr;      // the single device allocation behind every buffer below
    size_t shadow_off = 0;
    void *d_shadow = nullptr;  // compact shelf layer (uint8 when S <= 255, else uint16), the kernel's read path
    int build_kind = 0;        // rw_info::build_kind
    bool q_runtime = false;    // the chosen build reads the request-queue length at run time (StaticEntry::Q == -1)
    bool wide = false;
    bool image = false;        // IMAGE / IMAGE_DICT observation kernels
    bool stats = false;        // RW_STATS_ON: every launch carries OP_FLAG_STATS, RW_BUF_STAT_* are allocated
    int msg_bits = 0;          // communication bits per agent (FLATTENED only)
    int32_t *d_status = nullptr;
    hipEvent_t events[8]{};
    std::vector<uint8_t> h_highways;
    std::string err;
    hipDeviceProp_t prop{};
};

namespace {

int fail(rw_engine *eng, int code, const char *fmt, ...) {
    char tmp[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(tmp, sizeof tmp, fmt, ap);
    va_end(ap);
    if (eng) eng->err = tmp;
    else g_create_error = tmp;
    return code;
}

#define RW_HIP(eng, call)                                                                       \
    do {                                                                                        \
        hipError_t e_ = (call);                                                                 \
        if (e_ != hipSuccess)                                                                   \
            return fail(eng, RW_ERR_HIP, "%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                                    \
    } while (0)

using rw_tab::step_kernel_t;

// The exact-shape and size-static kernel builds live in rware_static.hip, one translation unit per group of the table in
// rware_static_table.h (they compile in parallel); rw_tab::static_group(g, &n) hands out group g's entries.
using rw_tab::StaticEntry;
}  // namespace
namespace rw_tab {
const StaticEntry *static_group(int group, int *n) {
    using fn_t = const StaticEntry *(*)(int *);
    static const fn_t kGroups[kStaticGroups] = {static_group_0,  static_group_1,  static_group_2,  static_group_3,  static_group_4,  static_group_5,
                                                static_group_6,  static_group_7,  static_group_8,  static_group_9,  static_group_10, static_group_11,
                                                static_group_12, static_group_13, static_group_14, static_group_15, static_group_16, static_group_17,
#if RW_WITH_PIPE
                                                static_group_18};
#else
                                                nullptr};  // (group 18 — the chunk-pipelined persistent builds — is only in a `make PIPE=1` library)
    if (!kGroups[group]) { *n = 0; return nullptr; }
#endif
    return kGroups[group](n);
}
}  // namespace rw_tab
namespace {

int launch(rw_engine *eng, rw::LaunchArgs la, int op, bool rollout = false, hipEvent_t start = nullptr, hipEvent_t stop = nullptr) {
    la.op = op | (la.timeline ? rw::OP_FLAG_TIMELINE : 0);
    const bool pipe = eng->pipe_kernel && op == rw::OP_STEP && !rollout;  // (persistent workgroups: no start stagger)
    if (!pipe) la.op |= (eng->stagger_ticks & 0xff) << 16 | (eng->stagger_shift & 0xf) << 24;
    if (eng->stats) la.op |= rw::OP_FLAG_STATS;
    if (rollout ? eng->prio_rollout : eng->prio) la.op |= rw::OP_FLAG_PRIO;
    if (op != rw::OP_OBS) eng->grid_stale = eng->agents_stale = eng->counters_stale = true;  // the kernels keep the shadow and the packed agent records current, not the int32 views
    if (!eng->own_stream && !eng->captured) {  // (a stream of the caller's may be capturing; the engine's own stream never is)
        hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
        if (hipStreamIsCapturing(eng->stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) eng->captured = true;
        else (void)hipGetLastError();
    }
    if (pipe) {
        if (start || stop)
            hipExtLaunchKernelGGL(eng->pipe_kernel, dim3(eng->pipe_grid), dim3(256), eng->pipe_lds, eng->stream, start, stop, 0,
                                  (const rw::Params *)eng->d_prm, RW_LAUNCH_ARGS(la));
        else
            hipLaunchKernelGGL(eng->pipe_kernel, dim3(eng->pipe_grid), dim3(256), eng->pipe_lds, eng->stream,
                               (const rw::Params *)eng->d_prm, RW_LAUNCH_ARGS(la));
        RW_HIP(eng, hipGetLastError());
        return RW_OK;
    }
    const int n_wg = rollout ? eng->roll_n_wg : eng->n_wg;
    const size_t lds = rollout ? eng->roll_lds_bytes : eng->lds_bytes;
    if (eng->jit_step) {  // a run-time specialised build: the same launch through the module API
        const rw::Params *cp = eng->d_prm;
        void *args[13] = {&cp, &la.actions, &la.op, &la.n_steps, &la.obs, &la.rewards, &la.terminated, &la.reset_mask, &la.timeline,
                          &la.act_stride, &la.obs_stride, &la.rew_stride, &la.term_stride};
        hipFunction_t f = rollout ? eng->jit_rollout : eng->jit_step;
        if (start || stop)
            RW_HIP(eng, hipExtModuleLaunchKernel(f, (uint32_t)n_wg * (uint32_t)eng->T, 1, 1, (uint32_t)eng->T, 1, 1, lds, eng->stream,
                                                 args, nullptr, start, stop, 0));
        else
            RW_HIP(eng, hipModuleLaunchKernel(f, (uint32_t)n_wg, 1, 1, (uint32_t)eng->T, 1, 1, (uint32_t)lds, eng->stream, args, nullptr));
        return RW_OK;
    }
    if (start || stop)  // the events ride on this dispatch (its own start / end timestamps): no marker packets in the stream
        hipExtLaunchKernelGGL(rollout ? eng->kernel_rollout : eng->kernel, dim3(n_wg), dim3(eng->T), lds,
                              eng->stream, start, stop, 0, (const rw::Params *)eng->d_prm, RW_LAUNCH_ARGS(la));
    else
        hipLaunchKernelGGL(rollout ? eng->kernel_rollout : eng->kernel, dim3(n_wg), dim3(eng->T), lds,
                           eng->stream, (const rw::Params *)eng->d_prm, RW_LAUNCH_ARGS(la));
    RW_HIP(eng, hipGetLastError());
    return RW_OK;
}

// the shelf shadow mirrors layer 1 of RW_BUF_GRID; rebuilt after any host write of the grid
int rebuild_shadow(rw_engine *eng) {
    const int B = eng->prm.B, HW = eng->prm.HW;
    const size_t n = (size_t)B * HW;
    const unsigned blocks = (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024);
    if (eng->wide)
        hipLaunchKernelGGL((rw::rware_shadow_kernel<uint16_t>), dim3(blocks), dim3(256), 0, eng->stream,
                           (const int32_t *)eng->buf[RW_BUF_GRID].ptr, (uint16_t *)eng->d_shadow, B, HW);
    else
        hipLaunchKernelGGL((rw::rware_shadow_kernel<uint8_t>), dim3(blocks), dim3(256), 0, eng->stream,
                           (const int32_t *)eng->buf[RW_BUF_GRID].ptr, (uint8_t *)eng->d_shadow, B, HW);
    RW_HIP(eng, hipGetLastError());
    return RW_OK;
}

// RW_BUF_GRID is a derived view: brought up to date from the shelf shadow and the agent coordinates when it is asked for
int refresh_grid(rw_engine *eng) {
    if (!eng->grid_stale && !eng->captured) return RW_OK;
    const int B = eng->prm.B, HW = eng->prm.HW, N = eng->prm.N;
    const size_t n = (size_t)B * HW, na = (size_t)B * N;
    // (grid-stride loops, one wavefront per workgroup, 64 cells / agents per thread: a few thousand workgroups at the big batches)
    const unsigned blocks = (unsigned)((n + 4095) / 4096 < 4096 ? (n + 4095) / 4096 : 4096);
    const unsigned ablocks = (unsigned)((na + 4095) / 4096 < 1024 ? (na + 4095) / 4096 : 1024);
    int32_t *grid = (int32_t *)eng->buf[RW_BUF_GRID].ptr;
    const uint32_t *rec = eng->d_rec;
    if (eng->wide) {
        hipLaunchKernelGGL((rw::rware_grid_cells_kernel<uint16_t>), dim3(blocks), dim3(64), 0, eng->stream,
                           (const uint16_t *)eng->d_shadow, grid, B, HW);
        hipLaunchKernelGGL((rw::rware_grid_agents_kernel<uint16_t>), dim3(ablocks), dim3(64), 0, eng->stream, rec, grid, B, HW, N);
    } else {
        hipLaunchKernelGGL((rw::rware_grid_cells_kernel<uint8_t>), dim3(blocks), dim3(64), 0, eng->stream,
                           (const uint8_t *)eng->d_shadow, grid, B, HW);
        hipLaunchKernelGGL((rw::rware_grid_agents_kernel<uint8_t>), dim3(ablocks), dim3(64), 0, eng->stream, rec, grid, B, HW, N);
    }
    RW_HIP(eng, hipGetLastError());
    eng->grid_stale = false;
    return RW_OK;
}

// RW_BUF_AGENT_X .. RW_BUF_AGENT_DELIVERED are derived views of the packed agent records: unpacked when somebody asks
bool is_agent_view(int kind) { return kind >= RW_BUF_AGENT_X && kind <= RW_BUF_AGENT_DELIVERED; }
unsigned agent_blocks(size_t n) { return (unsigned)((n + 255) / 256 < 1024 ? (n + 255) / 256 : 1024); }
int refresh_agents(rw_engine *eng) {
    if (!eng->agents_stale && !eng->captured) return RW_OK;
    const size_t n = (size_t)eng->prm.B * eng->prm.N;
    hipLaunchKernelGGL((rw::rware_unpack_agents_kernel<>), dim3(agent_blocks(n)), dim3(256), 0, eng->stream, (const uint32_t *)eng->d_rec,
                       (int32_t *)eng->buf[RW_BUF_AGENT_X].ptr, (int32_t *)eng->buf[RW_BUF_AGENT_Y].ptr, (int32_t *)eng->buf[RW_BUF_AGENT_DIR].ptr,
                       (int32_t *)eng->buf[RW_BUF_AGENT_CARRY].ptr, (int32_t *)eng->buf[RW_BUF_AGENT_DELIVERED].ptr, n, eng->prm.W);
    RW_HIP(eng, hipGetLastError());
    eng->agents_stale = false;
    return RW_OK;
}
// after a host write of one of the five views: the views are the state now, the records follow
int pack_agents(rw_engine *eng) {
    const size_t n = (size_t)eng->prm.B * eng->prm.N;
    hipLaunchKernelGGL((rw::rware_pack_agents_kernel<>), dim3(agent_blocks(n)), dim3(256), 0, eng->stream, eng->d_rec,
                       (const int32_t *)eng->buf[RW_BUF_AGENT_X].ptr, (const int32_t *)eng->buf[RW_BUF_AGENT_Y].ptr,
                       (const int32_t *)eng->buf[RW_BUF_AGENT_DIR].ptr, (const int32_t *)eng->buf[RW_BUF_AGENT_CARRY].ptr,
                       (const int32_t *)eng->buf[RW_BUF_AGENT_DELIVERED].ptr, n, eng->prm.W);
    RW_HIP(eng, hipGetLastError());
    eng->agents_stale = false;
    return RW_OK;
}

// RW_BUF_STEPS / RW_BUF_INACTIVE / RW_BUF_NEED_RESET are derived views of the per-env counter records
bool is_counter_view(int kind) { return kind == RW_BUF_STEPS || kind == RW_BUF_INACTIVE || kind == RW_BUF_NEED_RESET; }
int refresh_counters(rw_engine *eng) {
    if (!eng->counters_stale && !eng->captured) return RW_OK;
    const size_t n = (size_t)eng->prm.B;
    hipLaunchKernelGGL((rw::rware_unpack_counters_kernel<>), dim3(agent_blocks(n)), dim3(256), 0, eng->stream, (const int32_t *)eng->d_cnt,
                       (int32_t *)eng->buf[RW_BUF_STEPS].ptr, (int32_t *)eng->buf[RW_BUF_INACTIVE].ptr, (uint8_t *)eng->buf[RW_BUF_NEED_RESET].ptr, n);
    RW_HIP(eng, hipGetLastError());
    eng->counters_stale = false;
    return RW_OK;
}
int pack_counters(rw_engine *eng) {
    const size_t n = (size_t)eng->prm.B;
    hipLaunchKernelGGL((rw::rware_pack_counters_kernel<>), dim3(agent_blocks(n)), dim3(256), 0, eng->stream, eng->d_cnt,
                       (const int32_t *)eng->buf[RW_BUF_STEPS].ptr, (const int32_t *)eng->buf[RW_BUF_INACTIVE].ptr,
                       (const uint8_t *)eng->buf[RW_BUF_NEED_RESET].ptr, n);
    RW_HIP(eng, hipGetLastError());
    eng->counters_stale = false;
    return RW_OK;
}

size_t elem_size(int kind) {
    switch (kind) {
        case RW_BUF_OBS: case RW_BUF_REWARDS: case RW_BUF_FEATURES: case RW_BUF_FINAL_OBS: case RW_BUF_FINAL_FEATURES: return 4;
        case RW_BUF_TERMINATED: case RW_BUF_TRUNCATED: case RW_BUF_NEED_RESET: return 1;
        case RW_BUF_RNG: return 8;
        default: return 4;
    }
}

}  // namespace

extern "C" {

int rw_abi_version(void) { return RW_ABI_VERSION; }

const char *rw_last_error(const rw_engine *eng) { return eng ? eng->err.c_str() : g_create_error.c_str(); }

int rw_seed_state(uint64_t seed, uint64_t out[6]) {
    // numpy SeedSequence(entropy=seed).generate_state(4, uint64) -> PCG64(seed_seq) initial state.
    // Replaces gymnasium.utils.seeding.np_random(seed), reached from Warehouse.reset(seed=...)
    // (rware/warehouse.py:758-760).
    if (!out) return RW_ERR_INVALID_ARG;
    uint32_t ent[2] = {(uint32_t)seed, (uint32_t)(seed >> 32)};
    const int n_ent = ent[1] ? 2 : 1;
    uint32_t pool[4], hc = 0x43b0d7e5u;
    auto hashmix = [&hc](uint32_t v) {
        v ^= hc;
        hc *= 0x931e8875u;
        v *= hc;
        v ^= v >> 16;
        return v;
    };
    auto mix = [](uint32_t x, uint32_t y) {
        uint32_t r = 0xca01f9ddu * x - 0x4973f715u * y;
        r ^= r >> 16;
        return r;
    };
    for (int i = 0; i < 4; ++i) pool[i] = hashmix(i < n_ent ? ent[i] : 0u);
    for (int s = 0; s < 4; ++s)
        for (int d = 0; d < 4; ++d)
            if (s != d) pool[d] = mix(pool[d], hashmix(pool[s]));
    uint32_t w[8], hb = 0x8b51f9ddu;
    for (int i = 0; i < 8; ++i) {
        uint32_t v = pool[i & 3] ^ hb;
        hb *= 0x58f38dedu;
        v *= hb;
        v ^= v >> 16;
        w[i] = v;
    }
    uint64_t s64[4];
    for (int i = 0; i < 4; ++i) s64[i] = (uint64_t)w[2 * i] | ((uint64_t)w[2 * i + 1] << 32);
    const rw::u128 initstate = (((rw::u128)s64[0]) << 64) | s64[1];
    const rw::u128 initseq = (((rw::u128)s64[2]) << 64) | s64[3];
    rw::u128 inc = (initseq << 1) | 1u, state = 0;
    state = state * rw::pcg_mult() + inc;
    state += initstate;
    state = state * rw::pcg_mult() + inc;
    out[0] = (uint64_t)(state >> 64);
    out[1] = (uint64_t)state;
    out[2] = (uint64_t)(inc >> 64);
    out[3] = (uint64_t)inc;
    out[4] = 0;
    out[5] = 0;
    return RW_OK;
}

int rw_create(const rw_config *cfg, rw_engine **out) {
    if (!cfg || !out) return fail(nullptr, RW_ERR_INVALID_ARG, "null cfg/out");
    *out = nullptr;
    if (cfg->abi_version != RW_ABI_VERSION)
        return fail(nullptr, RW_ERR_INVALID_ARG, "abi_version %d != %d", cfg->abi_version, RW_ABI_VERSION);
    const int B = cfg->num_envs, H = cfg->grid_h, W = cfg->grid_w, N = cfg->n_agents, Q = cfg->request_queue_size;
    if (B < 1 || H < 1 || W < 1 || N < 1 || N > 64 || Q < 0 || !cfg->highways || !cfg->goals_xy || cfg->n_goals < 1)
        return fail(nullptr, RW_ERR_INVALID_ARG, "bad shape: B=%d H=%d W=%d N=%d Q=%d n_goals=%d", B, H, W, N, Q, cfg->n_goals);
    if (cfg->sensor_range < 1 || cfg->sensor_range > 5)
        return fail(nullptr, RW_ERR_UNSUPPORTED, "sensor_range %d not in 1..5", cfg->sensor_range);
    if (cfg->reward_type < 0 || cfg->reward_type > 2 || cfg->autoreset_mode < 0 || cfg->autoreset_mode > 2)
        return fail(nullptr, RW_ERR_INVALID_ARG, "bad reward_type/autoreset_mode");
    const int obs_type = cfg->observation_type ? cfg->observation_type : RW_OBS_FLATTENED;
    if (obs_type != RW_OBS_FLATTENED && obs_type != RW_OBS_IMAGE && obs_type != RW_OBS_IMAGE_DICT)
        return fail(nullptr, RW_ERR_UNSUPPORTED, "observation_type %d is not accelerated (FLATTENED, IMAGE, IMAGE_DICT are)", obs_type);
    int n_layers = cfg->n_image_layers;
    int layers[8] = {RW_LAYER_SHELVES, RW_LAYER_REQUESTS, RW_LAYER_AGENTS, RW_LAYER_GOALS, RW_LAYER_ACCESSIBLE, 0, 0, 0};
    if (n_layers == 0) n_layers = 5;  // the reference default (rware/warehouse.py:160-166)
    else {
        if (n_layers < 0 || n_layers > 8) return fail(nullptr, RW_ERR_INVALID_ARG, "n_image_layers %d not in 1..8", n_layers);
        for (int l = 0; l < n_layers; ++l) {
            layers[l] = cfg->image_layers[l];
            const int v = layers[l];
            if (v < RW_LAYER_SHELVES || v > RW_LAYER_ACCESSIBLE)
                return fail(nullptr, RW_ERR_INVALID_ARG, "unknown image layer %d", v);
        }
    }
    if (cfg->msg_bits < 0 || cfg->msg_bits > 16) return fail(nullptr, RW_ERR_INVALID_ARG, "msg_bits %d not in 0..16", cfg->msg_bits);
    const int HW = H * W;
    if (HW > 10000) return fail(nullptr, RW_ERR_UNSUPPORTED, "H*W > 10000 (numpy switches choice() algorithm)");
    if (N > HW) return fail(nullptr, RW_ERR_INVALID_ARG, "more agents than cells");
    int S = 0;
    for (int i = 0; i < HW; ++i) S += cfg->highways[i] ? 0 : 1;
    if (Q > S) return fail(nullptr, RW_ERR_INVALID_ARG, "request_queue_size %d > shelves %d", Q, S);
    if (cfg->n_goals > rw::MAX_GOALS) return fail(nullptr, RW_ERR_UNSUPPORTED, "more than %d goal cells", (int)rw::MAX_GOALS);
    for (int g = 0; g < cfg->n_goals; ++g) {
        const int x = cfg->goals_xy[2 * g], y = cfg->goals_xy[2 * g + 1];
        if (x < 0 || x >= W || y < 0 || y >= H) return fail(nullptr, RW_ERR_INVALID_ARG, "goal %d out of the grid", g);
    }

    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1)
        return fail(nullptr, RW_ERR_NO_DEVICE, "no HIP device visible; this engine has no CPU fallback");
    if (cfg->device_id < 0 || cfg->device_id >= n_dev)
        return fail(nullptr, RW_ERR_INVALID_ARG, "device_id %d not in [0,%d)", cfg->device_id, n_dev);

    rw_engine *eng = new (std::nothrow) rw_engine();
    if (!eng) return fail(nullptr, RW_ERR_HIP, "out of host memory");
    auto bail = [&](int code) {
        g_create_error = eng->err;
        rw_destroy(eng);
        return code;
    };
#define RW_HIP_C(call)                                                              \
    do {                                                                            \
        hipError_t e_ = (call);                                                     \
        if (e_ != hipSuccess) {                                                     \
            fail(eng, RW_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_));   \
            return bail(RW_ERR_HIP);                                                \
        }                                                                           \
    } while (0)

    eng->cfg = *cfg;
    eng->cfg.highways = nullptr;
    eng->cfg.goals_xy = nullptr;
    eng->h_highways.assign(cfg->highways, cfg->highways + HW);
    eng->S = S;
    eng->wide = S > 255;
    const int cell_bytes = eng->wide ? 2 : 1;
    const int R = cfg->sensor_range, CELLS = (2 * R + 1) * (2 * R + 1);
    eng->image = obs_type != RW_OBS_FLATTENED;
    eng->msg_bits = cfg->msg_bits;
    const int AM = 1 + cfg->msg_bits;
    eng->L = eng->image ? n_layers * CELLS : 8 + (7 + cfg->msg_bits) * CELLS;  // floats per agent in RW_BUF_OBS
    // LDS bit-string words per agent (must equal the kernel's OW): the flattened row, or n_layers image planes
    eng->OW = eng->image ? std::max((8 + 7 * CELLS + 31) / 32, (n_layers * CELLS + 31) / 32)
                         : (8 + (7 + cfg->msg_bits) * CELLS + 31) / 32;
    const int SW = (S + 32) / 32;

    RW_HIP_C(hipSetDevice(cfg->device_id));
    RW_HIP_C(hipGetDeviceProperties(&eng->prop, cfg->device_id));
    if (cfg->stream || (cfg->stream_flags & RW_STREAM_USE_GIVEN)) {
        eng->stream = (hipStream_t)cfg->stream;  // with RW_STREAM_USE_GIVEN, NULL is the device's default stream
    } else {
        RW_HIP_C(hipStreamCreateWithFlags(&eng->stream, hipStreamNonBlocking));
        eng->own_stream = true;
    }
    for (auto &ev : eng->events) RW_HIP_C(hipEventCreate(&ev));

    // workgroup geometry: E envs per workgroup (multiple of 4 keeps every chunk 16-byte aligned)
    int E = cfg->envs_per_workgroup, T = cfg->threads_per_workgroup;
    int roll_E = 0;  // != 0: the fused rollout runs on workgroups of this many envs (its own table entry), not on E
    bool wide4 = false;  // 13 .. 16 agents on the 4-env per-step build (see the table search below): priority instead of the start stagger
    if (T == 0) T = 256;
    if (T % 64 || T < 64 || T > 256) {
        fail(eng, RW_ERR_INVALID_ARG, "threads_per_workgroup %d must be 64..256, multiple of 64", T);
        return bail(RW_ERR_INVALID_ARG);
    }
    if (E == 0) {
        const size_t per_env = sizeof(int32_t) * (size_t)rw::make_lds_layout(4, N, Q, HW, SW, eng->OW, cell_bytes, AM).total / 4;
        E = (int)((32 * 1024) / per_env) & ~3;
        if (E < 4) E = 4;
        if (E > 16) E = 16;  // measured best on MI355X for the registered configs (profiles/)
    }
    if (E % 4 || E < 4) {
        fail(eng, RW_ERR_INVALID_ARG, "envs_per_workgroup %d must be a positive multiple of 4", E);
        return bail(RW_ERR_INVALID_ARG);
    }
    {
        // x / N == (x * ceil(2^18 / N)) >> 18 needs x * N < 2^18 for every x < E * N
        if ((long long)E * N * N >= (1 << 18)) {
            fail(eng, RW_ERR_INVALID_ARG, "envs_per_workgroup %d too large for N=%d Q=%d", E, N, Q);
            return bail(RW_ERR_INVALID_ARG);
        }
    }
    const bool want_stats = (cfg->stream_flags & RW_STATS_ON) != 0;
    if (want_stats && !rw_tab::generic_has_stats()) {
        fail(eng, RW_ERR_UNSUPPORTED, "RW_STATS_ON: this library was built without the event-counter code (RW_STATS_BUILD)");
        return bail(RW_ERR_UNSUPPORTED);
    }
    {   // the generic build for this sensor range (rware_generic.hip); an exact-shape build may replace it below
        using pick_t = step_kernel_t (*)(bool, bool, bool, bool);
        static const pick_t kGeneric[5] = {rw_tab::generic_r1, rw_tab::generic_r2, rw_tab::generic_r3, rw_tab::generic_r4,
                                           rw_tab::generic_r5};
        const pick_t pick = kGeneric[(R < 1 ? 1 : R > 5 ? 5 : R) - 1];
        eng->kernel = pick(false, eng->wide, eng->image, eng->msg_bits > 0);
        eng->kernel_rollout = pick(true, eng->wide, eng->image, eng->msg_bits > 0);
    }
    if (!(eng->image && eng->msg_bits > 0)) {  // (image + messages together: generic builds only)
        // Pick a specialised build: exact-shape entries before size-static ones, first match wins.
        const bool geom_default = cfg->envs_per_workgroup == 0 && cfg->threads_per_workgroup == 0;
        const char *pq = rw_hook("RWARE_PREFER_QRT");  // (test / A-B hook: skip the exact (N, Q) builds)
        const bool prefer_qrt = pq && pq[0] == '1';
        // find(want_e): the table's own choice for this batch (want_e == 0), or its entry with want_e envs per 256-thread workgroup
        auto find = [&](int want_e) -> const StaticEntry * {
        const StaticEntry *best = nullptr;
        for (int exact = 1; exact >= 0 && !best; --exact)
            for (int grp = 0; grp < rw_tab::kStaticGroups && !best; ++grp) {
            int n_se = 0;
            const StaticEntry *tab = rw_tab::static_group(grp, &n_se);
            for (int k_se = 0; k_se < n_se; ++k_se) {
                const StaticEntry &se = tab[k_se];
                if ((se.N != 0) != (exact != 0) || se.pipe) continue;
                if (se.image != (eng->image ? 1 : 0) || se.M != eng->msg_bits) continue;
                if (se.NL > 0) {  // a baked-in layer list serves exactly that list
                    uint32_t packed = 0;
                    for (int l = 0; l < n_layers && l < 8; ++l) packed |= (uint32_t)layers[l] << (4 * l);
                    if (se.NL != n_layers || se.layers != packed || se.directional != (cfg->image_directional ? 1 : 0)) continue;
                }
                // (Q == -1: an agent-count-static build — any queue length up to 2 N, read at run time)
                const bool q_ok = se.Q == Q || (se.Q < 0 && Q <= 2 * se.N);
                if (prefer_qrt && se.N != 0 && se.Q >= 0 && !se.image && se.M == 0) continue;  // (test / A-B hook: skip the exact (N, Q) builds)
                const bool shape = se.H == H && se.W == W && se.S == S && se.R == R && (se.N == 0 || (se.N == N && q_ok));
                if (!shape || B % se.E != 0) continue;
                if (want_e ? (se.E == want_e && se.T == 256)
                           : geom_default ? (se.max_B == 0 || B <= se.max_B) : (E == se.E && T == se.T)) { best = &se; break; }
            }
            }
        return best;
        };
        // 13 .. 16 agents at sensor_range 1 (round 6, second session; profiles/r06_1316_matrix.txt — 8 tasks x 10 batches x geometry x stagger x
        // priority): their 8-env workgroups hold TWO full agent wavefronts, which is why these launches wanted a start stagger and lose with
        // the wavefront priority; on 4-env workgroups (one agent wavefront each) they behave like the smaller tasks, and with the priority
        // on and no stagger beat the 8-env rule below one full round of 8-env workgroups (2048 envs: large-16ag 8.17 -> 6.61 us, 4096: 9.52
        // -> 7.35, medium-13ag 9.71 -> 7.61; 8192: 11.7 -> 11.1; 12288: 14.6 -> 13.2) and between one and four rounds (24576: large-16ag 25.1 ->
        // 22.0, 32768: 30.8 -> 27.8, medium-13ag 30.8 -> 27.2, small-15ag 34.4 -> 31.5; 49152: small-16ag 45.5 -> 39.7; 14 agents -2 .. -4 %).
        // At exactly one round (16384 envs) and from four rounds on the staggered 8-env launch stays (16384: medium-13ag 16.2 against 16.6,
        // large-14ag 15.9 / 16.7; 65536: small-16ag 56.7 / 60.9, large-16ag 58.9 / 63.6).  The FUSED ROLLOUTS keep the 8-env build at every
        // batch (4-env: +10 .. +47 %, profiles/r06_1316_rollout_geom.txt): the engine then launches its two kernels with different geometries.
        // 9 .. 12 and 17 .. 19 agents (profiles/r06_e4_small_batches.txt, r06_e4_others.txt): the 4-env build wins while its workgroups stay
        // under ONE round (fewer than 8 per CU: below 8192 envs) — small-10ag x 4096 8.47 -> 7.17 us, x 6144 8.87 -> 7.69, small-12ag x 4096
        // 9.27 -> 7.77, small-19ag x 4096 13.33 -> 11.31, x 6144 14.15 -> 12.06 — and loses from there on (small-10ag x 8192 9.24 -> 10.46).
        // Their fused rollouts: 9 .. 12 agents gain likewise (small-10ag x 4096 6.12 -> 5.02 us per step) and follow; 17 .. 19 agents do not
        // (small-17ag x 4096 9.0 -> 12.9) and keep the 8-env build.
        int want_e = 0;
        bool roll_follows = false;  // the fused rollout runs on the per-step launch's build (else on the table's own choice)
        if (geom_default && R == 1 && N >= 9 && N <= 19 && !eng->image && eng->msg_bits == 0 && B % 4 == 0) {
            const long long resident = 8LL * std::max(1, eng->prop.multiProcessorCount);
            const long long wg8 = ((long long)B + 7) / 8, wg4 = ((long long)B + 3) / 4;
            const bool two_waves = N >= 13 && N <= 16;  // (an 8-env workgroup of theirs holds two FULL agent wavefronts)
            const char *e4 = rw_hook("RWARE_WIDE_E4");  // (A/B and test hook: 0 = never, 1 = at every batch)
            if (e4 ? e4[0] == '1' : two_waves ? (wg8 < resident || (wg8 > resident && wg8 < 4 * resident)) : wg4 < resident) want_e = 4;
            roll_follows = N <= 12;
        }
        const StaticEntry *best = want_e ? find(want_e) : nullptr;
        const StaticEntry *best_roll = nullptr;
        if (best && !roll_follows) best_roll = find(0);            // (the rollouts' build: the table's own choice)
        if (!best || (!roll_follows && (!best_roll || best_roll->Q != best->Q))) { best = find(0); best_roll = nullptr; want_e = 0; }  // (no 4-env build of this shape: the tiny warehouse)
        // event counters (RW_STATS_ON): only kernels compiled with the counting code (RW_STATS_BUILD) will do — the generic ones, a run-time
        // compiled exact-shape build (below), and — in a library whose specialised builds were made with the switch — those
        if (best && want_stats && !rw_tab::static_has_stats()) {
            best = best_roll = nullptr;
            eng->jit_log = "event counters: the ahead-of-time exact-shape builds do not carry the counting code";
        }
        if (best) {
            E = best->E;
            T = best->T;
            eng->kernel = best->fn;
            eng->kernel_nt = best->fn_nt;
            eng->kernel_rollout = best_roll ? best_roll->fn_rollout : best->fn_rollout;
            roll_E = best_roll ? best_roll->E : 0;
            wide4 = want_e == 4;
            eng->specialised = true;
            eng->q_runtime = best->Q < 0;
            eng->build_kind = best->N == 0 ? 3 : best->Q < 0 ? 2 : 1;
        }
    }
    // Observation stores: non-temporal (stream) or cached.  Measured rule (round 3 / 4, same-box A/Bs, profiles/EXPERIMENTS.md):
    // the hint wins wherever a workgroup's observation chunk is small (every registered task up to 12 agents — small-4ag
    // B = 16384 7.13 -> 6.17 us, medium-6ag-hard 7.25 -> 6.26, large-8ag 12.25 -> 11.1, small-12ag 16.7 -> 15.8) and wherever a
    // step's observations outgrow the Infinity Cache (large-16ag r=2 B = 32768 86.3 -> 79.0); it loses for large chunks below
    // that size (large-16ag r=2 B = 16384 37.3 -> 43.7; small-19ag 28.9 vs 27.6 cached; large-16ag r=1: even).
    // (by chunk size in floats: 4544 small-4ag, 6816 small-12ag at 8 envs: the hint wins; 9088 large-16ag: even; 10792 small-19ag,
    //  23424 large-16ag r=2: it loses below the cache size; 9656 small-17ag: 26.1 vs 25.5 cached)
    auto nt_rule = [&](int e_) {
        const long long chunk = (long long)e_ * N * eng->L;                    // floats of one workgroup's observations
        const double obs_mb = (double)B * N * eng->L * 4 / 1e6;
        bool nt = chunk <= 9500 || obs_mb > 240.0;
        const char *pref = rw_hook("RWARE_OBS_STORES");  // (A/B hook: moves the default only — an explicit flag of the caller wins)
        if (pref && !strcmp(pref, "cached")) nt = false;
        if (pref && !strcmp(pref, "stream")) nt = true;
        if (cfg->stream_flags & RW_OBS_STORES_CACHED) nt = false;
        if (cfg->stream_flags & RW_OBS_STORES_STREAM) nt = true;
        return nt;
    };
#ifndef RW_NO_JIT
    {
        // Run-time specialisation (rware_jit.h): a shape without an exact or agent-count-static entry gets an exact-shape build
        // compiled now (or read from the disk cache) — unless the batch is small (the seconds a compile takes only pay off
        // on a long run), the caller said no, or hipRTC is not there.  RW_JIT_FORCE: also for small batches and for shapes
        // that have an ahead-of-time build (tests, A/B).
        int mode = 0;  // 0 auto, -1 off, 1 force
        const char *je = rw_hook("RWARE_JIT");
        if (je && (!strcmp(je, "0") || !strcmp(je, "off"))) mode = -1;
        if (je && !strcmp(je, "force")) mode = 1;
        if (cfg->stream_flags & RW_JIT_OFF) mode = -1;
        if (cfg->stream_flags & RW_JIT_FORCE) mode = 1;
        const bool aot_exact = eng->specialised && eng->build_kind != 3;
        const bool combo_ok = !(eng->image && eng->msg_bits > 0);
        if (combo_ok && (mode == 1 || (mode == 0 && !aot_exact && B >= 4096))) {
            const bool geom_given = cfg->envs_per_workgroup != 0 || cfg->threads_per_workgroup != 0;
            int prefs[5] = {8, 4, 16, 0, 0};
            if (geom_given) { prefs[0] = cfg->envs_per_workgroup ? cfg->envs_per_workgroup : 16; prefs[1] = 0; }
            else if (N <= 2) { prefs[0] = B >= 16384 ? 32 : 16; prefs[1] = 16; prefs[2] = 8; prefs[3] = 4; }
            else if (N <= 4) { prefs[0] = 16; prefs[1] = 8; prefs[2] = 4; }
            else if (N <= 8) { prefs[0] = B <= 16384 ? 8 : 16; prefs[1] = B <= 16384 ? 16 : 8; prefs[2] = 4; }
            int je_ = 0;
            for (int k = 0; k < 5 && !je_; ++k) {
                const int c = prefs[k];
                if (c < 4 || c % 4 || B % c) continue;
                if (((long long)c * HW * cell_bytes) % 16) continue;                         // the stage-in DMA moves whole 16-byte pieces
                if (N <= 19 && c > 4 * (64 / N)) continue;                                   // every env needs its own agent lanes
                if ((long long)c * N * N >= (1 << 18)) continue;
                const size_t lds = sizeof(int32_t) * (size_t)rw::make_lds_layout(c, N, Q, HW, SW, eng->OW, cell_bytes, AM).total;
                if (lds > 64 * 1024) continue;                                               // (a module kernel keeps the default LDS limit)
                je_ = c;
            }
            if (je_ && (!geom_given || cfg->threads_per_workgroup == 0 || cfg->threads_per_workgroup == 256)) {
                rw_jit::Shape sh{};
                sh.R = R; sh.H = H; sh.W = W; sh.N = N; sh.Q = Q; sh.S = S; sh.E = je_; sh.T = 256; sh.M = eng->msg_bits;
                sh.wide = eng->wide ? 1 : 0;
                sh.obs = eng->image ? rw::OBS_IMAGE : eng->msg_bits > 0 ? rw::OBS_FLATTENED_MSG : rw::OBS_FLATTENED;
                sh.NL = eng->image ? n_layers : 0;
                sh.directional = eng->image ? (cfg->image_directional ? 1 : 0) : -1;
                for (int l = 0; l < n_layers && l < 8 && eng->image; ++l) sh.layers |= (uint32_t)layers[l] << (4 * l);
                sh.nt = nt_rule(je_) ? 1 : 0;
                sh.stats = want_stats ? 1 : 0;
                // (two attempts: a CACHED code object the runtime refuses — a truncated or foreign file behind a well-formed header — is
                //  dropped from the cache and the shape compiled afresh, once; otherwise every later construction would trip over it)
                for (int attempt = 0; attempt < 2 && eng->jit_state != 1 && eng->jit_state != 2; ++attempt) {
                rw_jit::Result res;
                const bool built = rw_jit::compile(sh, eng->prop.gcnArchName, &res);
                eng->jit_log = (attempt ? eng->jit_log + " | retry: " : eng->jit_log.empty() ? std::string() : eng->jit_log + " | ") + res.log;
                eng->jit_state = -1;
                if (!built) break;
                {
                    hipError_t me = hipModuleLoadData(&eng->jit_module, res.code.data());
                    if (me == hipSuccess) me = hipModuleGetFunction(&eng->jit_step, eng->jit_module, res.step_name.c_str());
                    if (me == hipSuccess) me = hipModuleGetFunction(&eng->jit_rollout, eng->jit_module, res.rollout_name.c_str());
                    if (me == hipSuccess) {
                        E = je_;
                        T = 256;
                        roll_E = 0;
                        wide4 = false;
                        eng->specialised = true;
                        eng->q_runtime = false;
                        eng->build_kind = 1;
                        eng->jit_state = res.from_cache ? 2 : 1;
                        eng->kernel_nt = nullptr;
                        eng->jit_nt = sh.nt;
                    } else {
                        eng->jit_log += std::string(" | loading the code object failed: ") + hipGetErrorString(me);
                        eng->jit_step = eng->jit_rollout = nullptr;
                        if (eng->jit_module) { (void)hipModuleUnload(eng->jit_module); eng->jit_module = nullptr; }
                        (void)hipGetLastError();
                        if (!res.from_cache) break;   // (a fresh build that does not load: nothing a second compile would change)
                        const std::string stale = rw_jit::cache_file(sh, eng->prop.gcnArchName);
                        if (stale.empty() || unlink(stale.c_str()) != 0) break;
                        eng->jit_log += " | dropped the cached file";
                    }
                }
                }
            } else {
                eng->jit_log = "no workgroup geometry of an exact-shape build fits this batch / shape";
                eng->jit_state = -1;
            }
        }
    }
#endif
    eng->E = E;
    eng->T = T;
    eng->n_wg = (B + E - 1) / E;
    eng->roll_n_wg = roll_E ? (B + roll_E - 1) / roll_E : eng->n_wg;
    // (an agent-count-static build reserves 2 N queue slots per env in LDS whatever Q is)
    eng->lds_bytes = sizeof(int32_t) * (size_t)rw::make_lds_layout(E, N, eng->q_runtime ? 2 * N : Q, HW, SW, eng->OW, cell_bytes, AM).total;
    eng->roll_lds_bytes = roll_E ? sizeof(int32_t) * (size_t)rw::make_lds_layout(roll_E, N, eng->q_runtime ? 2 * N : Q, HW, SW, eng->OW, cell_bytes, AM).total
                                 : eng->lds_bytes;
    if (std::max(eng->lds_bytes, eng->roll_lds_bytes) > 160 * 1024) {
        fail(eng, RW_ERR_INVALID_ARG, "LDS footprint %zu B exceeds 160 KiB; lower envs_per_workgroup", eng->lds_bytes);
        return bail(RW_ERR_INVALID_ARG);
    }
    if (std::max(eng->lds_bytes, eng->roll_lds_bytes) > 64 * 1024) {
        hipError_t lds_err = hipFuncSetAttribute(reinterpret_cast<const void *>(eng->kernel),
                                                 hipFuncAttributeMaxDynamicSharedMemorySize, (int)eng->lds_bytes);
        if (lds_err == hipSuccess)
            lds_err = hipFuncSetAttribute(reinterpret_cast<const void *>(eng->kernel_rollout),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)eng->roll_lds_bytes);
        if (lds_err == hipSuccess && eng->kernel_nt)
            lds_err = hipFuncSetAttribute(reinterpret_cast<const void *>(eng->kernel_nt),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)eng->lds_bytes);
        RW_HIP_C(lds_err);
    }
    {
        // two or more rounds of workgroups (the GPU holds 8 four-wavefront workgroups per CU, fewer if LDS says so): stagger the start of
        // the first round (the kernel's prologue says why); k = blockIdx >> log2(CUs) needs a power-of-two CU count (MI355X: 256)
        const int n_cu = eng->prop.multiProcessorCount;
        const long long per_cu = std::min<long long>(8, (160 * 1024) / (long long)std::max<size_t>(eng->lds_bytes, 1));
        const bool pow2 = n_cu > 0 && (n_cu & (n_cu - 1)) == 0;
        while (pow2 && (1 << eng->stagger_shift) < n_cu) ++eng->stagger_shift;
        // Wavefront priority up to the agent-phase barrier (round 6, rware_kernels.h; same-box A/B of library variants and of this
        // switch: profiles/r06_prio_ab.txt, r06_prio_ab2.txt, r06_prio_wide.txt, r06_prio_sweep.txt).  On for every launch except the
        // 13 .. 16-agent ones at sensor_range 1 — the family that runs start-staggered: with the priority on top they lose (medium-13ag x
        // 16384 16.3 -> 17.7 us, small-15ag 19.3 -> 20.0, small-14ag x 65536 48.5 -> 52.2) — and except steps whose observations approach
        // the Infinity Cache size (200 MB of them and more: small-12ag x 65536 42.3 -> 42.9, small-19ag x 65536 81.2 -> 83.0, small-4ag x
        // 262144 58.9 -> 59.6), which do not move or lose a per cent or two.
        // The fused rollouts (every step of the launch raises the priority again): they gain at every agent count — small-4ag 3.95 ->
        // 3.72 us per step, medium-6ag-hard x 8192 4.20 -> 3.94, small-8ag 8.89 -> 8.33, medium-13ag 14.5 -> 13.85, large-16ag 21.75 -> 21.25
        // (profiles/r06_prio_rollout.txt) — so only the size limit applies to them.
        const bool fits = (double)B * N * eng->L * 4 <= 200e6;
        // (13 .. 16 agents — `wide4` below is only ever read for them: on their 4-env workgroups with the priority; on 8-env workgroups — the tiny warehouse has no 4-env build — with it
        //  only up to half a round of workgroups, where the stagger is a loss: 4096 envs large-16ag 9.52 us with the stagger, 9.00 without,
        //  8.80 with the priority instead; 8192: small-14ag 12.65 / 11.26 / 10.84; profiles/r06_1316_matrix.txt)
        const bool wide8_small = 2 * (long long)eng->n_wg <= per_cu * n_cu;
        eng->prio = fits && !(R == 1 && N >= 13 && N <= 16 && !wide4 && !wide8_small);
        eng->prio_rollout = fits;
        const char *pr = rw_hook("RWARE_PRIO");  // (A/B and test hooks: 0 = off, 1 = on whatever the shape; an explicit flag of the caller wins)
        if (pr && (pr[0] == '0' || pr[0] == '1')) eng->prio = pr[0] == '1';
        const char *prr = rw_hook("RWARE_PRIO_ROLLOUT");
        if (prr && (prr[0] == '0' || prr[0] == '1')) eng->prio_rollout = prr[0] == '1';
        if (cfg->stream_flags & RW_PRIO_OFF) eng->prio = eng->prio_rollout = false;  // (rware_amd.make_pipelines: two launches in flight)
        if (cfg->stream_flags & RW_PRIO_ON) eng->prio = eng->prio_rollout = true;
        // Start stagger.  Up to 12 agents (beyond, the workgroups are bound by their agent phases' instruction issue, there is little idle
        // phase to fill and the sweep is a wash — large-16ag -3 % at 4 rounds, +4 % at 2): 250 ns per slot from two rounds of workgroups
        // on (profiles/r04_stagger_sweep.txt) — WHERE THE PRIORITY IS OFF: with the chain at raised priority the rounds no longer run in
        // lock-step, and the delay is only a delay (profiles/r06_stagger_under_prio.txt, priority on, stagger 0 against 25: small-3ag x
        // 65536 13.9 against 15.05 us, small-5ag 21.5 / 22.6, small-7ag 26.9 / 28.7, small-6ag 21.85 / 22.7, medium-6ag-hard 22.3 / 22.85,
        // small-4ag 14.6 / 14.7; 8 .. 10 agents from four rounds on would keep a per cent or two of it: small-8ag x 65536 27.2 / 26.8).
        eng->stagger_ticks = (pow2 && N <= 12 && !eng->prio && (long long)eng->n_wg >= 2 * per_cu * n_cu) ? 25 : 0;  // x 10 ns per slot
        // 13 .. 16 agents (round 6, profiles/r06_stagger_single_round.txt, r06_t128_stagger.txt): these launches gain from WIDER slots, and
        // already when the launch is resident at once — eight workgroups per CU with two agent wavefronts each contend for the same
        // SIMDs in their agent phases, and 0.55 us between their starts takes the phases apart (16384 envs: medium-13ag 17.7 -> 16.4 us,
        // small-13ag 18.1 -> 17.0, small-15ag 20.4 -> 19.3, small-14ag 17.3 -> 17.0, large-16ag 18.0 -> 17.3; four rounds and more: small-14ag x
        // 65536 55.1 -> 49.5, large-16ag x 65536 61.1 -> 59.3).  Exactly two rounds lose (large-16ag x 32768 30.8 -> 32.0, medium-16ag x 32768
        // 31.2 -> 32.5): left alone.  9 .. 12 and 17 .. 19 agents lose or do not move at one round (small-10ag 13.1 -> 13.5, 19ag 21.2 -> 21.6).
        // sensor_range 2 (BASELINE config 5, 4-env workgroups, two rounds at 16384 envs): 40 ticks, 35.1 -> 34.5 — without the priority; with
        // it none (profiles/r06_cfg5_stagger_prio.txt: config 5's shard 33.9 against 34.3 with the 40 ticks, x 8192 19.56 / 19.66).
        if (pow2 && N >= 13 && N <= 16 && per_cu > 0) {
            const long long resident = per_cu * n_cu, wg = (long long)eng->n_wg;
            if (R == 1) eng->stagger_ticks = (!wide4 && ((2 * wg > resident && wg <= resident) || wg >= 4 * resident)) ? 55 : 0;
            else if (R == 2) eng->stagger_ticks = (!eng->prio && wg <= 2 * resident) ? 40 : 0;
        }
        const char *st = rw_hook("RWARE_STAGGER_TICKS");  // (A/B and test hook: 0 = off, n = ticks whatever the launch size)
        if (st && *st && pow2) eng->stagger_ticks = std::min(255, std::max(0, atoi(st)));
    }

    {
        // The chunk-pipelined persistent build (rware_kernels.h "PIPE").  Candidate: a `pipe` entry of the table with this shape whose
        // chunk size divides the batch (first match; RWARE_PIPE_E picks a geometry).  Whether it runs: the caller's RW_PIPE_ON /
        // RW_PIPE_OFF, else RWARE_PIPE=1|0 in the environment (A/B runs), else the measured rule below.
        int mode = 0;  // 0 rule, -1 off, 1 on
        const char *pe = rw_hook("RWARE_PIPE");
        if (pe && pe[0] == '0') mode = -1;
        if (pe && pe[0] == '1') mode = 1;
        if (cfg->stream_flags & RW_PIPE_OFF) mode = -1;
        if (cfg->stream_flags & RW_PIPE_ON) mode = 1;
        const char *pee = rw_hook("RWARE_PIPE_E");
        const int want_e = pee ? atoi(pee) : 0;
        const StaticEntry *pb = nullptr;
        if (mode >= 0 && !eng->image && eng->msg_bits == 0 && !eng->jit_step && !(want_stats && !rw_tab::static_has_stats()))
            for (int grp = 0; grp < rw_tab::kStaticGroups && !pb; ++grp) {
                int n_se = 0;
                const StaticEntry *tab = rw_tab::static_group(grp, &n_se);
                for (int k_se = 0; k_se < n_se && !pb; ++k_se) {
                    const StaticEntry &se = tab[k_se];
                    if (!se.pipe || se.H != H || se.W != W || se.S != S || se.R != R || se.N != N) continue;
                    if (!(se.Q == Q || (se.Q < 0 && Q <= 2 * se.N)) || B % se.E != 0 || (want_e && se.E != want_e)) continue;
                    pb = &se;
                }
            }
        if (pb) {
            const int n_cu = eng->prop.multiProcessorCount;
            const int n_chunks = B / pb->E;
            const size_t lds = 2 * sizeof(int32_t) * (size_t)rw::make_lds_layout(pb->E, N, pb->Q < 0 ? 2 * N : Q, HW, SW, eng->OW, cell_bytes, AM).total;
            step_kernel_t fn = nt_rule(pb->E) ? pb->fn_nt : pb->fn;
            hipError_t pe_ = lds > 64 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void *>(fn), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) : hipSuccess;
            int per_cu = 0;
            if (pe_ == hipSuccess) pe_ = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(fn), 256, lds);
            if (pe_ != hipSuccess || per_cu < 1) { (void)hipGetLastError(); per_cu = 0; }
            const char *pw = rw_hook("RWARE_PIPE_WGS_PER_CU");  // (A/B hook: fewer resident workgroups, more chunks each)
            if (pw && atoi(pw) > 0) per_cu = std::min(per_cu, atoi(pw));
            // (whole CUs' worth of workgroups: an "even" split — 1366 workgroups of 3 chunks instead of 1536 of 2 or 3 — loads the CUs
            //  unevenly and measured slower, small-4ag x 65536: 19.5 vs 17.5 us)
            long long grid = std::min<long long>(n_chunks, (long long)per_cu * n_cu);
            const char *pg = rw_hook("RWARE_PIPE_GRID");  // (test hook: a handful of workgroups, so that small batches walk several chunks each)
            if (pg && atoi(pg) > 0) grid = std::min<long long>(grid, atoi(pg));
            // The rule, as measured (profiles/EXPERIMENTS.md, round 5, profiles/r05_pipe_*.txt): NEVER by default.  In steady state the
            // pipelined workgroups run at the fabric's rate (small-4ag x 65536: 0.77 us per chunk and CU against the classic launch's
            // average of 0.95), but a launch is prologue + stream + tail, the classic launch's stream already runs at that rate once it
            // has started, and the persistent workgroups' prologue is longer (stage-in, then the agent phases of chunk 0, then its
            // gather: 4.6 us to the first store against 3.1) — every BASELINE config and batch size came out 8 .. 40 % slower.  The
            // build stays selectable (RW_PIPE_ON) and parity-tested.
            const bool rule = false;
            if (lds <= 160 * 1024 && grid >= 1 && (mode == 1 || (mode == 0 && rule && n_chunks >= 2 * grid))) {
                eng->pipe_kernel = fn;
                eng->pipe_E = pb->E;
                eng->pipe_grid = (int)grid;
                eng->pipe_lds = lds;
            }
        }
        if (mode == 1 && !eng->pipe_kernel) {  // asked for and not available: say so where rw_get_info().pipe_workgroups == 0 sends the caller (rw_jit_log)
            if (!eng->jit_log.empty()) eng->jit_log += " | ";
            eng->jit_log += "pipe: RW_PIPE_ON requested, the classic kernel runs: ";
#if RW_WITH_PIPE
            eng->jit_log += pb ? "the pipelined build of this shape does not fit (LDS / occupancy)"
                               : "no pipelined build for this shape (FLATTENED without messages, ahead-of-time builds, batch a multiple of its chunk size)";
#else
            eng->jit_log += "this library was built without the pipelined kernels (make PIPE=1)";
#endif
        }
    }

    // device buffers
    const size_t szB = (size_t)B;
    size_t n_elems[RW_BUF_KIND_COUNT];
    n_elems[RW_BUF_OBS] = szB * N * eng->L;
    n_elems[RW_BUF_REWARDS] = szB * N;
    n_elems[RW_BUF_TERMINATED] = szB;
    n_elems[RW_BUF_TRUNCATED] = szB;
    n_elems[RW_BUF_GRID] = szB * 2 * HW;
    for (int k = RW_BUF_AGENT_X; k <= RW_BUF_AGENT_DELIVERED; ++k) n_elems[k] = szB * N;
    n_elems[RW_BUF_QUEUE] = szB * (Q > 0 ? Q : 0);
    n_elems[RW_BUF_STEPS] = szB;
    n_elems[RW_BUF_INACTIVE] = szB;
    n_elems[RW_BUF_RNG] = szB * 6;
    n_elems[RW_BUF_NEED_RESET] = szB;
    n_elems[RW_BUF_ACTIONS] = szB * N * AM;
    n_elems[RW_BUF_FEATURES] = szB * N * 6;
    n_elems[RW_BUF_AGENT_MSG] = szB * N;
    // the terminal observation of SAME_STEP autoreset (+ its IMAGE_DICT feature vectors): allocated only where the kernel writes it
    const bool want_final = cfg->autoreset_mode == RW_AUTORESET_SAME_STEP;
    n_elems[RW_BUF_FINAL_OBS] = want_final ? szB * N * eng->L : 0;
    n_elems[RW_BUF_FINAL_FEATURES] = want_final && obs_type == RW_OBS_IMAGE_DICT ? szB * N * 6 : 0;
    // the event counters: allocated (and counted into) only on request
    eng->stats = want_stats;
    n_elems[RW_BUF_STAT_DELIVERIES] = n_elems[RW_BUF_STAT_FAILED_MOVES] = eng->stats ? szB : 0;
    // One slab for every buffer: the per-step working set (agent SoA, queue, counters, flags, rewards,
    // shelf shadow) sits in a few contiguous MiB, so a workgroup's ~15 streams share TLB entries
    // instead of touching 15 separate allocations.  Order = hot and small first.
    static const int order[RW_BUF_KIND_COUNT] = {
        RW_BUF_AGENT_X, RW_BUF_AGENT_Y, RW_BUF_AGENT_DIR, RW_BUF_AGENT_CARRY, RW_BUF_AGENT_DELIVERED, RW_BUF_QUEUE,
        RW_BUF_AGENT_MSG, RW_BUF_STEPS, RW_BUF_INACTIVE, RW_BUF_NEED_RESET, RW_BUF_REWARDS, RW_BUF_TERMINATED, RW_BUF_TRUNCATED,
        RW_BUF_STAT_DELIVERIES, RW_BUF_STAT_FAILED_MOVES, RW_BUF_ACTIONS, RW_BUF_RNG, RW_BUF_FEATURES, RW_BUF_OBS, RW_BUF_FINAL_OBS, RW_BUF_FINAL_FEATURES, RW_BUF_GRID};
    auto up = [](size_t x) { return (x + 4095) & ~(size_t)4095; };
    size_t slab_bytes = 0, off[RW_BUF_KIND_COUNT];
    eng->rec_off = 0;  // the packed agent records lead the hot set, the counter records follow
    slab_bytes += up(szB * N * sizeof(uint32_t));
    eng->cnt_off = slab_bytes;
    slab_bytes += up(szB * 2 * sizeof(int32_t) + 64);  // (+ the rounding pieces of the stage-in DMA)
    for (int k : order) {
        eng->buf[k].bytes = n_elems[k] * elem_size(k);
        off[k] = slab_bytes;
        slab_bytes += up(eng->buf[k].bytes ? eng->buf[k].bytes : 16);
        if (k == RW_BUF_ACTIONS) {  // the shelf shadow rides with the hot set
            eng->shadow_off = slab_bytes;
            slab_bytes += up(szB * HW * cell_bytes + 16);
        }
    }
    RW_HIP_C(hipMalloc(&eng->slab, slab_bytes));
    RW_HIP_C(hipMemsetAsync(eng->slab, 0, slab_bytes, eng->stream));
    for (int k = 0; k < RW_BUF_KIND_COUNT; ++k) eng->buf[k].ptr = (char *)eng->slab + off[k];
    eng->d_shadow = (char *)eng->slab + eng->shadow_off;
    eng->d_rec = (uint32_t *)((char *)eng->slab + eng->rec_off);
    eng->d_cnt = (int32_t *)((char *)eng->slab + eng->cnt_off);
    const int HWW = (HW + 31) / 32;
    // the static kernels stage the bitmap in whole 16-byte pieces: allocate (and zero) the rounded-up size
    const size_t hw_bytes = sizeof(uint32_t) * (size_t)rw::rw_up4(HWW);
    RW_HIP_C(hipMalloc(&eng->d_highway_bits, hw_bytes));
    RW_HIP_C(hipMemsetAsync(eng->d_highway_bits, 0, hw_bytes, eng->stream));
    RW_HIP_C(hipMalloc(&eng->d_shelf_init, sizeof(int32_t) * HW));
    RW_HIP_C(hipMalloc(&eng->d_mask, szB + 64));
    RW_HIP_C(hipMalloc(&eng->d_status, sizeof(int32_t)));
    RW_HIP_C(hipHostMalloc((void **)&eng->h_actions, n_elems[RW_BUF_ACTIONS] * sizeof(int32_t)));
    RW_HIP_C(hipEventCreate(&eng->h_actions_free));
    std::vector<int32_t> shelf_init(HW, 0);
    std::vector<uint32_t> hw_bits(HWW, 0u);
    for (int i = 0; i < HW; ++i)
        if (cfg->highways[i]) hw_bits[i >> 5] |= 1u << (i & 31);
    for (int i = 0, s = 0; i < HW; ++i)
        if (!cfg->highways[i]) shelf_init[i] = ++s;  // ids 1..S, row-major (rware/warehouse.py:771-778)
    RW_HIP_C(hipMemcpyAsync(eng->d_highway_bits, hw_bits.data(), sizeof(uint32_t) * HWW, hipMemcpyHostToDevice, eng->stream));
    RW_HIP_C(hipMemcpyAsync(eng->d_shelf_init, shelf_init.data(), sizeof(int32_t) * HW, hipMemcpyHostToDevice, eng->stream));
    RW_HIP_C(hipMemsetAsync(eng->d_status, 0, sizeof(int32_t), eng->stream));
    RW_HIP_C(hipStreamSynchronize(eng->stream));
#undef RW_HIP_C

    rw::Params &p = eng->prm;
    p.B = B; p.H = H; p.W = W; p.HW = HW; p.N = N; p.Q = Q; p.S = S; p.SW = SW;
    p.n_goals = cfg->n_goals;
    p.max_inactivity = cfg->max_inactivity_steps;
    p.max_steps = cfg->max_steps;
    p.reward_type = cfg->reward_type;
    p.autoreset = cfg->autoreset_mode;
    p.normalised = cfg->normalised_coordinates ? 1 : 0;
    p.envs_per_wg = E;
    {
        const bool nt = eng->jit_step ? eng->jit_nt != 0 : nt_rule(E);
        p.nt_obs = nt ? 1 : 0;
        if (nt && eng->kernel_nt) eng->kernel = eng->kernel_nt;  // (exact builds: the choice is a kernel, not a branch)
    }
    p.magic_n = rw::rw_magic18(N);
    p.groups_per_wave = 64 / N;
    p.HWW = HWW;
    for (int g = 0; g < rw::MAX_GOALS; ++g)
        p.goal_cells[g] = g < cfg->n_goals ? cfg->goals_xy[2 * g + 1] * W + cfg->goals_xy[2 * g] : 0;
    p.highway_bits = eng->d_highway_bits;
    p.shelf_init = eng->d_shelf_init;
    p.grid = (int32_t *)eng->buf[RW_BUF_GRID].ptr;
    p.arec = eng->d_rec;
    p.ax = (int32_t *)eng->buf[RW_BUF_AGENT_X].ptr;
    p.ay = (int32_t *)eng->buf[RW_BUF_AGENT_Y].ptr;
    p.adir = (int32_t *)eng->buf[RW_BUF_AGENT_DIR].ptr;
    p.acarry = (int32_t *)eng->buf[RW_BUF_AGENT_CARRY].ptr;
    p.adeliv = (int32_t *)eng->buf[RW_BUF_AGENT_DELIVERED].ptr;
    p.queue = (int32_t *)eng->buf[RW_BUF_QUEUE].ptr;
    p.counters = eng->d_cnt;
    p.steps = (int32_t *)eng->buf[RW_BUF_STEPS].ptr;
    p.inactive = (int32_t *)eng->buf[RW_BUF_INACTIVE].ptr;
    p.shelf_shadow = eng->d_shadow;
    p.rng = (uint64_t *)eng->buf[RW_BUF_RNG].ptr;
    p.need_reset = (uint8_t *)eng->buf[RW_BUF_NEED_RESET].ptr;
    p.truncated = (uint8_t *)eng->buf[RW_BUF_TRUNCATED].ptr;
    p.status = eng->d_status;
    p.n_layers = n_layers;
    p.directional = cfg->image_directional ? 1 : 0;
    p.transposed_layers = 0;
    if (eng->image)
        for (int l = 0; l < n_layers; ++l)
            p.transposed_layers |= (layers[l] == RW_LAYER_AGENT_DIRECTION ? 1 : 0) | (layers[l] == RW_LAYER_AGENT_LOAD ? 2 : 0);
    for (int l = 0; l < rw::MAX_IMAGE_LAYERS; ++l) p.layers[l] = l < n_layers ? layers[l] : 0;
    p.features = obs_type == RW_OBS_IMAGE_DICT ? (float *)eng->buf[RW_BUF_FEATURES].ptr : nullptr;
    p.msg_bits = cfg->msg_bits;
    p.amsg = (int32_t *)eng->buf[RW_BUF_AGENT_MSG].ptr;
    p.final_obs = want_final ? (float *)eng->buf[RW_BUF_FINAL_OBS].ptr : nullptr;
    p.final_features = n_elems[RW_BUF_FINAL_FEATURES] ? (float *)eng->buf[RW_BUF_FINAL_FEATURES].ptr : nullptr;
    p.stat_deliveries = eng->stats ? (int32_t *)eng->buf[RW_BUF_STAT_DELIVERIES].ptr : nullptr;
    p.stat_failed_moves = eng->stats ? (int32_t *)eng->buf[RW_BUF_STAT_FAILED_MOVES].ptr : nullptr;
    rw::LaunchArgs &la = eng->la;
    la.actions = (const int32_t *)eng->buf[RW_BUF_ACTIONS].ptr;
    la.reset_mask = eng->d_mask;
    la.obs = (float *)eng->buf[RW_BUF_OBS].ptr;
    la.rewards = (float *)eng->buf[RW_BUF_REWARDS].ptr;
    la.terminated = (uint8_t *)eng->buf[RW_BUF_TERMINATED].ptr;
    la.timeline = nullptr;
    la.op = rw::OP_STEP;
    la.n_steps = 1;
    la.act_stride = la.obs_stride = la.rew_stride = la.term_stride = 0;
    // the constant block goes to device memory once; launches pass only a pointer to it
    if (hipMalloc(&eng->d_prm, sizeof(rw::Params)) != hipSuccess ||
        hipMemcpy(eng->d_prm, &eng->prm, sizeof(rw::Params), hipMemcpyHostToDevice) != hipSuccess) {
        fail(eng, RW_ERR_HIP, "uploading the parameter block failed");
        g_create_error = eng->err;
        rw_destroy(eng);
        return RW_ERR_HIP;
    }
    *out = eng;
    return RW_OK;
}

int rw_destroy(rw_engine *eng) {
    if (!eng) return RW_OK;
    (void)hipSetDevice(eng->cfg.device_id);
    if (eng->stream) (void)hipStreamSynchronize(eng->stream);
    if (eng->jit_module) (void)hipModuleUnload(eng->jit_module);
    if (eng->slab) (void)hipFree(eng->slab);
    if (eng->d_highway_bits) (void)hipFree(eng->d_highway_bits);
    if (eng->d_shelf_init) (void)hipFree(eng->d_shelf_init);
    if (eng->d_mask) (void)hipFree(eng->d_mask);
    if (eng->d_prm) (void)hipFree(eng->d_prm);
    if (eng->d_status) (void)hipFree(eng->d_status);
    if (eng->h_actions) (void)hipHostFree(eng->h_actions);
    if (eng->h_actions_free) (void)hipEventDestroy(eng->h_actions_free);
    for (auto &ev : eng->events)
        if (ev) (void)hipEventDestroy(ev);
    if (eng->own_stream && eng->stream) (void)hipStreamDestroy(eng->stream);
    delete eng;
    return RW_OK;
}

int rw_reset(rw_engine *eng, const uint64_t *seeds, const uint8_t *mask) {
    if (!eng) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    const int B = eng->prm.B;
    if (seeds) {
        // reseed masked envs: SeedSequence -> PCG64 on the host, merged into the field-major RNG buffer
        std::vector<uint64_t> h((size_t)B * 6);
        RW_HIP(eng, hipMemcpyAsync(h.data(), eng->buf[RW_BUF_RNG].ptr, h.size() * 8, hipMemcpyDeviceToHost, eng->stream));
        RW_HIP(eng, hipStreamSynchronize(eng->stream));
        for (int e = 0; e < B; ++e) {
            if (mask && !mask[e]) continue;
            uint64_t st[6];
            rw_seed_state(seeds[e], st);
            for (int f = 0; f < 6; ++f) h[(size_t)f * B + e] = st[f];
        }
        RW_HIP(eng, hipMemcpyAsync(eng->buf[RW_BUF_RNG].ptr, h.data(), h.size() * 8, hipMemcpyHostToDevice, eng->stream));
        RW_HIP(eng, hipStreamSynchronize(eng->stream));
    }
    if (mask) {
        RW_HIP(eng, hipMemcpyAsync(eng->d_mask, mask, (size_t)B, hipMemcpyHostToDevice, eng->stream));
        RW_HIP(eng, hipStreamSynchronize(eng->stream));  // `mask` is caller-owned pageable memory
    } else {
        RW_HIP(eng, hipMemsetAsync(eng->d_mask, 1, (size_t)B, eng->stream));
    }
    return launch(eng, eng->la, rw::OP_RESET);
}

int rw_step_device(rw_engine *eng, const int32_t *actions_dev) {
    if (!eng || !actions_dev) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    rw::LaunchArgs la = eng->la;
    la.actions = actions_dev;
    return launch(eng, la, rw::OP_STEP);
}

namespace {
int step_tape(rw_engine *eng, const int32_t *tape_dev, int32_t tape_steps, int32_t first, int32_t n_steps,
              hipEvent_t start, hipEvent_t stop) {
    // n_steps consecutive rw_step_device launches from a device-resident action tape int32 [tape_steps][B][N][1+M]: step k
    // reads row (first + k) % tape_steps.  Same launches as n_steps calls of rw_step_device — one per step, each ordered
    // behind the previous one on the engine's stream — issued from one native loop instead of one host call each.
    if (!eng || !tape_dev || tape_steps < 1 || first < 0 || n_steps < 0) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    const size_t row = (size_t)eng->prm.B * eng->prm.N * (1 + eng->msg_bits);
    rw::LaunchArgs la = eng->la;
    for (int32_t k = 0; k < n_steps; ++k) {
        la.actions = tape_dev + (size_t)((first + k) % tape_steps) * row;
        const int rc = launch(eng, la, rw::OP_STEP, false, k == 0 ? start : nullptr, k == n_steps - 1 ? stop : nullptr);
        if (rc != RW_OK) return rc;
    }
    return RW_OK;
}
}  // namespace

int rw_step_tape_device(rw_engine *eng, const int32_t *tape_dev, int32_t tape_steps, int32_t first, int32_t n_steps) {
    return step_tape(eng, tape_dev, tape_steps, first, n_steps, nullptr, nullptr);
}

int rw_step_tape_device_timed(rw_engine *eng, const int32_t *tape_dev, int32_t tape_steps, int32_t first, int32_t n_steps,
                              int32_t start_slot, int32_t stop_slot) {
    if (!eng || start_slot < 0 || start_slot >= 8 || stop_slot < 0 || stop_slot >= 8 || start_slot == stop_slot || n_steps < 1)
        return RW_ERR_INVALID_ARG;
    return step_tape(eng, tape_dev, tape_steps, first, n_steps, eng->events[start_slot], eng->events[stop_slot]);
}

int rw_step(rw_engine *eng, const int32_t *actions_host) {
    if (!eng || !actions_host) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    // `actions_host` is caller-owned pageable memory that may be gone when this returns: stage it through the
    // engine's pinned buffer (re-used only once its previous copy has left it)
    RW_HIP(eng, hipEventSynchronize(eng->h_actions_free));
    memcpy(eng->h_actions, actions_host, eng->buf[RW_BUF_ACTIONS].bytes);
    RW_HIP(eng, hipMemcpyAsync(eng->buf[RW_BUF_ACTIONS].ptr, eng->h_actions, eng->buf[RW_BUF_ACTIONS].bytes,
                               hipMemcpyHostToDevice, eng->stream));
    RW_HIP(eng, hipEventRecord(eng->h_actions_free, eng->stream));
    return launch(eng, eng->la, rw::OP_STEP);
}

int rw_step_many_device(rw_engine *eng, const int32_t *actions_dev, int32_t n_steps, float *obs_tape,
                        float *reward_tape, uint8_t *terminated_tape) {
    if (!eng || !actions_dev || n_steps < 0) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    const size_t BN = (size_t)eng->prm.B * eng->prm.N;
    if (n_steps == 0) return RW_OK;
    // ONE launch: the kernel keeps each workgroup's env chunk in LDS across the n_steps steps
    rw::LaunchArgs la = eng->la;
    la.actions = actions_dev;
    la.n_steps = n_steps;
    la.act_stride = (int64_t)BN * (1 + eng->msg_bits);
    if (obs_tape) { la.obs = obs_tape; la.obs_stride = (int64_t)(BN * eng->L); }
    if (reward_tape) { la.rewards = reward_tape; la.rew_stride = (int64_t)BN; }
    if (terminated_tape) { la.terminated = terminated_tape; la.term_stride = (int64_t)eng->prm.B; }
    return launch(eng, la, rw::OP_STEP, /*rollout=*/true);
}

int rw_debug_timeline(rw_engine *eng, const int32_t *actions_dev, uint64_t *host_out, int32_t *n_workgroups, int32_t *n_marks) {
    // One OP_STEP launch with per-workgroup phase stamps (wall_clock64, 100 MHz); profiling aid only.
    if (!eng || !actions_dev) return RW_ERR_INVALID_ARG;
    const int n_launched = eng->pipe_kernel ? eng->pipe_grid : eng->n_wg;  // (the pipelined build: one row per persistent workgroup)
    if (n_workgroups) *n_workgroups = n_launched;
    if (n_marks) *n_marks = rw::TL_MARKS;
    if (!host_out) return RW_OK;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    const size_t bytes = sizeof(uint64_t) * (size_t)n_launched * rw::TL_MARKS;
    uint64_t *d = nullptr;
    RW_HIP(eng, hipMalloc(&d, bytes));
    RW_HIP(eng, hipMemsetAsync(d, 0, bytes, eng->stream));
    rw::LaunchArgs la = eng->la;
    la.actions = actions_dev;
    la.timeline = d;
    int rc = launch(eng, la, rw::OP_STEP);
    if (rc == RW_OK) {
        hipError_t e1 = hipMemcpyAsync(host_out, d, bytes, hipMemcpyDeviceToHost, eng->stream);
        hipError_t e2 = hipStreamSynchronize(eng->stream);
        if (e1 != hipSuccess || e2 != hipSuccess) rc = fail(eng, RW_ERR_HIP, "timeline copy failed");
    }
    (void)hipFree(d);
    return rc;
}

int rw_debug_store_floor(rw_engine *eng, int32_t n_launches, float *ms_per_launch) {
    // Measurement aid: n launches of a kernel that does NOTHING but write one step's observations — the engine's launch geometry,
    // the engine's store instruction (16 bytes per lane, non-temporal if the engine stores that way), back to back on the engine's
    // stream, HIP events on the first / last launch.  What any kernel that produces this step's observations pays at the least:
    // the practical floor beside the 8 TB/s one (bench.py `roofline.store_only_*`).  RW_BUF_OBS is refreshed afterwards.
    if (!eng || n_launches < 1 || !ms_per_launch) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    const int per_wg = eng->E * eng->prm.N * eng->L;  // floats
    float *obs = (float *)eng->buf[RW_BUF_OBS].ptr;
    const size_t total = (size_t)eng->prm.B * eng->prm.N * eng->L;  // (size_t, like the step kernel's offsets: past 2^31 floats for large batches)
    for (int k = 0; k < n_launches; ++k) {
        hipEvent_t a = k == 0 ? eng->events[6] : nullptr, b = k == n_launches - 1 ? eng->events[7] : nullptr;
        if (eng->prm.nt_obs)
            hipExtLaunchKernelGGL((rw::rware_store_floor_kernel<true>), dim3(eng->n_wg), dim3(eng->T), 0, eng->stream, a, b, 0, obs, per_wg, total);
        else
            hipExtLaunchKernelGGL((rw::rware_store_floor_kernel<false>), dim3(eng->n_wg), dim3(eng->T), 0, eng->stream, a, b, 0, obs, per_wg, total);
    }
    RW_HIP(eng, hipGetLastError());
    RW_HIP(eng, hipEventSynchronize(eng->events[7]));
    float ms = 0.0f;
    RW_HIP(eng, hipEventElapsedTime(&ms, eng->events[6], eng->events[7]));
    *ms_per_launch = ms / (float)n_launches;
    return launch(eng, eng->la, rw::OP_OBS);
}

int rw_device_malloc(rw_engine *eng, size_t bytes, void **dev_ptr) {
    if (!eng || !dev_ptr) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    RW_HIP(eng, hipMalloc(dev_ptr, bytes ? bytes : 16));
    return RW_OK;
}

int rw_device_free(rw_engine *eng, void *dev_ptr) {
    if (!eng) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    RW_HIP(eng, hipStreamSynchronize(eng->stream));
    if (dev_ptr) RW_HIP(eng, hipFree(dev_ptr));
    return RW_OK;
}

int rw_copy_to_device(rw_engine *eng, void *dev_dst, const void *host_src, size_t bytes) {
    if (!eng || (bytes && (!dev_dst || !host_src))) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    if (bytes) RW_HIP(eng, hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, eng->stream));
    RW_HIP(eng, hipStreamSynchronize(eng->stream));
    return RW_OK;
}

int rw_copy_to_host(rw_engine *eng, void *host_dst, const void *dev_src, size_t bytes) {
    if (!eng || (bytes && (!host_dst || !dev_src))) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    if (bytes) RW_HIP(eng, hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, eng->stream));
    RW_HIP(eng, hipStreamSynchronize(eng->stream));
    return RW_OK;
}

}  // extern "C"

struct rw_snapshot {
    void *mem = nullptr;
    size_t bytes = 0;
};

namespace {
// the state that reset()/step() evolve: (device pointer, size) pieces in a fixed order
std::vector<std::pair<void *, size_t>> state_pieces(rw_engine *eng) {
    static const int kinds[] = {RW_BUF_QUEUE, RW_BUF_RNG, RW_BUF_AGENT_MSG, RW_BUF_STAT_DELIVERIES, RW_BUF_STAT_FAILED_MOVES};  // (the last two: empty without RW_STATS_ON)
    std::vector<std::pair<void *, size_t>> v;
    v.emplace_back(eng->d_rec, (size_t)eng->prm.B * eng->prm.N * sizeof(uint32_t));  // the agents: their packed records
    v.emplace_back(eng->d_cnt, (size_t)eng->prm.B * 2 * sizeof(int32_t));            // steps, inactive, pending resets: the counter records
    for (int k : kinds) v.emplace_back(eng->buf[k].ptr, eng->buf[k].bytes);
    v.emplace_back(eng->d_shadow, (size_t)eng->prm.B * eng->prm.HW * (eng->wide ? 2 : 1));
    return v;
}
}  // namespace

extern "C" {

int rw_snapshot_create(rw_engine *eng, rw_snapshot **out) {
    if (!eng || !out) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    rw_snapshot *s = new (std::nothrow) rw_snapshot();
    if (!s) return fail(eng, RW_ERR_HIP, "out of host memory");
    for (auto &pc : state_pieces(eng)) s->bytes += (pc.second + 255) & ~(size_t)255;
    if (hipMalloc(&s->mem, s->bytes) != hipSuccess) {
        delete s;
        return fail(eng, RW_ERR_HIP, "hipMalloc of a %zu-byte snapshot failed", s->bytes);
    }
    *out = s;
    return RW_OK;
}

int rw_snapshot_save(rw_engine *eng, rw_snapshot *snap) {
    if (!eng || !snap || !snap->mem) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    size_t off = 0;
    for (auto &pc : state_pieces(eng)) {
        if (pc.second) RW_HIP(eng, hipMemcpyAsync((char *)snap->mem + off, pc.first, pc.second, hipMemcpyDeviceToDevice, eng->stream));
        off += (pc.second + 255) & ~(size_t)255;
    }
    return RW_OK;
}

int rw_snapshot_restore(rw_engine *eng, const rw_snapshot *snap) {
    if (!eng || !snap || !snap->mem) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    size_t off = 0;
    for (auto &pc : state_pieces(eng)) {
        if (pc.second) RW_HIP(eng, hipMemcpyAsync(pc.first, (const char *)snap->mem + off, pc.second, hipMemcpyDeviceToDevice, eng->stream));
        off += (pc.second + 255) & ~(size_t)255;
    }
    eng->grid_stale = eng->agents_stale = eng->counters_stale = true;  // (the views are not part of a snapshot: they are derived from what is)
    return launch(eng, eng->la, rw::OP_OBS);
}

int rw_snapshot_destroy(rw_engine *eng, rw_snapshot *snap) {
    if (!eng) return RW_ERR_INVALID_ARG;
    if (!snap) return RW_OK;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    RW_HIP(eng, hipStreamSynchronize(eng->stream));
    if (snap->mem) (void)hipFree(snap->mem);
    delete snap;
    return RW_OK;
}

int rw_refresh_obs(rw_engine *eng) {
    if (!eng) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    return launch(eng, eng->la, rw::OP_OBS);
}

int rw_sync(rw_engine *eng) {
    if (!eng) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    int32_t st = 0;
    RW_HIP(eng, hipMemcpyAsync(&st, eng->d_status, sizeof st, hipMemcpyDeviceToHost, eng->stream));
    RW_HIP(eng, hipStreamSynchronize(eng->stream));
    if (st) {
        RW_HIP(eng, hipMemsetAsync(eng->d_status, 0, sizeof st, eng->stream));
        RW_HIP(eng, hipStreamSynchronize(eng->stream));
        if (st & rw::STATUS_INVALID_ACTION)
            return fail(eng, RW_ERR_INVALID_ACTION, "an action outside 0..4 was submitted (executed as NOOP)");
        if (st & rw::STATUS_IMAGE_INDEX)
            return fail(eng, RW_ERR_INDEX, "AGENT_DIRECTION / AGENT_LOAD image layer: an agent at x >= grid height or y >= grid "
                        "width; the reference raises IndexError there (rware/warehouse.py:552,558)");
    }
    return RW_OK;
}

const char *rw_jit_log(const rw_engine *eng) { return eng ? eng->jit_log.c_str() : ""; }

int64_t rw_jit_probe(const int32_t shape[15], const char *arch, char *log, size_t log_len) {
    // Compiles (or finds in the disk cache) the exact-shape build of `shape` for `arch` WITHOUT a device: the compile half of
    // what rw_create does for a shape with no ahead-of-time build.  Returns the size of the code object, or -1.
    if (!shape || !arch) return -1;
#ifndef RW_NO_JIT
    rw_jit::Shape sh{};
    sh.R = shape[0]; sh.H = shape[1]; sh.W = shape[2]; sh.N = shape[3]; sh.Q = shape[4]; sh.S = shape[5]; sh.E = shape[6]; sh.T = shape[7];
    sh.M = shape[8]; sh.wide = shape[9]; sh.obs = shape[10]; sh.NL = shape[11]; sh.layers = (uint32_t)shape[12]; sh.directional = shape[13];
    sh.nt = shape[14];
    rw_jit::Result res;
    const bool ok = rw_jit::compile(sh, arch, &res);
    if (log && log_len) snprintf(log, log_len, "%s", res.log.c_str());
    return ok ? (int64_t)res.code.size() : -1;
#else
    if (log && log_len) snprintf(log, log_len, "built without run-time specialisation");
    return -1;
#endif
}

int rw_set_stream(rw_engine *eng, void *stream) {
    // everything enqueued so far stays ordered on the old stream; the caller orders the two streams (events) if it has to
    if (!eng) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    if (eng->own_stream && eng->stream) {
        RW_HIP(eng, hipStreamSynchronize(eng->stream));
        RW_HIP(eng, hipStreamDestroy(eng->stream));
        eng->own_stream = false;
    }
    eng->stream = (hipStream_t)stream;
    return RW_OK;
}

int rw_mark_views_stale(rw_engine *eng) {
    if (!eng) return RW_ERR_INVALID_ARG;
    eng->grid_stale = eng->agents_stale = eng->counters_stale = true;  // (every derived view: grid, the five agent arrays, steps / inactive / need_reset)
    return RW_OK;
}

int rw_refresh_grid(rw_engine *eng) {
    if (!eng) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    int rc = refresh_agents(eng);  // (all derived views: the five agent arrays and the counter views as well)
    if (rc == RW_OK) rc = refresh_counters(eng);
    return rc != RW_OK ? rc : refresh_grid(eng);
}

int rw_get_buffer(rw_engine *eng, int kind, void **dev_ptr, size_t *bytes) {
    if (!eng || kind < 0 || kind >= RW_BUF_KIND_COUNT) return RW_ERR_INVALID_ARG;
    if (kind == RW_BUF_GRID || is_agent_view(kind) || is_counter_view(kind)) {
        RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
        const int rc = kind == RW_BUF_GRID ? refresh_grid(eng) : is_agent_view(kind) ? refresh_agents(eng) : refresh_counters(eng);
        if (rc != RW_OK) return rc;
    }
    if (dev_ptr) *dev_ptr = eng->buf[kind].ptr;
    if (bytes) *bytes = eng->buf[kind].bytes;
    return RW_OK;
}

int rw_read(rw_engine *eng, int kind, void *host_dst, size_t bytes) {
    if (!eng || kind < 0 || kind >= RW_BUF_KIND_COUNT || (!host_dst && bytes)) return RW_ERR_INVALID_ARG;
    if (bytes != eng->buf[kind].bytes)
        return fail(eng, RW_ERR_INVALID_ARG, "rw_read kind %d: %zu bytes given, buffer holds %zu", kind, bytes, eng->buf[kind].bytes);
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    if (kind == RW_BUF_GRID || is_agent_view(kind) || is_counter_view(kind)) {
        const int rc = kind == RW_BUF_GRID ? refresh_grid(eng) : is_agent_view(kind) ? refresh_agents(eng) : refresh_counters(eng);
        if (rc != RW_OK) return rc;
    }
    if (bytes) RW_HIP(eng, hipMemcpyAsync(host_dst, eng->buf[kind].ptr, bytes, hipMemcpyDeviceToHost, eng->stream));
    RW_HIP(eng, hipStreamSynchronize(eng->stream));
    return RW_OK;
}

int rw_read_outputs(rw_engine *eng, float *obs, float *rewards, uint8_t *terminated, float *features) {
    // what step() returns, in ONE round trip: the copies are enqueued back to back and waited for once (four rw_read calls
    // are four synchronisations — 137 us per step at B = 1024 where the kernel takes 5)
    if (!eng) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    const struct { void *dst; int kind; } parts[] = {{obs, RW_BUF_OBS}, {rewards, RW_BUF_REWARDS}, {terminated, RW_BUF_TERMINATED},
                                                     {features, RW_BUF_FEATURES}};
    for (const auto &pt : parts)
        if (pt.dst && eng->buf[pt.kind].bytes)
            RW_HIP(eng, hipMemcpyAsync(pt.dst, eng->buf[pt.kind].ptr, eng->buf[pt.kind].bytes, hipMemcpyDeviceToHost, eng->stream));
    RW_HIP(eng, hipStreamSynchronize(eng->stream));
    return RW_OK;
}

int rw_write(rw_engine *eng, int kind, const void *host_src, size_t bytes) {
    if (!eng || kind < 0 || kind >= RW_BUF_KIND_COUNT || (!host_src && bytes)) return RW_ERR_INVALID_ARG;
    if (bytes != eng->buf[kind].bytes)
        return fail(eng, RW_ERR_INVALID_ARG, "rw_write kind %d: %zu bytes given, buffer holds %zu", kind, bytes, eng->buf[kind].bytes);
    // engine-owned flags: the step kernel stores them only when they change (truncated never does, :942), so a host write
    // would stick for the engine's lifetime — refused rather than silently different from the reference
    if (kind == RW_BUF_TRUNCATED)
        return fail(eng, RW_ERR_INVALID_ARG, "rw_write: RW_BUF_TRUNCATED is read-only (the reference never truncates, rware/warehouse.py:942)");
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    if (is_agent_view(kind) || is_counter_view(kind)) {  // the sibling views have to be current before they are all packed again
        const int rc = is_agent_view(kind) ? refresh_agents(eng) : refresh_counters(eng);
        if (rc != RW_OK) return rc;
    }
    if (bytes) RW_HIP(eng, hipMemcpyAsync(eng->buf[kind].ptr, host_src, bytes, hipMemcpyHostToDevice, eng->stream));
    if (is_agent_view(kind) || is_counter_view(kind)) {
        const int rc = is_agent_view(kind) ? pack_agents(eng) : pack_counters(eng);
        if (rc != RW_OK) return rc;
    }
    if (kind == RW_BUF_GRID) {
        const int rc = rebuild_shadow(eng);
        if (rc != RW_OK) return rc;
        eng->grid_stale = false;  // the caller's grid is the state now
    } else if (kind == RW_BUF_AGENT_X || kind == RW_BUF_AGENT_Y) {
        eng->grid_stale = true;   // layer 0 of the derived int32 grid follows the coordinates: rebuilt on the next read
    }
    RW_HIP(eng, hipStreamSynchronize(eng->stream));
    return RW_OK;
}

int rw_recalc_grid(rw_engine *eng, const int32_t *shelf_xy, int32_t n_shelves) {
    // Host-side (state injection is a test/debug path, not the hot path): exactly _recalc_grid (:749-755).
    if (!eng || !shelf_xy || n_shelves < 0) return RW_ERR_INVALID_ARG;
    const int B = eng->prm.B, N = eng->prm.N, HW = eng->prm.HW, W = eng->prm.W, H = eng->prm.H;
    std::vector<int32_t> ax((size_t)B * N), ay((size_t)B * N), grid((size_t)B * 2 * HW, 0);
    int rc = rw_read(eng, RW_BUF_AGENT_X, ax.data(), ax.size() * 4);
    if (rc) return rc;
    rc = rw_read(eng, RW_BUF_AGENT_Y, ay.data(), ay.size() * 4);
    if (rc) return rc;
    for (int e = 0; e < B; ++e) {
        int32_t *g = grid.data() + (size_t)e * 2 * HW;
        const int32_t *sx = shelf_xy + (size_t)e * n_shelves * 2;
        for (int k = 0; k < n_shelves; ++k) {
            const int x = sx[2 * k], y = sx[2 * k + 1];
            if (x < 0 || x >= W || y < 0 || y >= H) return fail(eng, RW_ERR_INVALID_ARG, "shelf %d of env %d outside the grid", k + 1, e);
            g[HW + y * W + x] = k + 1;
        }
        for (int i = 0; i < N; ++i) {
            const int x = ax[(size_t)e * N + i], y = ay[(size_t)e * N + i];
            if (x < 0 || x >= W || y < 0 || y >= H) return fail(eng, RW_ERR_INVALID_ARG, "agent %d of env %d outside the grid", i + 1, e);
            g[y * W + x] = i + 1;
        }
    }
    return rw_write(eng, RW_BUF_GRID, grid.data(), grid.size() * 4);
}

int rw_get_info(const rw_engine *eng, rw_info *out) {
    if (!eng || !out) return RW_ERR_INVALID_ARG;
    memset(out, 0, sizeof *out);
    const rw::Params &p = eng->prm;
    out->num_envs = p.B; out->grid_h = p.H; out->grid_w = p.W; out->n_agents = p.N;
    out->request_queue_size = p.Q; out->n_shelves = p.S; out->obs_length = eng->L;
    out->envs_per_workgroup = eng->E; out->threads_per_workgroup = eng->T; out->n_workgroups = eng->n_wg;
    out->lds_bytes = (int32_t)eng->lds_bytes;
    out->device_id = eng->cfg.device_id;
    out->compute_units = eng->prop.multiProcessorCount;
    out->stagger_ticks = eng->stagger_ticks;
    out->pipe_envs_per_workgroup = eng->pipe_kernel ? eng->pipe_E : 0;
    out->pipe_workgroups = eng->pipe_kernel ? eng->pipe_grid : 0;
    out->stats = eng->stats ? 1 : 0;
    out->wave_priority = (eng->prio ? 1 : 0) | (eng->prio_rollout ? 2 : 0);
    out->specialised = eng->specialised ? 1 : 0;
    out->build_kind = eng->build_kind;
    out->obs_stores_stream = p.nt_obs;
    out->jit = eng->jit_state;
    // A = 8HW + 4N + 40N + 4Q + 16 + 4NL + 4N + 4
    out->algorithmic_bytes_per_env_step =
        8LL * p.HW + 4LL * p.N + 40LL * p.N + 4LL * p.Q + 16 + 4LL * p.N * eng->L + 4LL * p.N + 4;
    // What THIS layout has to move per env-step: the shelf shadow (read), the packed agent records (read +
    // write), the actions, the request queue (read; written only on a delivery), the counter record — steps, inactive, the
    // pending-reset bit: 8 bytes (read + write) —, the observation, the rewards, `terminated`; with communication bits the stored messages
    // (read + write); IMAGE_DICT: the feature vectors.  The physical (PMC) traffic of a step is checked against this figure
    // (profiles/tools/sweep_collect.py), and bench.py's `frac_engine` is priced on it: a fraction of a bandwidth, never above 1.
    out->engine_bytes_per_env_step =
        (int64_t)p.HW * (eng->wide ? 2 : 1) + 8LL * p.N + 4LL * p.N * (1 + eng->msg_bits) + 4LL * p.Q + 16 + 4LL * p.N * eng->L +
        4LL * p.N + 1 + (eng->msg_bits ? 8LL * p.N : 0) + (p.features ? 24LL * p.N : 0);
    snprintf(out->device_name, sizeof out->device_name, "%s", eng->prop.name);
    snprintf(out->arch_name, sizeof out->arch_name, "%s", eng->prop.gcnArchName);
    return RW_OK;
}

int rw_event_record(rw_engine *eng, int32_t slot) {
    if (!eng || slot < 0 || slot >= 8) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    RW_HIP(eng, hipEventRecord(eng->events[slot], eng->stream));
    return RW_OK;
}

int rw_event_elapsed_ms(rw_engine *eng, int32_t a, int32_t b, float *ms) {
    if (!eng || !ms || a < 0 || a >= 8 || b < 0 || b >= 8) return RW_ERR_INVALID_ARG;
    RW_HIP(eng, hipSetDevice(eng->cfg.device_id));
    RW_HIP(eng, hipEventSynchronize(eng->events[b]));
    RW_HIP(eng, hipEventElapsedTime(ms, eng->events[a], eng->events[b]));
    return RW_OK;
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------------------------------------
// rw_multi: ONE call that enqueues a step on every engine of a single-process multi-device env ("use one
// thread per device or a single C call that fans out").  Every engine but the first has a launcher thread of its own, bound to
// that engine's device; rw_multi_step_device hands the action pointers over, launches engine 0 itself and returns when every
// launch has been enqueued (nothing waits for the GPUs).  A launcher spins on the round counter for a few tens of microseconds
// after each round — a training loop calls every ~10 us, so it stays hot — and sleeps on a condition variable otherwise.
struct rw_multi {
    std::vector<rw_engine *> engs;
    std::vector<const int32_t *> actions;
    std::vector<int> rc;
    std::vector<std::thread> threads;
    std::atomic<uint64_t> round{0};
    std::atomic<int> pending{0};
    std::atomic<bool> stop{false};
    std::atomic<int> sleepers{0};
    std::mutex mu;
    std::condition_variable cv;
};

namespace {
void multi_worker(rw_multi *m, int k) {
    (void)hipSetDevice(m->engs[(size_t)k]->cfg.device_id);
    uint64_t seen = 0;
    for (;;) {
        int spins = 0;
        while (m->round.load(std::memory_order_acquire) == seen && !m->stop.load(std::memory_order_acquire)) {
            if (++spins < 20000) {
#if defined(__x86_64__)
                __builtin_ia32_pause();
#endif
                continue;
            }
            // (sleepers and round are a store-then-load pair on both sides — this thread: sleepers, then round; the caller: round, then
            //  sleepers — which needs sequential consistency to rule out "both read the old value" on weakly ordered CPUs)
            std::unique_lock<std::mutex> lk(m->mu);
            m->sleepers.fetch_add(1, std::memory_order_seq_cst);
            m->cv.wait(lk, [&] { return m->round.load(std::memory_order_seq_cst) != seen || m->stop.load(std::memory_order_seq_cst); });
            m->sleepers.fetch_sub(1, std::memory_order_seq_cst);
        }
        if (m->stop.load(std::memory_order_acquire)) return;
        seen = m->round.load(std::memory_order_acquire);
        rw_engine *eng = m->engs[(size_t)k];
        rw::LaunchArgs la = eng->la;
        la.actions = m->actions[(size_t)k];
        m->rc[(size_t)k] = la.actions ? launch(eng, la, rw::OP_STEP) : RW_ERR_INVALID_ARG;
        m->pending.fetch_sub(1, std::memory_order_acq_rel);
    }
}
}  // namespace

extern "C" {

int rw_multi_create(rw_engine **engines, int32_t n, rw_multi **out) {
    if (!engines || n < 1 || !out) return RW_ERR_INVALID_ARG;
    for (int k = 0; k < n; ++k)
        if (!engines[k]) return RW_ERR_INVALID_ARG;
    rw_multi *m = new (std::nothrow) rw_multi();
    if (!m) return RW_ERR_HIP;
    m->engs.assign(engines, engines + n);
    m->actions.assign((size_t)n, nullptr);
    m->rc.assign((size_t)n, RW_OK);
    // Launcher threads only pay when every engine has a device (and therefore a submission queue) of its own; engines that
    // share a device serialise on its queue anyway and the hand-off costs more than it hides (measured, 8 engines on one
    // MI355X: 44.6 us per round with threads, 30.9 us for eight Python calls) — those are looped over by the caller's thread.
    bool distinct = true;
    for (int a = 0; a < n && distinct; ++a)
        for (int b = a + 1; b < n; ++b)
            if (engines[a]->cfg.device_id == engines[b]->cfg.device_id) { distinct = false; break; }
    const char *force = rw_hook("RWARE_MULTI_THREADS");  // (1 / 0: force either mode — tests)
    if (force && (force[0] == '0' || force[0] == '1')) distinct = force[0] == '1';
    if (distinct)
        for (int k = 1; k < n; ++k) m->threads.emplace_back(multi_worker, m, k);
    *out = m;
    return RW_OK;
}

int rw_multi_step_device(rw_multi *m, const int32_t *const *actions_dev) {
    if (!m || !actions_dev) return RW_ERR_INVALID_ARG;
    const int n = (int)m->engs.size();
    if (m->threads.empty()) {  // engines that share devices: one loop, this thread
        // (whatever fails, and on whichever engine: the message goes to engine 0, the one callers of this entry point read)
        rw_engine *e0 = m->engs[0];
        for (int k = 0; k < n; ++k) {
            rw_engine *e = m->engs[(size_t)k];
            if (!actions_dev[k]) return fail(e0, RW_ERR_INVALID_ARG, "rw_multi_step_device: no action array for engine %d", k);
            int rc1 = RW_OK;
            const hipError_t he = hipSetDevice(e->cfg.device_id);
            if (he != hipSuccess) {
                rc1 = fail(e, RW_ERR_HIP, "hipSetDevice(%d): %s", e->cfg.device_id, hipGetErrorString(he));
            } else {
                rw::LaunchArgs la = e->la;
                la.actions = actions_dev[k];
                rc1 = launch(e, la, rw::OP_STEP);
            }
            if (rc1 != RW_OK) return k == 0 ? rc1 : fail(e0, rc1, "rw_multi_step_device: engine %d: %s", k, e->err.c_str());
        }
        return RW_OK;
    }
    for (int k = 0; k < n; ++k) m->actions[(size_t)k] = actions_dev[k];
    m->pending.store(n - 1, std::memory_order_release);
    m->round.fetch_add(1, std::memory_order_seq_cst);
    if (m->sleepers.load(std::memory_order_seq_cst) > 0) {
        std::lock_guard<std::mutex> lk(m->mu);
        m->cv.notify_all();
    }
    rw_engine *e0 = m->engs[0];
    int rc;
    if (!actions_dev[0]) {
        rc = fail(e0, RW_ERR_INVALID_ARG, "rw_multi_step_device: no action array for engine 0");
    } else if (const hipError_t he = hipSetDevice(e0->cfg.device_id); he != hipSuccess) {
        rc = fail(e0, RW_ERR_HIP, "hipSetDevice(%d): %s", e0->cfg.device_id, hipGetErrorString(he));
    } else {
        rw::LaunchArgs la = e0->la;
        la.actions = actions_dev[0];
        rc = launch(e0, la, rw::OP_STEP);
    }
    while (m->pending.load(std::memory_order_acquire) > 0) {
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    for (int k = 1; k < n && rc == RW_OK; ++k)
        if ((rc = m->rc[(size_t)k]) != RW_OK)  // which engine, and why: into engine 0's message, the one callers of this entry point read
            fail(e0, rc, "rw_multi_step_device: engine %d: %s", k, m->engs[(size_t)k]->err.c_str());
    return rc;
}

int rw_multi_destroy(rw_multi *m) {
    if (!m) return RW_OK;
    m->stop.store(true, std::memory_order_release);
    {
        std::lock_guard<std::mutex> lk(m->mu);
        m->cv.notify_all();
    }
    for (auto &t : m->threads) t.join();
    delete m;
    return RW_OK;
}

}  // extern "C"
