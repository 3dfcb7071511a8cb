// rware_kernels.h — CDNA4 (gfx950) device code of the vectorised RWARE step engine.
//
// One fused kernel advances `envs_per_wg` independent warehouses per workgroup through every
// phase of rware.warehouse.Warehouse.step (rware/warehouse.py:804-946) and the FLATTENED
// observation gather (:598-674), with the whole per-env state staged in LDS:
//
//   P0  coalesced dwordx4 loads of the workgroup's env chunk (grid, agent SoA, queue, actions)
//   P1  move intent + shelf-block cancel                       (:825-846, Agent.req_location :102-116)
//   P2  collision resolution in closed form — no graph library (:848-876; see resolve notes)
//   P3  apply (move / turn / load / unload), incremental grid update instead of _recalc_grid
//                                                              (:880-901, :749-755)
//   P5  goals, request replacement (numpy-exact PCG64 draw), rewards, termination (:903-942)
//   RS  on-device reset for autoreset / rw_reset, numpy-exact draws      (:757-802)
//   P7  observation: per (agent, window cell) 7-bit codes OR-ed into an L-bit string in LDS,
//       expanded to float32 and written with coalesced dwordx4 stores     (:598-674)
//   ST  coalesced write-back of the agent SoA / queue / counters; the int32 grid in HBM is
//       patched only at the <= 2N cells per layer that changed.
//
// HBM layout (env-major, see include/rware_hip.h): grid int32 [B][2][H][W]; agent fields int32
// [B][N] x5; queue int32 [B][Q]; counters int32 [B]; PCG64 state uint64 [6][B] (field-major so
// that a mass reset reads it coalesced); obs float32 [B][N][L].
//
// Roofline: integer/indexing work, no MFMA; bound by HBM bytes.  Algorithmic bytes per
// env-step A = 8HW + 4N + 40N + 4Q + 16 + 4NL + 4N + 4.
//
// Collision resolution, closed form.  The reference builds a digraph on cells with one
// out-edge per agent (start -> target) and, per weakly connected component, commits either
// the agents on its cycle (but nobody if the cycle is a 2-swap) or the agents on the longest
// path.  Because every node has out-degree <= 1, each component is an in-tree draining into
// one sink or one cycle, so with nxt(i) = the agent standing on i's target cell:
//   - i stationary (target == start, incl. wall-clamped FORWARD and cancelled moves): commits.
//   - depth(i) = longest chain of movers following i (atomicMax walk, <= N hops).
//   - win(i)   = i has the largest (depth, then LOWEST id) among movers with the same target.
//   - walk i -> nxt(i) -> ...: reaches an empty cell  => commit iff every agent on the walk wins;
//                              reaches a stationary agent => fail;
//                              returns to i after len hops => commit iff len >= 3 (cycle);
//                              N hops without either       => i feeds a cycle => fail.
// The tie rule (lowest agent id among equal depths) is the pinned rule of DESIGN.md §tie-break.
#pragma once
#include <stdint.h>

#include "rware_pcg64.h"

namespace rw {

enum : int { OP_STEP = 0, OP_RESET = 1, OP_OBS = 2 };
enum : int { ACT_NOOP = 0, ACT_FORWARD = 1, ACT_LEFT = 2, ACT_RIGHT = 3, ACT_TOGGLE = 4 };
enum : int { DIR_UP = 0, DIR_DOWN = 1, DIR_LEFT = 2, DIR_RIGHT = 3 };
enum : int { REW_GLOBAL = 0, REW_INDIVIDUAL = 1, REW_TWO_STAGE = 2 };
enum : int { AR_DISABLED = 0, AR_NEXT_STEP = 1, AR_SAME_STEP = 2 };
enum : int { STATUS_INVALID_ACTION = 1 };

struct Params {
    // config
    int32_t B, H, W, HW, N, Q, S, SW;  // SW = dwords of the requested-shelf bitmap = (S+32)/32
    int32_t n_goals, max_inactivity, max_steps, reward_type, autoreset, normalised;
    int32_t envs_per_wg;
    // static per config (device)
    const uint8_t *highways;    // [HW]
    const int32_t *goal_cells;  // [n_goals] cell index y*W+x, list order
    const int32_t *shelf_init;  // [HW] shelf layer right after reset: ids 1..S row-major on non-highway cells
    // state (device)
    int32_t *grid, *ax, *ay, *adir, *acarry, *adeliv, *queue, *steps, *inactive;
    uint64_t *rng;        // [6][B]
    uint8_t *need_reset;  // [B]
    // per-launch io
    const int32_t *actions;     // [B][N]           (OP_STEP)
    const uint8_t *reset_mask;  // [B] or nullptr   (OP_RESET)
    float *obs;                 // [B][N][L]
    float *rewards;             // [B][N]
    uint8_t *terminated, *truncated;  // [B]
    int32_t *status;            // [1] sticky error bits
};

// LDS carve-up, in dwords.  Every sub-array starts on a 16-byte boundary.
struct LdsLayout {
    int grid, ax, ay, dir, carry, deliv, act, start, tgt, nxt, depth, win, rew, queue, req, obits, envi, misc, total;
};
enum : int { ENVI_STEPS = 0, ENVI_INACTIVE = 1, ENVI_RESET = 2, ENVI_DONE = 3, ENVI_SKIP = 4, ENVI_W = 8 };

RW_HD int rw_up4(int x) { return (x + 3) & ~3; }

RW_HD LdsLayout make_lds_layout(int E, int N, int Q, int HW, int SW, int OW) {
    LdsLayout l;
    int o = 0;
    l.grid = o;  o += rw_up4(E * 2 * HW);
    const int en = rw_up4(E * N);
    l.ax = o;    o += en;
    l.ay = o;    o += en;
    l.dir = o;   o += en;
    l.carry = o; o += en;
    l.deliv = o; o += en;
    l.act = o;   o += en;
    l.start = o; o += en;
    l.tgt = o;   o += en;
    l.nxt = o;   o += en;
    l.depth = o; o += en;
    l.win = o;   o += en;
    l.rew = o;   o += en;
    l.queue = o; o += rw_up4(E * Q);
    l.req = o;   o += rw_up4(E * SW);
    l.obits = o; o += rw_up4(E * N * OW);
    l.envi = o;  o += rw_up4(E * ENVI_W);
    l.misc = o;  o += 4;
    l.total = o;
    return l;
}

// flat dword copy global -> LDS, dwordx4 when both sides are 16-byte aligned
__device__ __forceinline__ void copy_in(int32_t *dst, const int32_t *src, int n, int tid, int T) {
    if ((((uintptr_t)src) & 15u) == 0) {
        const int n4 = n >> 2;
        const int4 *s4 = reinterpret_cast<const int4 *>(src);
        int4 *d4 = reinterpret_cast<int4 *>(dst);
        for (int i = tid; i < n4; i += T) d4[i] = s4[i];
        for (int i = (n4 << 2) + tid; i < n; i += T) dst[i] = src[i];
    } else {
        for (int i = tid; i < n; i += T) dst[i] = src[i];
    }
}
__device__ __forceinline__ void copy_out(int32_t *dst, const int32_t *src, int n, int tid, int T) {
    if ((((uintptr_t)dst) & 15u) == 0) {
        const int n4 = n >> 2;
        const int4 *s4 = reinterpret_cast<const int4 *>(src);
        int4 *d4 = reinterpret_cast<int4 *>(dst);
        for (int i = tid; i < n4; i += T) d4[i] = s4[i];
        for (int i = (n4 << 2) + tid; i < n; i += T) dst[i] = src[i];
    } else {
        for (int i = tid; i < n; i += T) dst[i] = src[i];
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

template <int R>
__global__ void __launch_bounds__(256) rware_step_kernel(const Params p, const int op) {
    constexpr int WIN = 2 * R + 1, CELLS = WIN * WIN, L = 8 + 7 * CELLS, OW = (L + 31) / 32;
    extern __shared__ __align__(16) int32_t smem[];

    const int tid = threadIdx.x, T = blockDim.x;
    const int E = p.envs_per_wg;
    const int e0 = blockIdx.x * E;
    const int ne = min(E, p.B - e0);
    if (ne <= 0) return;
    const int N = p.N, Q = p.Q, HW = p.HW, W = p.W, H = p.H, SW = p.SW, B = p.B;
    const int nea = ne * N;

    const LdsLayout lo = make_lds_layout(E, N, Q, HW, SW, OW);
    int32_t *s_grid = smem + lo.grid;
    int32_t *s_ax = smem + lo.ax, *s_ay = smem + lo.ay, *s_dir = smem + lo.dir;
    int32_t *s_carry = smem + lo.carry, *s_deliv = smem + lo.deliv;
    int32_t *s_act = smem + lo.act, *s_start = smem + lo.start, *s_tgt = smem + lo.tgt;
    int32_t *s_nxt = smem + lo.nxt, *s_depth = smem + lo.depth, *s_win = smem + lo.win;
    float *s_rew = reinterpret_cast<float *>(smem + lo.rew);
    int32_t *s_queue = smem + lo.queue;
    uint32_t *s_req = reinterpret_cast<uint32_t *>(smem + lo.req);
    uint32_t *s_obits = reinterpret_cast<uint32_t *>(smem + lo.obits);
    int32_t *s_envi = smem + lo.envi;
    int32_t *s_misc = smem + lo.misc;

    // ---------------------------------------------------------------- P0: stage the env chunk
    copy_in(s_grid, p.grid + (size_t)e0 * 2 * HW, ne * 2 * HW, tid, T);
    copy_in(s_ax, p.ax + (size_t)e0 * N, nea, tid, T);
    copy_in(s_ay, p.ay + (size_t)e0 * N, nea, tid, T);
    copy_in(s_dir, p.adir + (size_t)e0 * N, nea, tid, T);
    copy_in(s_carry, p.acarry + (size_t)e0 * N, nea, tid, T);
    copy_in(s_deliv, p.adeliv + (size_t)e0 * N, nea, tid, T);
    copy_in(s_queue, p.queue + (size_t)e0 * Q, ne * Q, tid, T);
    if (op == OP_STEP) copy_in(s_act, p.actions + (size_t)e0 * N, nea, tid, T);
    for (int e = tid; e < ne; e += T) {
        int32_t *ev = s_envi + e * ENVI_W;
        ev[ENVI_STEPS] = p.steps[e0 + e];
        ev[ENVI_INACTIVE] = p.inactive[e0 + e];
        int rs = 0;
        if (op == OP_STEP) rs = (p.autoreset == AR_NEXT_STEP) ? (int)p.need_reset[e0 + e] : 0;
        else if (op == OP_RESET) rs = p.reset_mask ? (int)p.reset_mask[e0 + e] : 1;
        ev[ENVI_RESET] = rs;
        ev[ENVI_SKIP] = rs;  // an env that resets in this call does not step
        ev[ENVI_DONE] = 0;
    }
    for (int i = tid; i < nea; i += T) {
        s_depth[i] = 0;
        s_rew[i] = 0.0f;
    }
    for (int i = tid; i < nea * OW; i += T) s_obits[i] = 0u;
    for (int i = tid; i < ne * SW; i += T) s_req[i] = 0u;
    if (tid == 0) s_misc[0] = 0;
    __syncthreads();

    if (op == OP_STEP) {
        // ------------------------------------------------------------ P1: intent (:825-846)
        for (int i = tid; i < nea; i += T) {
            const int e = i / N;
            if (s_envi[e * ENVI_W + ENVI_SKIP]) continue;
            const int32_t *gA = s_grid + e * 2 * HW, *gS = gA + HW;
            int a = s_act[i];
            if ((unsigned)a > 4u) {  // Action(a) raises in the reference (:814); flagged, runs as NOOP
                atomicOr(p.status, STATUS_INVALID_ACTION);
                a = ACT_NOOP;
            }
            const int x = s_ax[i], y = s_ay[i], d = s_dir[i];
            int tx = x, ty = y;
            if (a == ACT_FORWARD) {  // clamped at the walls (:105-112)
                if (d == DIR_UP) ty = max(0, y - 1);
                else if (d == DIR_DOWN) ty = min(H - 1, y + 1);
                else if (d == DIR_LEFT) tx = max(0, x - 1);
                else tx = min(W - 1, x + 1);
            }
            const int st = y * W + x;
            int tg = ty * W + tx;
            if (s_carry[i] && tg != st && gS[tg]) {
                const int occ = gA[tg];
                if (!(occ && s_carry[e * N + occ - 1])) {  // a standing shelf blocks a loaded agent
                    a = ACT_NOOP;
                    tg = st;
                }
            }
            s_act[i] = a;
            s_start[i] = st;
            s_tgt[i] = tg;
            // successor on the chain: agent index on the target cell, -1 empty, -2 == i is stationary
            s_nxt[i] = (tg == st) ? -2 : (gA[tg] - 1);
        }
        __syncthreads();
        // ------------------------------------------------------------ P2a: follower depth
        for (int i = tid; i < nea; i += T) {
            const int e = i / N, base = e * N, me = i - base;
            if (s_envi[e * ENVI_W + ENVI_SKIP] || s_nxt[i] == -2) continue;
            int j = s_nxt[i], dd = 1;
            while (j >= 0 && j != me && dd <= N && s_nxt[base + j] != -2) {
                atomicMax(&s_depth[base + j], dd);
                j = s_nxt[base + j];
                ++dd;
            }
        }
        __syncthreads();
        // ------------------------------------------------------------ P2b: winner per contested cell
        for (int i = tid; i < nea; i += T) {
            const int e = i / N, base = e * N, me = i - base;
            if (s_envi[e * ENVI_W + ENVI_SKIP]) continue;
            int w = 1;
            if (s_nxt[i] != -2) {
                const int tg = s_tgt[i], dme = s_depth[i];
                for (int k = 0; k < N; ++k) {
                    if (k == me || s_nxt[base + k] == -2 || s_tgt[base + k] != tg) continue;
                    const int dk = s_depth[base + k];
                    if (dk > dme || (dk == dme && k < me)) w = 0;
                }
            }
            s_win[i] = w;
        }
        __syncthreads();
        // ------------------------------------------------------------ P2c + P3: commit, apply (:871-899)
        for (int i = tid; i < nea; i += T) {
            const int e = i / N, base = e * N, me = i - base;
            if (s_envi[e * ENVI_W + ENVI_SKIP]) continue;
            int32_t *gA = s_grid + e * 2 * HW, *gS = gA + HW;
            int a = s_act[i];
            if (s_nxt[i] != -2) {  // a mover: walk the chain ahead
                int j = me, hops = 0, ok = 1, commit = 0;
                for (;;) {
                    ok &= s_win[base + j];
                    const int nj = s_nxt[base + j];
                    ++hops;
                    if (nj == -1) { commit = ok; break; }        // drains into an empty cell
                    if (nj == me) { commit = (hops >= 3); break; }  // a cycle through me; 2-swap refused
                    if (s_nxt[base + nj] == -2) break;            // blocked by a stationary agent
                    if (hops >= N) break;                          // feeds a cycle it is not part of
                    j = nj;
                }
                if (!commit) a = ACT_NOOP;
            }
            const int st = s_start[i], tg = s_tgt[i];
            int carry = s_carry[i];
            if (a == ACT_FORWARD) {
                if (tg != st) {
                    s_ax[i] = tg % W;
                    s_ay[i] = tg / W;
                    gA[st] = 0;  // clear phase of the incremental _recalc_grid
                    if (carry) gS[st] = 0;
                }
            } else if (a == ACT_LEFT || a == ACT_RIGHT) {
                // wraplist [UP, RIGHT, DOWN, LEFT] (:119): RIGHT 0->3->1->2->0, LEFT 0->2->1->3->0
                const int d = s_dir[i];
                const int right = (0x1023 >> (4 * d)) & 0xF;  // d: 0->3, 1->2, 2->0, 3->1
                const int left = (0x0132 >> (4 * d)) & 0xF;   // d: 0->2, 1->3, 2->1, 3->0
                s_dir[i] = (a == ACT_RIGHT) ? right : left;
            } else if (a == ACT_TOGGLE) {
                if (!carry) {
                    const int sid = gS[st];
                    if (sid) s_carry[i] = sid;
                } else if (!p.highways[st]) {
                    s_carry[i] = 0;
                    if (s_deliv[i] && p.reward_type == REW_TWO_STAGE) s_rew[i] += 0.5f;
                    s_deliv[i] = 0;
                }
            }
            s_act[i] = (a == ACT_FORWARD && tg != st) ? 1 : 0;  // from here on: "moved" flag
        }
        __syncthreads();
        for (int i = tid; i < nea; i += T) {  // set phase
            const int e = i / N;
            if (s_envi[e * ENVI_W + ENVI_SKIP] || !s_act[i]) continue;
            int32_t *gA = s_grid + e * 2 * HW, *gS = gA + HW;
            gA[s_tgt[i]] = (i - e * N) + 1;
            if (s_carry[i]) gS[s_tgt[i]] = s_carry[i];
        }
        __syncthreads();
        // ------------------------------------------------------------ grid patch to HBM + P5 goals
        for (int i = tid; i < nea; i += T) {
            const int e = i / N;
            if (s_envi[e * ENVI_W + ENVI_SKIP] || !s_act[i]) continue;
            const int32_t *gA = s_grid + e * 2 * HW, *gS = gA + HW;
            int32_t *hA = p.grid + (size_t)(e0 + e) * 2 * HW, *hS = hA + HW;
            const int st = s_start[i], tg = s_tgt[i];
            hA[st] = gA[st];
            hA[tg] = gA[tg];
            if (s_carry[i]) {
                hS[st] = gS[st];
                hS[tg] = gS[tg];
            }
        }
        for (int e = tid; e < ne; e += T) {
            int32_t *ev = s_envi + e * ENVI_W;
            if (ev[ENVI_SKIP]) continue;
            const int32_t *gA = s_grid + e * 2 * HW, *gS = gA + HW;
            int32_t *q = s_queue + e * Q;
            bool delivered = false;
            for (int g = 0; g < p.n_goals; ++g) {  // in list order (:904)
                const int cell = p.goal_cells[g];
                const int sid = gS[cell];
                if (!sid) continue;
                int slot = -1;
                for (int k = 0; k < Q; ++k)
                    if (q[k] == sid) { slot = k; break; }
                if (slot < 0) continue;
                delivered = true;
                // candidates = shelves not in the queue, id order; one bounded draw (:915-916)
                Pcg64 rg;
                rng_load(rg, p.rng, B, e0 + e);
                const int idx = (int)pcg_bounded(rg, (uint32_t)(p.S - Q - 1));
                rng_store(rg, p.rng, B, e0 + e);
                int cand = idx + 1;  // idx-th id (0-based) among ids 1..S that are not queued
                for (;;) {
                    int c = 0;
                    for (int k = 0; k < Q; ++k) c += (q[k] <= cand) ? 1 : 0;
                    const int nc = idx + 1 + c;
                    if (nc == cand) break;
                    cand = nc;
                }
                q[slot] = cand;
                if (p.reward_type == REW_GLOBAL) {
                    for (int k = 0; k < N; ++k) s_rew[e * N + k] += 1.0f;
                } else {
                    const int aid = gA[cell];
                    const int ai = aid > 0 ? aid - 1 : N - 1;  // rewards[-1] when nobody stands there
                    if (p.reward_type == REW_INDIVIDUAL) {
                        s_rew[e * N + ai] += 1.0f;
                    } else {
                        s_deliv[e * N + ai] = 1;
                        s_rew[e * N + ai] += 0.5f;
                    }
                }
            }
            ev[ENVI_INACTIVE] = delivered ? 0 : ev[ENVI_INACTIVE] + 1;
            ev[ENVI_STEPS] += 1;
            const int done = ((p.max_inactivity && ev[ENVI_INACTIVE] >= p.max_inactivity) ||
                              (p.max_steps && ev[ENVI_STEPS] >= p.max_steps)) ? 1 : 0;
            ev[ENVI_DONE] = done;
            if (done && p.autoreset == AR_SAME_STEP) ev[ENVI_RESET] = 1;
        }
        __syncthreads();
    }

    // ---------------------------------------------------------------- RS: reset flagged envs (:757-802)
    for (int e = tid; e < ne; e += T)
        if (s_envi[e * ENVI_W + ENVI_RESET]) s_misc[0] = 1;
    __syncthreads();
    if (s_misc[0]) {  // workgroup-uniform
        for (int c = tid; c < ne * 2 * HW; c += T) {
            const int e = c / (2 * HW);
            if (!s_envi[e * ENVI_W + ENVI_RESET]) continue;
            const int r = c - e * 2 * HW;
            s_grid[c] = (r < HW) ? 0 : p.shelf_init[r - HW];
        }
        __syncthreads();
        for (int e = tid; e < ne; e += T) {
            int32_t *ev = s_envi + e * ENVI_W;
            if (!ev[ENVI_RESET]) continue;
            Pcg64 rg;
            rng_load(rg, p.rng, B, e0 + e);
            int32_t *cells = s_tgt + e * N;  // scratch
            pcg_choice_no_replace(rg, HW, N, cells);  // agent cells (:781-786)
            int32_t *gA = s_grid + e * 2 * HW;
            for (int k = 0; k < N; ++k) {
                const int c = cells[k];
                s_ax[e * N + k] = c % W;
                s_ay[e * N + k] = c / W;
                gA[c] = k + 1;
            }
            for (int k = 0; k < N; ++k) {  // directions (:788)
                s_dir[e * N + k] = (int)pcg_bounded(rg, 3u);
                s_carry[e * N + k] = 0;
                s_deliv[e * N + k] = 0;
            }
            int32_t *q = s_queue + e * Q;  // request queue (:796-800)
            pcg_choice_no_replace(rg, p.S, Q, q);
            for (int k = 0; k < Q; ++k) q[k] += 1;
            rng_store(rg, p.rng, B, e0 + e);
            ev[ENVI_STEPS] = 0;
            ev[ENVI_INACTIVE] = 0;
        }
        __syncthreads();
        for (int c = tid; c < ne * 2 * HW; c += T) {  // a reset rewrites the env's whole grid in HBM
            const int e = c / (2 * HW);
            if (s_envi[e * ENVI_W + ENVI_RESET]) p.grid[(size_t)e0 * 2 * HW + c] = s_grid[c];
        }
    }

    // ---------------------------------------------------------------- P7: observation bits (:598-674)
    for (int i = tid; i < ne * Q; i += T) {
        const int e = i / Q, sid = s_queue[i];
        atomicOr(&s_req[e * SW + (sid >> 5)], 1u << (sid & 31));
    }
    __syncthreads();
    // bit k of agent i's string == obs[i][k] for k >= 2 (k = 0,1 are the coordinates)
    for (int i = tid; i < nea; i += T) {
        const int c = s_ay[i] * W + s_ax[i];
        const uint32_t self = (s_carry[i] ? 4u : 0u) | (8u << s_dir[i]) | (p.highways[c] ? 128u : 0u);
        atomicOr(&s_obits[i * OW], self);
    }
    for (int w = tid; w < nea * CELLS; w += T) {
        const int i = w / CELLS, cidx = w - i * CELLS;
        const int e = i / N;
        const int dy = cidx / WIN - R, dx = cidx % WIN - R;
        const int x = s_ax[i] + dx, y = s_ay[i] + dy;
        uint32_t code = 2u;  // empty / off-map cell: has_agent 0, direction one-hot [1,0,0,0] (:659)
        if ((unsigned)x < (unsigned)W && (unsigned)y < (unsigned)H) {
            const int32_t *gA = s_grid + e * 2 * HW, *gS = gA + HW;
            const int c = y * W + x;
            const int ida = gA[c], ids = gS[c];
            if (ida) code = 1u | (2u << s_dir[e * N + ida - 1]);
            if (ids) code |= 32u | (((s_req[e * SW + (ids >> 5)] >> (ids & 31)) & 1u) << 6);
        }
        const int bit = 8 + 7 * cidx;
        const int wd = bit >> 5, sh = bit & 31;
        atomicOr(&s_obits[i * OW + wd], code << sh);
        if (sh > 25) atomicOr(&s_obits[i * OW + wd + 1], code >> (32 - sh));
    }
    __syncthreads();

    // ---------------------------------------------------------------- obs expansion + coalesced store
    {
        const int nf = nea * L;
        float *out = p.obs + (size_t)e0 * N * L;
        const float nx = p.normalised ? 1.0f : 0.0f;
        auto elem = [&](int g) -> float {
            const int i = g / L, k = g - i * L;
            if (k >= 2) return ((s_obits[i * OW + (k >> 5)] >> (k & 31)) & 1u) ? 1.0f : 0.0f;
            const int v = (k == 0) ? s_ax[i] : s_ay[i];
            if (nx != 0.0f) return (float)((double)v / (double)((k == 0 ? W : H) - 1));  // :636-638
            return (float)v;
        };
        if ((((uintptr_t)out) & 15u) == 0) {
            const int nf4 = nf >> 2;
            float4 *out4 = reinterpret_cast<float4 *>(out);
            for (int q4 = tid; q4 < nf4; q4 += T) {
                float4 v;
                v.x = elem(4 * q4 + 0);
                v.y = elem(4 * q4 + 1);
                v.z = elem(4 * q4 + 2);
                v.w = elem(4 * q4 + 3);
                out4[q4] = v;
            }
            for (int g = (nf4 << 2) + tid; g < nf; g += T) out[g] = elem(g);
        } else {
            for (int g = tid; g < nf; g += T) out[g] = elem(g);
        }
    }

    // ---------------------------------------------------------------- ST: write back state
    if (op != OP_OBS) {
        copy_out(p.ax + (size_t)e0 * N, s_ax, nea, tid, T);
        copy_out(p.ay + (size_t)e0 * N, s_ay, nea, tid, T);
        copy_out(p.adir + (size_t)e0 * N, s_dir, nea, tid, T);
        copy_out(p.acarry + (size_t)e0 * N, s_carry, nea, tid, T);
        copy_out(p.adeliv + (size_t)e0 * N, s_deliv, nea, tid, T);
        copy_out(p.queue + (size_t)e0 * Q, s_queue, ne * Q, tid, T);
        copy_out(reinterpret_cast<int32_t *>(p.rewards) + (size_t)e0 * N,
                 reinterpret_cast<const int32_t *>(s_rew), nea, tid, T);
        for (int e = tid; e < ne; e += T) {
            const int32_t *ev = s_envi + e * ENVI_W;
            if (op == OP_RESET && !ev[ENVI_RESET]) continue;  // a masked reset leaves other envs' flags alone
            p.steps[e0 + e] = ev[ENVI_STEPS];
            p.inactive[e0 + e] = ev[ENVI_INACTIVE];
            p.terminated[e0 + e] = (uint8_t)ev[ENVI_DONE];
            p.truncated[e0 + e] = 0;  // the reference never truncates (:942)
            p.need_reset[e0 + e] = (uint8_t)((p.autoreset == AR_NEXT_STEP) ? ev[ENVI_DONE] : 0);
        }
    }
}

}  // namespace rw
