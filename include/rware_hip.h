/*
 * rware_hip.h — C-ABI of the MI355X-native vectorised RWARE step engine.
 *
 * This is the drop-in boundary for ONE path of semitable/robotic-warehouse: the per-step hot
 * path of `rware.warehouse.Warehouse` (reset / step / FLATTENED observation), batched over
 * `num_envs` independent warehouses resident in HBM on one HIP device.  The reference has no
 * native code, so there is no existing FFI to mirror; each entry point below cites the Python
 * interface it replaces.  The Python binding a maintainer would add is in INTEGRATION.md; the
 * in-tree one is robotic-warehouse_amd/_capi.py.
 *
 * Conventions
 *   - plain C types only; every call returns an `int` status (RW_OK == 0, negatives are errors)
 *     and never throws or aborts across the boundary; `rw_last_error` gives the message.
 *   - the engine owns all device memory for its lifetime; pointers from `rw_get_buffer` are
 *     borrowed, stable until `rw_destroy`, and the obs/reward/terminated buffers are overwritten
 *     by the next step.  Callers own the action arrays they pass in.
 *   - one engine == one HIP device + one stream.  Calls on one engine are not re-entrant;
 *     different engines (one per GPU) may be driven from different host threads.
 *   - all work is enqueued asynchronously on the engine's stream; `rw_sync` waits for it.
 *   - env index e, agent index i (agent id = i+1), cell index c = y*W + x.
 */
#ifndef RWARE_HIP_H
#define RWARE_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define RW_ABI_VERSION 3

typedef struct rw_engine rw_engine;

enum rw_status {
    RW_OK = 0,
    RW_ERR_INVALID_ARG = -1,    /* bad config / null pointer / size mismatch                     */
    RW_ERR_INVALID_ACTION = -2, /* an action outside 0..4 was seen (reference: Action(a) raises
                                   ValueError, rware/warehouse.py:814); sticky until rw_sync      */
    RW_ERR_HIP = -3,            /* a HIP runtime call failed                                      */
    RW_ERR_UNSUPPORTED = -4,    /* feature outside the accelerated path (DICT observations, ...)  */
    RW_ERR_NO_DEVICE = -5,      /* no usable HIP device: there is NO CPU fallback                 */
    RW_ERR_INDEX = -6,          /* an AGENT_DIRECTION / AGENT_LOAD image layer met an agent at
                                   x >= grid_h or y >= grid_w, where the reference raises IndexError
                                   (rware/warehouse.py:552,558); sticky until rw_sync             */
    RW_ERR_SELFTEST = -7        /* rw_selftest: this device / toolchain breaks an assumption the kernels are built on */
};

/* rware/warehouse.py:31-36 */
enum rw_action { RW_NOOP = 0, RW_FORWARD = 1, RW_LEFT = 2, RW_RIGHT = 3, RW_TOGGLE_LOAD = 4 };
/* rware/warehouse.py:39-43 */
enum rw_direction { RW_UP = 0, RW_DOWN = 1, RW_DIR_LEFT = 2, RW_DIR_RIGHT = 3 };
/* rware/warehouse.py:46-49 */
enum rw_reward_type { RW_REWARD_GLOBAL = 0, RW_REWARD_INDIVIDUAL = 1, RW_REWARD_TWO_STAGE = 2 };

/* rware/warehouse.py:52-56.  DICT (nested Python objects) is FLATTENED at this boundary: the host layer
 * un-flattens the same vector (rware/warehouse.py:432-443). */
enum rw_observation_type { RW_OBS_FLATTENED = 1, RW_OBS_IMAGE = 2, RW_OBS_IMAGE_DICT = 3 };
/* rware/warehouse.py:59-70.  AGENT_DIRECTION and AGENT_LOAD are reproduced as the reference writes them:
 * with transposed indices (`layer[ag.x, ag.y]` on an (H, W) array, :552/:558), i.e. the value lands on the
 * mirrored cell, and a state in which the reference raises IndexError is reported as RW_ERR_INDEX by the
 * next rw_sync (every registered layout has H > W, so that happens as soon as an agent reaches y >= W). */
enum rw_image_layer { RW_LAYER_SHELVES = 0, RW_LAYER_REQUESTS = 1, RW_LAYER_AGENTS = 2, RW_LAYER_AGENT_DIRECTION = 3,
                      RW_LAYER_AGENT_LOAD = 4, RW_LAYER_GOALS = 5, RW_LAYER_ACCESSIBLE = 6 };

/* What a step does with an env whose episode ended (Gymnasium vector-env autoreset modes).
 * The reference env itself never resets (the caller calls reset()); DISABLED reproduces that. */
enum rw_autoreset {
    RW_AUTORESET_DISABLED = 0,
    RW_AUTORESET_NEXT_STEP = 1, /* the step after `terminated` performs reset(): action ignored,
                                   reward 0, terminated 0 (Gymnasium >= 1.0 default)              */
    RW_AUTORESET_SAME_STEP = 2  /* the terminating step returns the reset observation; the terminal
                                   observation is kept in RW_BUF_FINAL_OBS (Gymnasium's
                                   info["final_obs"])                                             */
};

/* How `rw_config.stream` is read.  A hipStream_t of NULL is also the handle of the device's default
 * ("null") stream — the stream frameworks such as PyTorch run on unless told otherwise — so "NULL == make
 * your own" cannot express "enqueue on the default stream".  RW_STREAM_USE_GIVEN takes the handle
 * literally: NULL then means the default stream, and the engine's launches are ordered with everything
 * else the caller enqueues there (a policy writing the action tensor before, a learner reading the
 * observation tensor after) without any explicit synchronisation. */
enum rw_stream_flags {
    RW_STREAM_USE_GIVEN = 1,
    /* How the observation stream is stored.  Default (neither bit): the engine decides — non-temporal stores (the observation
     * lines are written once and not re-read by the engine: with the hint they do not displace the state the next step
     * reads) wherever that was measured faster, which is everywhere except workgroups with a large observation chunk
     * (16 agents) below the Infinity Cache size.  A learner that reads the observations right behind the step finds
     * non-temporal lines in HBM rather than in the cache: RW_OBS_STORES_CACHED keeps them cached, RW_OBS_STORES_STREAM forces
     * the hint.  (A/B runs: RWARE_OBS_STORES=cached|stream in the environment, honoured with RWARE_HOOKS=1 — csrc/rware_hooks.h.) */
    RW_OBS_STORES_CACHED = 2,
    RW_OBS_STORES_STREAM = 4,
    /* Run-time specialisation.  A task without an ahead-of-time exact-shape kernel build — a `layout=` string, column_height
     * != 8, sensor_range 2..5, more than 19 agents (rware/warehouse.py:146-170; ids of rware/__init__.py:83-175) — gets one
     * compiled by rw_create through hipRTC (a few seconds; cached on disk under $RWARE_JIT_CACHE or ~/.cache/rware_amd/jit)
     * when the batch has at least 4096 envs; without hipRTC, or when the compile fails, the ahead-of-time generic kernel
     * runs (rw_info.jit, rw_jit_log say what happened).  RW_JIT_OFF: never.  RW_JIT_FORCE: always — small batches and shapes
     * that have an ahead-of-time build included (tests, A/B).  (A/B runs: RWARE_JIT=off|force with RWARE_HOOKS=1.) */
    RW_JIT_OFF = 8,
    RW_JIT_FORCE = 16,
    /* The chunk-pipelined persistent build of the per-step kernel (round 5): instead of one workgroup per chunk of envs, as many
     * workgroups as the GPU holds at once walk the chunks, the agent phases of the next chunk running beside the observation stores
     * of the current one and the chunk after that already on its way into LDS.  Measured SLOWER than the classic launch on every
     * configuration (profiles/EXPERIMENTS.md, round 5), so: rw_create NEVER picks it by itself, and the default library does not even
     * contain it — only a `make PIPE=1` build does (rware_static_table.h, group 18: the BASELINE shapes and two agent-count-static
     * ones).  RW_PIPE_ON: use it where such a build exists for the shape (FLATTENED, no messages, ahead-of-time kernels, batch a multiple of
     * its chunk size), else the classic kernel runs and rw_jit_log() says why; rw_info.pipe_workgroups != 0 tells which one runs.
     * RW_PIPE_OFF: never (the default).  (A/B runs: RWARE_PIPE=0|1 with RWARE_HOOKS=1.) */
    RW_PIPE_OFF = 32,
    RW_PIPE_ON = 64,
    /* Event counters (SURVEY.md §5 "metrics"): with RW_STATS_ON the engine keeps two running totals per env in RW_BUF_STAT_DELIVERIES and
     * RW_BUF_STAT_FAILED_MOVES (below) — off by default, and the reference's `info` stays {} either way (rware/warehouse.py:746-747).
     * Counted by the service wavefront beside the observation stores (re-derived from what the state write-back holds; nothing is added
     * to the agent phases), in every launch form: rw_step*, fused rollouts, tapes, HIP graphs, rw_multi. */
    RW_STATS_ON = 128,
    /* Wavefront priority (rw_info.wave_priority): the launches run the dependent chain in front of their first observation store at raised
     * priority by rw_create's measured rule (on, except the per-step launches of 13 .. 16 agents at sensor_range 1 and steps of 200 MB of
     * observations and more).  RW_PRIO_OFF: never — what rware_amd.make_pipelines asks for when a sub-batch has 8192 envs or fewer: with TWO
     * launches in flight the raised chain of one takes the issue slots of the other's store phase, which is what fills the gap (two
     * pipelines of 8192 envs, per step of the whole batch: small-8ag 7.9 us without against 9.1 with, small-10ag 10.7 / 11.9, medium-13ag
     * 13.4 / 15.4; profiles/r06_pipelines_prio.txt).  RW_PRIO_ON: always (per-step launches and fused rollouts alike).  A scheduling
     * hint either way: same results.  (A/B runs: RWARE_PRIO=0|1, RWARE_PRIO_ROLLOUT=0|1 with RWARE_HOOKS=1 move the default only.) */
    RW_PRIO_OFF = 256,
    RW_PRIO_ON = 512
};

/* Device buffers (all env-major, C-contiguous).  Replaces the attributes callers read off the
 * reference object: env.grid (:302), env.agents[i].{x,y,dir,carrying_shelf,has_delivered}
 * (:82-93), env.request_queue (:263), env._cur_steps/_cur_inactive_steps (:249-250),
 * env.np_random.bit_generator.state, and the step() return tuple (:944-946). */
enum rw_buffer_kind {
    RW_BUF_OBS = 0,          /* float32 [B][N][L]  FLATTENED observation (:598-674), or, for the IMAGE
                                                   types, [B][N][C][2r+1][2r+1] (:527-596)        */
    RW_BUF_REWARDS = 1,      /* float32 [B][N]                                                    */
    RW_BUF_TERMINATED = 2,   /* uint8   [B]        `done` (:935-941)                              */
    RW_BUF_TRUNCATED = 3,    /* uint8   [B]        always 0 (:942); read-only (rw_write refuses it) */
    RW_BUF_GRID = 4,         /* int32   [B][2][H][W] layer 0 agent ids, layer 1 shelf ids (:11-14).
                                                   A DERIVED VIEW: rebuilt from the state the kernels keep (shelf
                                                   layer, agent coordinates) by rw_read / rw_get_buffer of this kind
                                                   and by rw_refresh_grid — steps enqueued after that do not touch
                                                   it (every other buffer is current after every step)           */
    /* The five agent arrays are DERIVED VIEWS as well (round 3): the kernels keep an agent as ONE packed dword
     * (cell, dir, has_delivered, carried shelf) — one load stream and one store stream per step instead of five each.
     * rw_read / rw_get_buffer of these kinds unpack the records first; rw_write of one of them re-packs all five;
     * rw_refresh_grid brings pointers borrowed earlier up to date.                                                  */
    RW_BUF_AGENT_X = 5,      /* int32   [B][N]     Agent.x (:86)                                  */
    RW_BUF_AGENT_Y = 6,      /* int32   [B][N]     Agent.y                                        */
    RW_BUF_AGENT_DIR = 7,    /* int32   [B][N]     rw_direction                                   */
    RW_BUF_AGENT_CARRY = 8,  /* int32   [B][N]     carried shelf id, 0 == none                    */
    RW_BUF_AGENT_DELIVERED = 9, /* int32 [B][N]    has_delivered                                  */
    RW_BUF_QUEUE = 10,       /* int32   [B][Q]     requested shelf ids in slot order              */
    RW_BUF_STEPS = 11,       /* int32   [B]                                                       */
    RW_BUF_INACTIVE = 12,    /* int32   [B]                                                       */
    RW_BUF_RNG = 13,         /* uint64  [6][B]     FIELD-major: state_hi, state_lo, inc_hi, inc_lo,
                                                   has_uint32, uinteger (numpy PCG64 state)       */
    RW_BUF_NEED_RESET = 14,  /* uint8   [B]        NEXT_STEP autoreset: env resets on next step   */
    RW_BUF_ACTIONS = 15,     /* int32   [B][N][1+M] staging buffer used by rw_step (host actions)  */
    RW_BUF_FEATURES = 16,    /* float32 [B][N][6]  IMAGE_DICT features: one-hot direction, on_highway,
                                                   carrying (:727-742); unused otherwise          */
    RW_BUF_AGENT_MSG = 17,   /* int32   [B][N]     bit k == message[k] of the agent (msg_bits > 0, :89)   */
    RW_BUF_FINAL_OBS = 18,   /* float32 [B][N][L]  SAME_STEP autoreset: the observation of the terminating step itself — what
                                                   Warehouse.step returns with done = True (:929-946; the image, :527-596, for the
                                                   IMAGE types) — written for the envs whose `terminated` flag that step set (rows of
                                                   other envs keep older contents); the reset observation goes to RW_BUF_OBS.  Empty in
                                                   the other autoreset modes                                              */
    RW_BUF_FINAL_FEATURES = 19, /* float32 [B][N][6] ... and, IMAGE_DICT, the feature vectors of that observation (:727-742)   */
    /* RW_STATS_ON only (empty otherwise): running totals per env since rw_create — steps add to them, NOTHING resets them (not reset(),
     * not autoreset: an episode's figure is the difference between two reads, and SAME_STEP's terminating step is counted before its
     * reset); rw_write zeroes or seeds them, snapshots carry them; int32, wrapping.  What is counted, in the reference's terms: */
    RW_BUF_STAT_DELIVERIES = 20,   /* int32 [B]  requested shelves brought to a goal: one per queue slot replaced (:907-917)             */
    RW_BUF_STAT_FAILED_MOVES = 21, /* int32 [B]  (agent, step) pairs whose FORWARD the step turned into NOOP: a loaded agent facing a
                                                 standing shelf (:836-846) or a mover that lost the collision resolution (:871-876).
                                                 A FORWARD into a wall is clamped to a self-target (:105-112) and is NOT one            */
    RW_BUF_KIND_COUNT = 22
};

/* Mirrors the constructor of rware.warehouse.Warehouse (rware/warehouse.py:146-170).  The
 * layout (`highways`, `goals`) is computed host-side exactly as _make_layout_from_params /
 * _make_layout_from_str do (:294-350) and handed over as plain arrays. */
typedef struct rw_config {
    int32_t abi_version;          /* RW_ABI_VERSION                                              */
    int32_t num_envs;             /* B                                                           */
    int32_t grid_h, grid_w;       /* grid_size (:297-300)                                        */
    int32_t n_agents;             /* N, 1..64                                                    */
    int32_t sensor_range;         /* r, 1..5 ; L = 8 + 7*(2r+1)^2 (:432-443, msg_bits == 0)      */
    int32_t request_queue_size;   /* Q                                                           */
    int32_t max_inactivity_steps; /* 0 == None                                                   */
    int32_t max_steps;            /* 0 == None                                                   */
    int32_t reward_type;          /* rw_reward_type                                              */
    int32_t normalised_coordinates;
    int32_t autoreset_mode;       /* rw_autoreset                                                */
    int32_t n_goals;
    int32_t device_id;            /* HIP device ordinal                                          */
    int32_t envs_per_workgroup;   /* 0 == engine default; otherwise a multiple of 4              */
    int32_t threads_per_workgroup;/* 0 == engine default; otherwise a multiple of 64             */
    int32_t observation_type;     /* rw_observation_type; 0 is read as FLATTENED                 */
    int32_t image_directional;    /* image_observation_directional (:167)                        */
    int32_t n_image_layers;       /* 0 == the reference default list (:160-166)                  */
    int32_t image_layers[8];      /* rw_image_layer values, channel order                        */
    int32_t msg_bits;             /* M communication bits (:152): an agent's action is then
                                     [Action, bit_0..bit_{M-1}]; FLATTENED L = 8 + (7+M)(2r+1)^2      */
    int32_t stream_flags;         /* rw_stream_flags                                              */
    const uint8_t *highways;      /* host, [H*W], 1 == highway (no shelf spawns, no unloading)   */
    const int32_t *goals_xy;      /* host, [n_goals][2] = (x, y), list order == reward order     */
    void *stream;                 /* hipStream_t to enqueue on; NULL == engine creates its own
                                     (non-blocking) stream, unless RW_STREAM_USE_GIVEN is set     */
} rw_config;

/* -- lifetime ------------------------------------------------------------------------- */
/* replaces Warehouse.__init__ (:146-292) */
int rw_create(const rw_config *cfg, rw_engine **out);
int rw_destroy(rw_engine *eng);
/* message for the most recent failing call on this engine (or on rw_create when eng == NULL) */
const char *rw_last_error(const rw_engine *eng);

/* -- the hot path ----------------------------------------------------------------------- */
/* replaces Warehouse.reset(seed, options) (:757-802).
 *   seeds: host [B] or NULL.  Non-NULL: env e is reseeded with numpy SeedSequence(seeds[e]) ->
 *          PCG64 (what gymnasium.utils.seeding.np_random does, :758-760) before the reset draws.
 *          NULL: the env's existing stream continues (reset() with seed=None).
 *   mask:  host [B] or NULL (all).  Only envs with mask[e] != 0 are reseeded/reset.
 * Observations of all envs are refreshed in RW_BUF_OBS. */
int rw_reset(rw_engine *eng, const uint64_t *seeds, const uint8_t *mask);

/* replaces Warehouse.step(actions) (:804-946) for all B envs.
 *   rw_step:        actions is a HOST array int32 [B][N] ([B][N][1+M] with msg_bits = M); copied into a
 *                   pinned staging buffer owned by the engine before the call returns (the caller may
 *                   free or overwrite its array immediately), from there to RW_BUF_ACTIONS, then launched.
 *   rw_step_device: actions is a DEVICE array of the same shape (e.g. the policy's output tensor);
 *                   no copy.  Must stay valid until the step has executed.
 * Results land in RW_BUF_OBS / REWARDS / TERMINATED / TRUNCATED. */
int rw_step(rw_engine *eng, const int32_t *actions_host);
int rw_step_device(rw_engine *eng, const int32_t *actions_dev);

/* n_steps consecutive rw_step_device launches (ONE kernel launch per step, each ordered behind the previous one on the
 * engine's stream) from a device-resident action tape int32 [tape_steps][B][N][1+M]: step k reads tape row
 * (first + k) % tape_steps.  Exactly the launches n_steps calls of rw_step_device would make, issued from one native loop:
 * replaying recorded actions, open-loop evaluation, benchmarking without per-call host overhead. */
int rw_step_tape_device(rw_engine *eng, const int32_t *tape_dev, int32_t tape_steps, int32_t first, int32_t n_steps);
/* The same launches, with two of the engine's timing events (rw_event_record slots) riding on the dispatches themselves:
 * `start_slot` takes the START timestamp of the first launch, `stop_slot` the END timestamp of the last one
 * (hipExtLaunchKernel's start / stop events) — rw_event_elapsed_ms(start_slot, stop_slot) is then the device time of the
 * n_steps launches without the two marker packets that rw_event_record would put around them (≈9 µs on a 150 µs region).
 * n_steps >= 1, start_slot != stop_slot. */
int rw_step_tape_device_timed(rw_engine *eng, const int32_t *tape_dev, int32_t tape_steps, int32_t first, int32_t n_steps,
                              int32_t start_slot, int32_t stop_slot);

/* T consecutive steps from a device-resident action tape int32 [T][B][N] in ONE kernel launch: each
 * workgroup keeps its env chunk in LDS across the T steps, so per step only the actions are read and
 * obs / rewards / terminated written (rollout API; open-loop by construction —
 * the tape must exist before the launch).  Results are identical to T rw_step_device calls.  If `obs_tape` /
 * `reward_tape` / `terminated_tape` are non-NULL device pointers they receive every step's
 * outputs ([T][B][N][L] f32, [T][B][N] f32, [T][B] u8); otherwise only the last step's remain. */
int rw_step_many_device(rw_engine *eng, const int32_t *actions_dev, int32_t n_steps,
                        float *obs_tape, float *reward_tape, uint8_t *terminated_tape);

/* One call that enqueues a step on SEVERAL engines — the shards of a single-process multi-device env (one engine per GPU).  When
 * every engine sits on a device of its own, every engine but the first is launched by a thread that rw_multi_create starts
 * (bound to that engine's device, spinning briefly between rounds, asleep otherwise), so the launches are issued side by
 * side; engines that share a device are looped over by the calling thread (their launches serialise on the device's queue
 * anyway).  rw_multi_step_device(actions_dev[k] = engine k's device action array) returns when all launches are enqueued.  Results as for rw_step_device; the engines stay usable
 * on their own between rounds (never concurrently with a round).  Destroy the rw_multi before its engines. */
typedef struct rw_multi rw_multi;
int rw_multi_create(rw_engine **engines, int32_t n, rw_multi **out);
int rw_multi_step_device(rw_multi *m, const int32_t *const *actions_dev);
int rw_multi_destroy(rw_multi *m);

/* raw device memory for action / output tapes (callers without a GPU array library) */
int rw_device_malloc(rw_engine *eng, size_t bytes, void **dev_ptr);
int rw_device_free(rw_engine *eng, void *dev_ptr);
int rw_copy_to_device(rw_engine *eng, void *dev_dst, const void *host_src, size_t bytes);
int rw_copy_to_host(rw_engine *eng, void *host_dst, const void *dev_src, size_t bytes);

/* bring the derived views — RW_BUF_GRID and RW_BUF_AGENT_X .. RW_BUF_AGENT_DELIVERED — up to date with the steps enqueued so
 * far (three small kernels on the engine's stream; a no-op when nothing ran since the last refresh).  For callers that
 * hold a borrowed pointer / a zero-copy tensor of one of them. */
int rw_refresh_grid(rw_engine *eng);
/* Move the engine to another stream (a hipStream_t; NULL = the device's default stream): later calls enqueue there.  Work
 * already enqueued stays where it is — ordering the two streams is the caller's business (events).  An engine-owned stream is
 * drained and destroyed.  For callers that set up stream capture after the engine exists (WarehouseVecEnv.capture_loop). */
int rw_set_stream(rw_engine *eng, void *stream);
/* Tell the engine that steps ran which its host side did not see — a HIP graph holding rw_step_device launches was replayed
 * (replays run without host code) — so that the next rw_read / rw_get_buffer / rw_write of a derived view rebuilds it.  The
 * engine also notices by itself when one of its launches is being captured (hipStreamIsCapturing) and from then on rebuilds
 * the derived views on every request; this call is for graphs captured by other means. */
int rw_mark_views_stale(rw_engine *eng);

/* recompute RW_BUF_OBS from the current state (after rw_write of state buffers) */
int rw_refresh_obs(rw_engine *eng);

/* wait for everything enqueued; returns RW_ERR_INVALID_ACTION if any step since the last
 * rw_sync saw an out-of-range action (that action was executed as NOOP), else RW_OK */
int rw_sync(rw_engine *eng);

/* -- buffers ---------------------------------------------------------------------------- */
int rw_get_buffer(rw_engine *eng, int kind, void **dev_ptr, size_t *bytes);
/* synchronous copies between a buffer and host memory (state inspection / injection) */
int rw_read(rw_engine *eng, int kind, void *host_dst, size_t bytes);
/* the step() return tuple (:944-946) to host memory in one round trip: RW_BUF_OBS / REWARDS / TERMINATED (/ FEATURES) copied
 * back to back, one synchronisation; any pointer may be NULL (skipped).  `truncated` is always False (:942): nothing to read. */
int rw_read_outputs(rw_engine *eng, float *obs, float *rewards, uint8_t *terminated, float *features);
int rw_write(rw_engine *eng, int kind, const void *host_src, size_t bytes);
/* rebuild RW_BUF_GRID exactly like Warehouse._recalc_grid (:749-755) from explicit shelf
 * positions (host int32 [B][S][2] = (x,y) per shelf id; later ids overwrite earlier) and the
 * current agent positions.  This is how the reference's own tests inject state. */
int rw_recalc_grid(rw_engine *eng, const int32_t *shelf_xy, int32_t n_shelves);

/* -- snapshot / restore (checkpointing the batched env state) ------------ */
/* A snapshot is a device-resident copy of everything reset()/step() evolve: shelf layer, packed agent
 * records, queue, counters, PCG64 streams, pending-autoreset flags (the int32 views are derived from these).  Saving and restoring are device-to-
 * device copies on the engine's stream (tens of microseconds); restore also refreshes RW_BUF_OBS, so
 * the engine continues bit-identically from the saved point. */
typedef struct rw_snapshot rw_snapshot;
int rw_snapshot_create(rw_engine *eng, rw_snapshot **out);
int rw_snapshot_save(rw_engine *eng, rw_snapshot *snap);
int rw_snapshot_restore(rw_engine *eng, const rw_snapshot *snap);
int rw_snapshot_destroy(rw_engine *eng, rw_snapshot *snap);

/* -- introspection ---------------------------------------------------------------------- */
typedef struct rw_info {
    int32_t num_envs, grid_h, grid_w, n_agents, request_queue_size, n_shelves, obs_length;
    int32_t envs_per_workgroup, threads_per_workgroup, n_workgroups, lds_bytes;   /* of the per-step launches; the fused rollout may run on another
                             build of the same shape (13 .. 16 agents: per-step launches on 4-env workgroups below one round of workgroups and
                             between one and four, rollouts on 8-env ones; 17 .. 19 agents below 8192 envs likewise; rw_create's measured rule) */
    int32_t device_id, compute_units;
    int32_t specialised;  /* 1: a kernel build with this task's shapes folded in at compile time is in use */
    int32_t wave_priority; /* bit 0: the per-step launches raise their wavefronts' priority (s_setprio 3) from the start of the kernel to
                             the barrier behind the agent phases, so that the dependent chain in front of the first observation store is
                             not queued behind the neighbours' store phase on the same CU; bit 1: the fused rollouts do (every step of the
                             launch).  rw_create's measured rule: on while a step's observations stay under 200 MB (past that nothing moves),
                             the per-step launches of 13 .. 16 agents at sensor_range 1 excepted (A/B runs: RWARE_PRIO=0|1,
                             RWARE_PRIO_ROLLOUT=0|1 with RWARE_HOOKS=1).  A scheduling hint, never a different result.  (The slot was
                             `state_layout`, 0 since round 3.) */
    int32_t build_kind;   /* which build of the step kernel runs: 0 generic (every shape at run time), 1 exact-shape, 2 agent-count-
                             static (shapes + agent count folded in, request-queue length at run time), 3 size-static (grid folded in,
                             agent count and queue length at run time).  (Occupies what was alignment padding: same struct size.) */
    int64_t algorithmic_bytes_per_env_step; /* SURVEY.md §8(d) formula                           */
    char device_name[128];
    char arch_name[64];
    int32_t obs_stores_stream; /* 1: the per-step kernel stores the observation stream with the non-temporal hint
                                  (rw_stream_flags); the fused rollout kernel (rw_step_many_device) always stores cached */
    int32_t jit;               /* 0: an ahead-of-time build runs; 1: an exact-shape build compiled by this rw_create (hipRTC); 2: the
                                  same, read from the disk cache; -1: run-time specialisation was tried and is not in use (rw_jit_log) */
    int64_t engine_bytes_per_env_step; /* bytes this engine's state layout has to move per env-step: shelf shadow (1 or 2 B per
                                  cell) + packed agent records r/w + actions + queue + counters / flags + observation + rewards +
                                  terminated (+ messages r/w, IMAGE_DICT features).  The PMC traffic of a step is checked against
                                  it; bench.py prices `frac_engine` on it (<= 1 by construction)                               */
    int32_t stagger_ticks;     /* > 0: the k-th of the first eight workgroups a CU receives starts k * stagger_ticks * 10 ns late, so that
                                  the workgroups of a CU do not run their load / agent / store phases in lock-step.  rw_create's measured rule:
                                  13 .. 16 agents at sensor_range 1 — 550 ns at one round and from four rounds on; launches that do NOT run at
                                  raised wavefront priority (wave_priority bit 0 clear: steps of >= 200 MB of observations) — <= 12 agents 250 ns
                                  from two rounds of workgroups on, 13 .. 16 agents at sensor_range 2 400 ns up to two rounds; else 0: with the
                                  priority the delay is only a delay (A/B runs: RWARE_STAGGER_TICKS=n with RWARE_HOOKS=1; 0 = off).  A delay,
                                  never a different result */
    int32_t pipe_envs_per_workgroup; /* != 0: rw_step* launches run the chunk-pipelined persistent build with chunks of this many envs ... */
    int32_t pipe_workgroups;         /* ... on this many persistent workgroups (rw_stream_flags RW_PIPE_ON / RW_PIPE_OFF)              */
    int32_t stats;                   /* 1: RW_STATS_ON — RW_BUF_STAT_* are kept (was `reserved[1]`: same struct size)                 */
} rw_info;
int rw_get_info(const rw_engine *eng, rw_info *out);
/* what the run-time specialisation did for this engine: cache file / compile time, or why it is not in use ("" if not tried) */
const char *rw_jit_log(const rw_engine *eng);
/* the compile half of the run-time specialisation, without a device (build checks, cache warm-up on a login node): `shape` =
 * {sensor_range, H, W, N, Q, S, envs per workgroup, 256, msg_bits, wide shelf ids, observation kind (0 FLATTENED, 1 IMAGE,
 * 2 FLATTENED + messages), baked image layers, packed layer list, image_directional (-1 for FLATTENED), non-temporal stores};
 * returns the size of the gfx code object (compiled or found in the disk cache), -1 on failure (`log` says why). */
int64_t rw_jit_probe(const int32_t shape[15], const char *arch, char *log, size_t log_len);

/* On-device self-test (no engine needed; no reference counterpart — the reference is CPU Python): runs, on HIP device `device_id`,
 * the two toolchain / hardware facts the step kernels are written around — cross-lane exchange results compared as values in
 * registers of their own (a DPP move folded into a non-commutative instruction came out with swapped operands on gfx950), and an
 * LDS-DMA stage-in that is waited for explicitly in front of the workgroup barrier (a run-time compiled build once left it in flight)
 * — with code compiled into this library, so it runs where there is no hipcc.  RW_OK, or RW_ERR_SELFTEST with the counts in `log`.
 * RWARE_SELFTEST_BREAK=1 (with RWARE_HOOKS=1) runs the known-bad forms instead (tests: the guard has to be able to fail). */
int rw_selftest(int32_t device_id, char *log, size_t log_len);

/* numpy SeedSequence(seed) -> PCG64 initial state, as 6 uint64 in RW_BUF_RNG field order.
 * Pure host function (exposed so tests can check the seeding against numpy). */
int rw_seed_state(uint64_t seed, uint64_t out[6]);

/* -- timing on the engine's own stream (hipEvents), used by bench.py --------------------- */
int rw_event_record(rw_engine *eng, int32_t slot /* 0..7 */);
int rw_event_elapsed_ms(rw_engine *eng, int32_t slot_begin, int32_t slot_end, float *ms);

/* measurement aid (no reference counterpart): `n_launches` back-to-back launches of a kernel that only WRITES one step's
 * observations — the engine's launch geometry and store instruction, nothing else — timed with HIP events on the first / last launch:
 * the least any kernel producing this step's observations can take on this device (bench.py reports it beside the 8 TB/s roofline).
 * RW_BUF_OBS is refreshed afterwards (rw_refresh_obs).  Uses the engine's timing-event slots 6 and 7 (rw_event_record). */
int rw_debug_store_floor(rw_engine *eng, int32_t n_launches, float *ms_per_launch);

/* profiling aid: runs ONE step (device actions) with per-workgroup phase stamps taken from the
 * 100 MHz wall clock; host_out receives uint64 [n_workgroups][n_marks].  Call with host_out ==
 * NULL to query the two sizes first. */
int rw_debug_timeline(rw_engine *eng, const int32_t *actions_dev, uint64_t *host_out,
                      int32_t *n_workgroups, int32_t *n_marks);

int rw_abi_version(void);

#ifdef __cplusplus
}
#endif
#endif /* RWARE_HIP_H */
