/* rware_c_example.c — the C-ABI boundary (include/rware_hip.h) used from plain C, no Python, no torch.
 *
 *   gcc -std=c11 -O2 -Iinclude examples/rware_c_example.c -Lrobotic-warehouse_amd/csrc -lrware_hip \
 *       -Wl,-rpath,$PWD/robotic-warehouse_amd/csrc -o /tmp/rware_c_example
 *   /tmp/rware_c_example [num_envs] [steps] [seed]
 *
 * What a binding in any host language does: build the warehouse layout as plain arrays exactly like
 * Warehouse._make_layout_from_params (rware/warehouse.py:294-326), fill an rw_config (the constructor's arguments, :146-170),
 * rw_create, rw_reset with per-env seeds (env i <- seed + i, the Gymnasium vector convention), then rw_step with HOST actions and
 * rw_read_outputs per step — the reference's `obs, rewards, done, truncated, info = env.step(actions)`.  Task: rware-tiny-2ag
 * (shelf_columns 3, shelf_rows 1, column_height 8, 2 agents, 2 requests).  Actions come from a small LCG so that a test can replay them.
 * Prints one line per run: an FNV-1a checksum over every step's observations, rewards and flags, the reward sum and the episode ends
 * — tests/test_gpu_parity.py compares them with what the Python layer gets on the same seeds and actions. */
#include <inttypes.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "rware_hip.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc_ = (call);                                                                             \
        if (rc_ != RW_OK) {                                                                           \
            fprintf(stderr, "%s failed: %d (%s)\n", #call, rc_, rw_last_error(eng));                  \
            return 1;                                                                                 \
        }                                                                                             \
    } while (0)

static uint64_t fnv1a(uint64_t h, const void *p, size_t n) {
    const unsigned char *b = (const unsigned char *)p;
    for (size_t i = 0; i < n; ++i) { h ^= b[i]; h *= 1099511628211ULL; }
    return h;
}

int main(int argc, char **argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 64, steps = argc > 2 ? atoi(argv[2]) : 50;
    const uint64_t seed = argc > 3 ? strtoull(argv[3], NULL, 10) : 7;
    enum { COLS = 3, ROWS = 1, HEIGHT = 8, N = 2, Q = 2, R = 1 };
    const int H = (HEIGHT + 1) * ROWS + 2, W = 3 * COLS + 1, mid = W / 2;
    /* highways (:302-318): every third column, every (column_height + 1)-th row, the bottom row, the two middle columns near the bottom */
    uint8_t *highways = (uint8_t *)calloc((size_t)H * W, 1);
    for (int y = 0; y < H; ++y)
        for (int x = 0; x < W; ++x)
            highways[y * W + x] = (x % 3 == 0) || (y % (HEIGHT + 1) == 0) || (y == H - 1) ||
                                  ((y > H - (HEIGHT + 3)) && (x == mid - 1 || x == mid));
    const int32_t goals_xy[4] = {mid - 1, H - 1, mid, H - 1}; /* (:303-306), list order == reward order */

    rw_engine *eng = NULL;
    rw_config cfg;
    memset(&cfg, 0, sizeof cfg);
    cfg.abi_version = RW_ABI_VERSION;
    cfg.num_envs = B;
    cfg.grid_h = H; cfg.grid_w = W;
    cfg.n_agents = N; cfg.sensor_range = R; cfg.request_queue_size = Q;
    cfg.max_inactivity_steps = 0; cfg.max_steps = 30;            /* (short episodes: the run crosses autoresets) */
    cfg.reward_type = RW_REWARD_INDIVIDUAL;
    cfg.autoreset_mode = RW_AUTORESET_NEXT_STEP;
    cfg.n_goals = 2;
    cfg.device_id = 0;
    cfg.highways = highways; cfg.goals_xy = goals_xy;
    CHECK(rw_create(&cfg, &eng));

    rw_info info;
    CHECK(rw_get_info(eng, &info));
    const int L = info.obs_length;
    uint64_t *seeds = (uint64_t *)malloc(sizeof(uint64_t) * (size_t)B);
    for (int e = 0; e < B; ++e) seeds[e] = seed + (uint64_t)e;
    CHECK(rw_reset(eng, seeds, NULL));

    float *obs = (float *)malloc(sizeof(float) * (size_t)B * N * L), *rew = (float *)malloc(sizeof(float) * (size_t)B * N);
    uint8_t *term = (uint8_t *)malloc((size_t)B);
    int32_t *act = (int32_t *)malloc(sizeof(int32_t) * (size_t)B * N);
    CHECK(rw_read_outputs(eng, obs, NULL, NULL, NULL)); /* the reset observation */
    uint64_t h = fnv1a(14695981039346656037ULL, obs, sizeof(float) * (size_t)B * N * L);
    uint32_t lcg = (uint32_t)seed * 2654435761u + 12345u;
    double reward_sum = 0.0;
    long ends = 0;
    for (int t = 0; t < steps; ++t) {
        for (int i = 0; i < B * N; ++i) {
            lcg = lcg * 1664525u + 1013904223u;
            const uint32_t r = (lcg >> 16) % 10u; /* FORWARD half of the time, the rest spread over the other actions */
            act[i] = r < 5 ? RW_FORWARD : (int32_t)(r - 5u);  /* rw_action */
        }
        CHECK(rw_step(eng, act));
        CHECK(rw_read_outputs(eng, obs, rew, term, NULL));
        h = fnv1a(h, obs, sizeof(float) * (size_t)B * N * L);
        h = fnv1a(h, rew, sizeof(float) * (size_t)B * N);
        h = fnv1a(h, term, (size_t)B);
        for (int i = 0; i < B * N; ++i) reward_sum += rew[i];
        for (int e = 0; e < B; ++e) ends += term[e];
    }
    CHECK(rw_sync(eng)); /* a sticky device-side error (an action out of range) would surface here */
    printf("rware_c_example envs=%d steps=%d seed=%" PRIu64 " obs_length=%d build_kind=%d checksum=%016" PRIx64 " reward_sum=%.1f episode_ends=%ld device=%s\n",
           B, steps, seed, L, info.build_kind, h, reward_sum, ends, info.device_name);
    CHECK(rw_destroy(eng));
    free(highways); free(seeds); free(obs); free(rew); free(term); free(act);
    return 0;
}
