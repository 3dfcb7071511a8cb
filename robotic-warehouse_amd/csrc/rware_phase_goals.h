// rware_phase_goals.h — part of rw::rware_step_kernel (rware_kernels.h), included INSIDE the kernel body: P5 — goals, request replacement (numpy-exact draw), rewards, counters, termination: one env, run by its leader lane; count_events (RW_STATS_ON)
// A textual unit, not a function: the phases share ~60 locals (LDS pointers, shapes, the agent lanes' registers), and every
// way of passing them that was tried — lambdas, always_inline or not — reschedules the kernels around it (round 5: +-10
// instructions per kernel, two 13/14-agent builds over a register cliff).  Splitting the text keeps every build's ISA.
    // P5 of one env, run by its leader lane once the moves are applied (:903-942): goals in list order, request replacement
    // with the numpy-exact draw, rewards, the env's counters and termination.  ONE copy, used by both agent-phase
    // implementations below (each kernel instantiation has exactly one call site, so it is inlined there).
    auto goals_and_termination = [&](int e, int ge, int base, int32_t *ev, CellT *gS, uint8_t *gA) {
        int32_t *q = s_queue + e * Q;
        bool delivered = false;
        int n_deliv = 0;  // (only read by the RW_STATS_BUILD kernels)
        for (int gi = 0; gi < k_n_goals; ++gi) {  // in list order (:904)
            const int cell = gi == 0 ? k_goal0 : gi == 1 ? k_goal1 : p.goal_cells[gi];
            const int sid = gS[cell];
            if (!sid) continue;
            int slot = -1;  // first queue slot holding sid; all Q entries read in one LDS batch (no early exit)
            for (int k = Q - 1; k >= 0; --k) slot = (q[k] == sid) ? k : slot;
            if (slot < 0) continue;
            delivered = true;
            ++n_deliv;
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
        if (stats_on) ev[ENVI_NDELIV] = n_deliv;
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
    // count_events (RW_STATS_BUILD kernels with RW_STATS_ON; off the common path everywhere): adds the step's deliveries and failed moves of the chunk's envs
    // to the per-env running totals in HBM.  Nothing of it lives in the agent phases: a delivery count is left in ENVI_NDELIV by the
    // (rare) goal path, and a failed move is re-derived here from what the write-back holds anyway — the agent asked for FORWARD
    // (its action, re-read: an L2 hit), did not move (s_mv) and the cell ahead is inside the grid (a wall-clamped FORWARD is a
    // self-target the reference leaves alone, :105-112): the shelf-block cancel (:843-846) or a lost resolution (:871-876).
    //   terminal == false  the envs this launch stepped and did not reset (from the write-back, role 1);
    //   terminal == true   SAME_STEP autoreset: the envs this step terminated, before RS overwrites their arrays (from RS).
    auto count_events = [&](bool terminal, int first, int stride) {
        if (op != OP_STEP) return;
        for (int i = first; i < nea; i += stride) {
            const int e = rw_div18(i, mN);
            const int32_t *ev = s_envi + e * ENVI_W;
            if (ev[ENVI_SKIP] || (terminal ? !(ev[ENVI_DONE] && ev[ENVI_RESET]) : ev[ENVI_RESET] != 0)) continue;
            const int x = s_ax[i], y = s_ay[i], d = s_dir[i];
            const bool ahead = d == DIR_UP ? y > 0 : d == DIR_DOWN ? y < H - 1 : d == DIR_LEFT ? x > 0 : x < W - 1;
            if (act_t[((size_t)e0 * N + i) * AM] == ACT_FORWARD && s_mv[i] < 0 && ahead) atomicAdd(p.stat_failed_moves + (e0 + e), 1);
        }
        for (int e = first; e < ne; e += stride) {
            const int32_t *ev = s_envi + e * ENVI_W;
            if (ev[ENVI_SKIP] || (terminal ? !(ev[ENVI_DONE] && ev[ENVI_RESET]) : ev[ENVI_RESET] != 0)) continue;
            if (ev[ENVI_INACTIVE] == 0) atomicAdd(p.stat_deliveries + (e0 + e), ev[ENVI_NDELIV]);
        }
    };
