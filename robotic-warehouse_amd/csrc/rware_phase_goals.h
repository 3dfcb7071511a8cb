// rware_phase_goals.h — part of rw::rware_step_kernel (rware_kernels.h), included INSIDE the kernel body: P5 — goals, request replacement (numpy-exact draw), rewards, counters, termination: one env, run by its leader lane
// A textual unit, not a function: the phases share ~60 locals (LDS pointers, shapes, the agent lanes' registers), and every
// way of passing them that was tried — lambdas, always_inline or not — reschedules the kernels around it (round 5: +-10
// instructions per kernel, two 13/14-agent builds over a register cliff).  Splitting the text keeps every build's ISA.
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
