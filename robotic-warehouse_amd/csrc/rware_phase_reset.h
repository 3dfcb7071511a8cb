// rware_phase_reset.h — part of rw::rware_step_kernel (rware_kernels.h), included INSIDE the kernel body: RS — on-device reset of flagged envs (numpy-exact draws) and, SAME_STEP, the terminal observation written first (rare path)
// A textual unit, not a function: the phases share ~60 locals (LDS pointers, shapes, the agent lanes' registers), and every
// way of passing them that was tried — lambdas, always_inline or not — reschedules the kernels around it (round 5: +-10
// instructions per kernel, two 13/14-agent builds over a register cliff).  Splitting the text keeps every build's ISA.
    // ---------------------------------------------------------------- RS: reset flagged envs (:757-802)
    if (RW_RARE(s_misc[0] != 0)) {  // workgroup-uniform; rare
        if (RW_RARE(stats_on) && k_autoreset == AR_SAME_STEP) {  // the terminating step's events, while the arrays still hold that step
            count_events(true, tid, T);
            lds_barrier();
        }
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
