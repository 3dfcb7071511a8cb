// rware_phase_gather.h — part of rw::rware_step_kernel (rware_kernels.h), included INSIDE the kernel body: OS + P7 — the self part of the observation and the window gather into ONE bit string per workgroup (FLATTENED, messages, IMAGE)
// A textual unit, not a function: the phases share ~60 locals (LDS pointers, shapes, the agent lanes' registers), and every
// way of passing them that was tried — lambdas, always_inline or not — reschedules the kernels around it (round 5: +-10
// instructions per kernel, two 13/14-agent builds over a register cliff).  Splitting the text keeps every build's ISA.
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
