// rware_phase_agents_lds.h — part of rw::rware_step_kernel (rware_kernels.h), included INSIDE the kernel body: AG — the agent phases through LDS arrays under wave-local syncs (any agent count, run-time shapes)
// A textual unit, not a function: the phases share ~60 locals (LDS pointers, shapes, the agent lanes' registers), and every
// way of passing them that was tried — lambdas, always_inline or not — reschedules the kernels around it (round 5: +-10
// instructions per kernel, two 13/14-agent builds over a register cliff).  Splitting the text keeps every build's ISA.
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
