// rware_phase_agents_reg.h — part of rw::rware_step_kernel (rware_kernels.h), included INSIDE the kernel body: AG — the agent phases with the exchange in registers (1 .. 19 agents; kCell: through the per-cell layer in LDS from 9 agents on)
// A textual unit, not a function: the phases share ~60 locals (LDS pointers, shapes, the agent lanes' registers), and every
// way of passing them that was tried — lambdas, always_inline or not — reschedules the kernels around it (round 5: +-10
// instructions per kernel, two 13/14-agent builds over a register cliff).  Splitting the text keeps every build's ISA.
    constexpr int KN = Cfg::kN, KG = 64 / KN, QS = (Cfg::kQcap + KN - 1) / KN;
    const int KQ = Cfg::kQrt ? Q : Cfg::kQ;  // (a compile-time constant unless the build reads the queue length at run time)
    static_assert(Cfg::kH * Cfg::kW < 0x8000, "cell indices are packed into 16 bits");
    for (int eb = wave * KG; eb < ne; eb += nw * KG) {  // wave-uniform
        const int g = lane / KN, a_idx = lane - g * KN;
        const bool mine = (g < KG) && (eb + g < ne);
        const int lane_base = (g < KG ? g : KG - 1) * KN;  // (the 64 % N idle tail lanes gather from the last group)
        const int e = mine ? eb + g : eb;  // keep every address in range for idle lanes
        const int base = e * KN, i = base + (mine ? a_idx : 0);
        CellT *gS = s_gs + e * HW;
        uint8_t *gA = s_ga + e * HW;
        int32_t *ev = s_envi + e * ENVI_W;
        const int ge = e0 + e;  // global env index
        // ---- own record, env flags and counters: from registers (kDirect, first step of the launch), else LDS read batch 1
        int ev_skip, ev_reset, ev_steps, ev_inact, x, y, d, carry, deliv, a_lds;
        if (kDirect && t == 0) {
            if constexpr (!kEarly) unpack_own();  // (kEarly: done in front of the stage-in barrier)
            ev_skip = ev_reset = r_flag; ev_steps = r_steps; ev_inact = r_inact;
            x = r_x; y = r_y; d = r_d; carry = r_carry; deliv = r_deliv; a_lds = r_act;
        } else {
            ev_skip = ev[ENVI_SKIP]; ev_reset = ev[ENVI_RESET]; ev_steps = ev[ENVI_STEPS]; ev_inact = ev[ENVI_INACTIVE];
            x = s_ax[i]; y = s_ay[i]; d = s_dir[i]; carry = s_carry[i]; deliv = s_deliv[i];
            a_lds = (t == 0) ? s_act[i * AM] : (int)ACT_NOOP;
        }
        const bool stepping = (op == OP_STEP) && mine && !ev_skip;
        int a = ACT_NOOP;
        if (!kEarly && mine) {
            if (stepping) a = (t == 0) ? a_lds : (act_prefetch ? a_pref : act_t[((size_t)ge * KN + a_idx) * AM]);
            if (act_prefetch && t + 1 < n_steps) a_pref = (act_t + la.act_stride)[(size_t)ge * KN + a_idx];
        }
        if constexpr (kMsg && kDirect) {  // agent.message[:] = action[1:] — for every agent, whatever its move does (:812)
            if (mine) {
                int msg = r_msg;  // an env that does not step keeps its stored messages (first step of the launch: registers)
                if (stepping) {
                    msg = 0;
#pragma unroll
                    for (int k = 0; k < KMW; ++k) {
                        const int v = (t == 0) ? r_mw[k] : act_t[((size_t)ge * KN + a_idx) * AM + 1 + k];
                        if (RW_RARE((unsigned)v > 1u)) atomicOr(p.status, STATUS_INVALID_ACTION);  // MultiDiscrete([5, 2, 2, ...])
                        msg |= (v & 1) << k;
                    }
                }
                if (stepping || t == 0) s_msg[i] = msg;
            }
        } else if (kMsg && mine && stepping) {
            int msg = 0;
            for (int k = 0; k < M; ++k) {
                const int v = (t == 0) ? s_act[i * AM + 1 + k] : act_t[((size_t)ge * KN + a_idx) * AM + 1 + k];
                if ((unsigned)v > 1u) atomicOr(p.status, STATUS_INVALID_ACTION);  // MultiDiscrete([5, 2, 2, ...])
                msg |= (v & 1) << k;
            }
            s_msg[i] = msg;
        }
        // ------------------------------------------------------------ P1: intent (:825-846), branch-free
        Intent in = kEarly ? early : intent_of(stepping, a, x, y, d);
        a = in.a;
        const int st = in.st, tg0 = in.tg0, tx0 = in.tx0, ty0 = in.ty0;
        RW_AG_MARK(TL_AG_RECORD, st, a);
        if constexpr (kCell) {  // the start-of-step agent layer (id | 0x80 if loaded; zeroed by the clear): where everybody stands
            if (mine && !ev_reset) gA[st] = (uint8_t)((a_idx + 1) | (carry ? 0x80 : 0));
            wave_sync();
        }
        // ---- LDS read batch 2 (the only one of the common kDirect step): the shelf layer at the target, under the agent and
        // on the first two goal cells (start-of-step values), the highway word of the agent's cell
        const int sh_tg = gS[tg0], shelf_here = gS[st], sh_g0 = gS[k_goal0], sh_g1 = gS[k_goal1];
        const uint32_t hw_word = s_hw[st >> 5];
        int qv[QS > 0 ? QS : 1];  // the queue slots this lane publishes in the requested-shelf bitmap
        const int eq = Cfg::kQrt ? __mul24(e, KQ) : e * KQ;  // row of the env in the LDS queue
#pragma unroll
        for (int q = 0; q < QS; ++q) {
            if (Cfg::kQrt && q * KN >= KQ) { qv[q] = 0; continue; }  // (scalar test: a slot group beyond the run-time queue length)
            qv[q] = s_queue[eq + (Cfg::kQrt ? max(min(a_idx + q * KN, KQ - 1), 0) : min(a_idx + q * KN, KQ - 1))];
        }
        int occ_w;
        if constexpr (kCell) {  // (one byte of the agent layer, read in the same batch)
            const int ag_tg = gA[tg0];
            occ_w = (ag_tg & 0x7f) ? ((((ag_tg & 0x7f) - 1) << 20) | ((ag_tg & 0x80) << 9)) : -1;
        } else {
            occ_w = kEarly ? in.occ_w : occupant_of(in, carry, a_idx, lane_base);  // (issued beside the LDS reads above)
        }
        if (kDirect && t == 0 && mine && a_idx == 0) {  // the env's leader lane publishes the flags and counters the other
            // phases read (from its registers; LDS stores issued while the reads above are in flight)
            ev[ENVI_STEPS] = r_steps; ev[ENVI_INACTIVE] = r_inact; ev[ENVI_RESET] = r_flag; ev[ENVI_SKIP] = r_flag;
            ev[ENVI_DONE] = 0; ev[ENVI_QDIRTY] = 0;
            if (r_flag) atomicOr(&s_misc[0], 1);
        }
        const int occ = occ_w >> 20;  // -1: nobody there
        const int occ_loaded = (occ_w >> 16) & 1 & ~(occ_w >> 31);
        // a standing shelf blocks a loaded agent (:836-846)
        const bool blocked = (carry != 0) & (tg0 != st) & (sh_tg != 0) & (occ_loaded == 0);
        RW_AG_MARK(TL_AG_CELLS, (int)blocked, sh_g0 + sh_g1);
        a = blocked ? (int)ACT_NOOP : a;
        const int tg = blocked ? st : tg0, tx = blocked ? x : tx0, ty = blocked ? y : ty0;
        // successor on the chain: agent index on the target cell, -1 empty, -2 == this agent is stationary
        const int nxt = (tg == st) ? -2 : occ;
        // Chains (an agent stepping onto a cell another agent stands on) are rare; when the wavefront has none, every
        // follower depth is 0 and a mover commits iff it wins its cell.
        const bool chains = wave_any(nxt >= 0);  // wave-uniform
        int depth = 0, lose = 0, commit = 0;
        if constexpr (kCell) {
            // ---- through LDS, O(1) per agent: chain links and contested-cell keys are published, follower depth by walking the
            // links with atomicMax (a chain of movers is short), the winner test looks at the four neighbours of the target
            // cell — whoever else wants that cell stands on one of them —, the chain walk chases pointers.
            if (mine) { s_nxt[i] = nxt; s_tgt[i] = (nxt != -2) ? tg : -1; }  // (s_depth was zeroed by the clear)
            wave_sync();
            if (chains) {
                if (nxt >= 0) {
                    int j = nxt, dd = 1;
                    while (j >= 0 && j != a_idx && dd <= KN && s_nxt[base + j] != -2) {
                        atomicMax(&s_depth[base + j], dd);
                        j = s_nxt[base + j];
                        ++dd;
                    }
                }
                wave_sync();
                depth = s_depth[i];
            }
            // winner of a contested cell: larger follower depth, then the LOWER agent id; only movers compete.  The start-of-step
            // agent layer says who stands on the four neighbours of my target; their published keys say whether they want it.
            {
                const bool okn[4] = {ty > 0, ty < H - 1, tx > 0, tx < W - 1};
                const int nbc[4] = {tg - W, tg + W, tg - 1, tg + 1};
                int kk[4];
                bool val[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {  // (read batch 1: an off-grid neighbour reads my own cell and is masked)
                    const int ida = gA[okn[q] ? nbc[q] : st] & 0x7f;
                    kk[q] = ida - 1;
                    val[q] = okn[q] & (ida != 0) & (kk[q] != a_idx);
                }
                int tk[4], dk[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) {  // (read batch 2)
                    const int jq = base + (val[q] ? kk[q] : 0);
                    tk[q] = s_tgt[jq];
                    dk[q] = s_depth[jq];
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    lose |= (val[q] & (tk[q] == tg) & ((dk[q] > depth) | ((dk[q] == depth) & (kk[q] < a_idx)))) ? 1 : 0;
                lose = (nxt != -2) ? lose : 0;
            }
            // ---- commit (:871-876)
            commit = (nxt == -1) ? (lose ^ 1) : ((nxt == -2) ? 1 : 0);
            if (chains) {  // walk the chain ahead: i -> nxt(i) -> ...
                if (mine) s_win[i] = lose ^ 1;
                wave_sync();
                if (nxt >= 0) {
                    int j = a_idx, hops = 0, ok = 1, cm = 0;
                    for (;;) {
                        ok &= s_win[base + j];
                        const int nj = s_nxt[base + j];
                        ++hops;
                        if (nj == -1) { cm = ok; break; }                  // drains into an empty cell
                        if (nj == a_idx) { cm = (hops >= 3) ? 1 : 0; break; }  // a cycle through me; the 2-swap is refused
                        if (s_nxt[base + nj] == -2) break;                 // blocked by a stationary agent
                        if (hops >= KN) break;                             // feeds a cycle it is not part of
                        j = nj;
                    }
                    commit = cm;
                }
            }
            wave_lds_order();  // every lane's reads of the start-of-step agent layer come before the first lane's update of it (P3)
        } else {
            // ------------------------------------------------------------ P2a: follower depth (longest chain of movers behind me)
            int nxv[KN];
            if (chains) {
                env_gather<KN>(nxt, lane_base, nxv);
                for (int it = 1; it < KN; ++it) {  // relaxation; a chain of movers has at most N - 1 links
                    int dv[KN];
                    env_gather<KN>(depth, lane_base, dv);
                    int nd = 0;
#pragma unroll
                    for (int k = 0; k < KN; ++k) nd = max(nd, (nxv[k] == a_idx) ? dv[k] + 1 : 0);
                    nd = (nxt != -2) ? nd : 0;  // only movers carry a depth
                    const bool changed = nd != depth;
                    depth = nd;
                    if (!wave_any(changed)) break;  // wave-uniform (agents on a cycle never settle: their depth is not used)
                }
            }
            // ------------------------------------------------------------ P2b: winner per contested cell
            // larger follower depth wins, then the LOWER agent id; only movers compete.  One word per agent, target cell above
            // the priority (depth << IB | 2^IB - 1 - index; IB = 4 bits up to 16 agents, 5 beyond): agent k beats me iff the cell
            // fields agree and its word is the larger one.  A stationary agent announces a cell nobody can target (0x1fff00 | index)
            // and so neither beats nor is beaten.
            constexpr int IB = KN <= 16 ? 4 : 5, PW = 2 * IB;  // (a depth is at most N - 1: the same width)
            const uint32_t vme = (nxt != -2) ? ((uint32_t)tg << PW) | ((uint32_t)depth << IB) | (uint32_t)((1 << IB) - 1 - a_idx)
                                             : (0x1fff00u | (uint32_t)a_idx) << PW;
            int kv[KN];
            env_gather<KN>((int)vme, lane_base, kv);
            // (A subtract-and-running-minimum form of this test — one compare at the end — passed the host emulation and failed a
            //  golden trace on the GPU: hipcc folds the DPP move into `v_subrev_u32_dpp` and the result came out with the
            //  operands swapped; profiles/tools/dpp_subrev_probe.hip.  Keep the exchange results in registers of their own.)
#pragma unroll
            for (int k = 0; k < KN; ++k)  // (bitwise on purpose: no short-circuit branches)
                lose |= ((((uint32_t)kv[k] ^ vme) < (1u << PW)) & ((uint32_t)kv[k] > vme)) ? 1 : 0;
            // ------------------------------------------------------------ P2c: commit (:871-876)
            commit = (nxt == -1) ? (lose ^ 1) : ((nxt == -2) ? 1 : 0);
            if (chains) {  // walk the chain ahead: i -> nxt(i) -> ... on the gathered links
                // every agent's (nxt + 2 | win << LB) as one field of a word every lane of the env holds: following a link is
                // a shift and a mask (a register array indexed by a run-time agent index would live in scratch memory).
                // N <= 6: 3 + 1 bits per agent in 32 bits; 7 <= N <= 12: 4 + 1 bits per agent in 64 bits (two OR-reductions);
                // 13 <= N <= 19: 5 + 1 bits per agent in 128 bits — ONE gather of every agent's field (N cross-lane moves), the word
                // assembled in each lane with compile-time shifts (four OR-reductions would be 4 N moves).
                constexpr int LB = KN <= 6 ? 3 : KN <= 12 ? 4 : 5, FW = LB + 1;
                constexpr uint32_t LM = (1u << LB) - 1u;
                using links_t = typename pick_type<KN <= 6, uint32_t, typename pick_type<KN <= 12, uint64_t, u128>::type>::type;
                links_t links;
                {
                    const uint32_t own_field = (uint32_t)((nxt + 2) | ((lose ^ 1) << LB));
                    if constexpr (KN <= 6) {
                        links = (links_t)(uint32_t)env_or<KN>((int)(own_field << (FW * a_idx)), lane_base);
                    } else if constexpr (KN <= 12) {
                        const uint64_t own = (uint64_t)own_field << (FW * a_idx);
                        const uint32_t lo = (uint32_t)env_or<KN>((int)(uint32_t)own, lane_base);
                        const uint32_t hi = (uint32_t)env_or<KN>((int)(uint32_t)(own >> 32), lane_base);
                        links = (links_t)(((uint64_t)hi << 32) | lo);
                    } else {
                        int fv[KN];
                        env_gather<KN>((int)own_field, lane_base, fv);
                        links = 0;
#pragma unroll
                        for (int k = 0; k < KN; ++k) links |= (links_t)(uint32_t)fv[k] << (FW * k);
                    }
                }
                int j = a_idx, hops = 0, ok = 1, cm = 0;
                bool done = nxt < 0;
#pragma unroll
                for (int h = 0; h < KN; ++h) {
                    const uint32_t ent = (uint32_t)(links >> (FW * j)) & ((1u << FW) - 1u);
                    const int nj = (int)(ent & LM) - 2;
                    ok &= (int)(ent >> LB);
                    ++hops;
                    const int nnj = (int)((uint32_t)(links >> (FW * (nj & (KN <= 6 ? 7 : (nj < 0 ? 0 : 31))))) & LM) - 2;  // nxt of the successor (not used when nj < 0)
                    const bool to_empty = nj == -1;                     // drains into an empty cell
                    const bool back = nj == a_idx;                      // a cycle through me; the 2-swap is refused
                    const bool stuck = (nj >= 0) & (nnj == -2);         // blocked by a stationary agent
                    cm = (!done & to_empty) ? ok : cm;
                    cm = (!done & !to_empty & back) ? ((hops >= 3) ? 1 : 0) : cm;
                    done = done | to_empty | back | stuck | (hops >= KN);  // hops == N: feeds a cycle it is not part of
                    j = (nj >= 0) ? nj : j;
                    if (h + 1 < KN && !wave_any(!done)) break;  // wave-uniform: the usual chain is one or two links long
                }
                commit = (nxt >= 0) ? cm : commit;
            }
        }
        // ------------------------------------------------------------ P3: apply (:878-899)
        RW_AG_MARK(TL_AG_WINNERS, commit, lose);
        a = commit ? a : (int)ACT_NOOP;  // a failed mover does nothing (:875)
        const bool moved = (a == ACT_FORWARD) & (tg != st);
        x = moved ? tx : x;
        y = moved ? ty : y;
        // wraplist [UP, RIGHT, DOWN, LEFT] (:119): RIGHT 0->3->1->2->0, LEFT 0->2->1->3->0
        const int right = (0x1023 >> (4 * d)) & 0xF;  // d: 0->3, 1->2, 2->0, 3->1
        const int left = (0x0132 >> (4 * d)) & 0xF;   // d: 0->2, 1->3, 2->1, 3->0
        d = (a == ACT_RIGHT) ? right : ((a == ACT_LEFT) ? left : d);
        // TOGGLE_LOAD (:886-899): pick up the shelf under the agent, or put the carried one down off the highways
        const bool toggle = (a == ACT_TOGGLE);
        const bool drop = toggle & (carry != 0) & (((hw_word >> (st & 31)) & 1u) == 0u);
        const bool pick = toggle & (carry == 0) & (shelf_here != 0);
        const float rew = (drop & (deliv != 0) & (k_reward_type == REW_TWO_STAGE)) ? 0.5f : 0.0f;
        const bool mcar = moved & (carry != 0);  // a loaded mover drags its shelf along the shelf layer
        deliv = drop ? 0 : deliv;
        carry = drop ? 0 : (pick ? shelf_here : carry);
        // ---- results to LDS (stores only; nothing below waits for them on the common path)
        if (kDirect ? mine : stepping) { s_ax[i] = x; s_ay[i] = y; s_dir[i] = d; s_carry[i] = carry; s_deliv[i] = deliv; }
        if (mine) s_rew[i] = rew;  // every agent of the chunk gets its reward slot
        if (mine) s_mv[i] = moved ? (st | (tg << 16)) : -1;  // which two cells changed (write-back hand-off)
        if (mcar) gS[st] = 0;  // incremental _recalc_grid (:749-755): clear phase ...
        if constexpr (kCell) { if (moved) gA[st] = 0; }  // (kCell: the layer holds the start-of-step marks — a mover's goes first)
        wave_lds_order();
        if (mcar) gS[tg] = (CellT)carry;  // ... then set phase, for the whole wavefront in this order
        // the agent layer was zeroed at the start of the step: final position only (id | 0x80 if loaded)
        if (mine && !ev_reset) gA[moved ? tg : st] = (uint8_t)((a_idx + 1) | (carry ? 0x80 : 0));
        // ------------------------------------------------------------ P5: goals, rewards, termination (:903-942)
        // Is there a shelf on a goal cell after the moves?  From registers: a loaded mover that arrived there, or the
        // start-of-step shelf unless a loaded mover took it away.  (More than two goal cells: always take the LDS path.)
        int gflags;
        if constexpr (kCell) {  // (four ballots instead of N cross-lane moves)
            gflags = (env_any<KN>(mcar & (tg == k_goal0), lane_base) ? 1 : 0) | (env_any<KN>(mcar & (tg == k_goal1), lane_base) ? 2 : 0) |
                     (env_any<KN>(mcar & (st == k_goal0), lane_base) ? 4 : 0) | (env_any<KN>(mcar & (st == k_goal1), lane_base) ? 8 : 0);
        } else {
            gflags = env_or<KN>(mcar ? ((tg == k_goal0 ? 1 : 0) | (tg == k_goal1 ? 2 : 0) | (st == k_goal0 ? 4 : 0) |
                                         (st == k_goal1 ? 8 : 0)) : 0, lane_base);
        }
        // bit g of `on_goal`: a shelf stands on goal g after the moves — a loaded mover arrived (gflags bits 0, 1), or the
        // start-of-step shelf is still there (bits 2, 3 say a loaded mover took it away).  Integer arithmetic on purpose.
        const int had = min(sh_g0, 1) | (min(sh_g1, 1) << 1);
        const int on_goal = (gflags | (had & ~(gflags >> 2))) & (k_n_goals > 1 ? 3 : 1);
        const bool goal_hit = (on_goal != 0) | (k_n_goals > 2);
        const bool leader = stepping && a_idx == 0;
        RW_AG_MARK(TL_AG_APPLIED, (int)goal_hit, (int)moved);
        if (RW_RARE(wave_any(leader && goal_hit))) {  // wave-uniform; a delivery may be due: the LDS path
            wave_sync();
            if (leader) {
                goals_and_termination(e, ge, base, ev, gS, gA);
            }
            wave_sync();
#pragma unroll
            for (int q = 0; q < QS; ++q) {  // a request may have been replaced
                if (Cfg::kQrt && q * KN >= KQ) continue;
                qv[q] = s_queue[eq + (Cfg::kQrt ? max(min(a_idx + q * KN, KQ - 1), 0) : min(a_idx + q * KN, KQ - 1))];
            }
        } else if (leader) {  // nothing on a goal: counters and termination from registers
            const int inact = ev_inact + 1, steps = ev_steps + 1;
            const int done = ((k_max_inactivity && inact >= k_max_inactivity) || (k_max_steps && steps >= k_max_steps)) ? 1 : 0;
            ev[ENVI_INACTIVE] = inact;
            ev[ENVI_STEPS] = steps;
            ev[ENVI_DONE] = done;
            if (done && k_autoreset == AR_SAME_STEP) {
                ev[ENVI_RESET] = 1;
                atomicOr(&s_misc[0], 1);
            }
        }
        RW_AG_MARK(TL_AG_GOALS, 0, 0);
        // requested-shelf bitmap of the (post-step) queue.  Envs that reset in this launch are included: RS clears and
        // rebuilds their bitmap.
        if (mine) {
#pragma unroll
            for (int q = 0; q < QS; ++q) {
                if (Cfg::kQrt && q * KN >= KQ) continue;  // (scalar)
                if (a_idx + q * KN < KQ) atomicOr(&s_req[e * SW + (qv[q] >> 5)], 1u << (qv[q] & 31));
            }
        }
    }
