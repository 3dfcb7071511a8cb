// rware_phase_stage_in.h — part of rw::rware_step_kernel (rware_kernels.h), included INSIDE the kernel body: P0 — scratch clear, the agent lanes' own-record loads, intent / occupant helpers, the classic flow's stage-in DMA and barrier
// A textual unit, not a function: the phases share ~60 locals (LDS pointers, shapes, the agent lanes' registers), and every
// way of passing them that was tried — lambdas, always_inline or not — reschedules the kernels around it (round 5: +-10
// instructions per kernel, two 13/14-agent builds over a register cliff).  Splitting the text keeps every build's ISA.
    // ---------------------------------------------------------------- P0: stage the env chunk
    auto clear_scratch = [&]() {  // agent layer, depth, obs bit string, request bitmap
        int4 *z = reinterpret_cast<int4 *>(smem + lo.ga);
        const int nz = (lo.zero_end - lo.ga) >> 2;
        for (int i = tid; i < nz; i += T) z[i] = int4{0, 0, 0, 0};
        for (int i = tid; i < nea; i += T) s_depth[i] = 0;
        for (int i = tid; i < nea * OW + 4; i += T) s_obits[i] = 0u;
        for (int i = tid; i < ne * SW; i += T) s_req[i] = 0u;
        if (tid == 0) s_misc[0] = 0;
    };
    // kDirect: every agent lane fetches ITS OWN record (and its env's flags and counters) from HBM straight into registers,
    // first thing in the kernel: the loads fly beside the clear and the stage-in DMA, are complete at the barrier that
    // drains the DMA, and the agent phases start without an LDS read.  The LDS copies the later phases read (window
    // gather, write-back) are written by the agent lanes together with their results.
    // which envs reset in this launch: OP_RESET — the caller's mask (all-ones when none was given); OP_STEP — the pending-reset
    // bit of the counter record (NEXT_STEP autoreset); OP_OBS — none
    const RW_GLOBAL uint8_t *const q_mask = as_global(la.reset_mask);
    auto flag_of = [&](int cnt_x, int mask_byte) -> int { return op == OP_OBS ? 0 : op == OP_RESET ? mask_byte : (int)((uint32_t)cnt_x >> 31); };
    int r_x = 0, r_y = 0, r_d = 0, r_carry = 0, r_deliv = 0, r_act = ACT_NOOP, r_flag = 0, r_steps = 0, r_inact = 0;
    int r_cx = 0, r_mask = 0;  // (the counter record's first word and the reset-mask byte as loaded: decoded in unpack_own)
    uint32_t r_rec = 0;
    auto unpack_own = [&]() {  // (W is a compile-time constant in the builds that use this)
        r_y = rec_cell(r_rec) / W; r_x = rec_cell(r_rec) - r_y * W;
        r_d = rec_dir(r_rec); r_carry = rec_carry(r_rec); r_deliv = rec_deliv(r_rec);
        r_flag = flag_of(r_cx, r_mask); r_steps = r_cx & 0x7fffffff;
    };
    constexpr int KMW = (kMsg && Cfg::kM) ? Cfg::kM : 1;
    int r_mw[KMW] = {};   // the action's message words (_MSG builds), r_msg: the agent's stored message
    int r_msg = 0;
    if constexpr (kDirect && !kPipe) {
        constexpr int KN = Cfg::kN, KG = 64 / KN;
        static_assert(Cfg::kE <= (Cfg::kT / 64) * KG, "every env of the chunk needs its own agent lanes");
        const int g = lane / KN, a_idx = lane - g * KN, le = wave * KG + g;
        if (g < KG && le < Cfg::kE) {
            const int ge = e0 + le;
            const size_t gi = (size_t)ge * KN + a_idx;
            r_rec = q_rec[gi];  // ONE load per agent: the packed record (unpacked where it is first needed — unpack_own —
                                // so that nothing up here waits for it: the clear and the DMA issue run under its latency)
            if (op == OP_STEP) r_act = la.actions[gi * AM];
            if constexpr (kMsg) {
                r_msg = as_global(p.amsg)[gi];
                if (op == OP_STEP)
#pragma unroll
                    for (int k = 0; k < KMW; ++k) r_mw[k] = la.actions[gi * AM + 1 + k];
            }
            const Cnt c = cnt_load(ge);  // ONE 8-byte load: steps, pending-reset bit, inactive — decoded where the record is
            r_cx = c.x;                  // (unpack_own), so that nothing up here waits for it
            r_inact = c.y;
            if (op == OP_RESET) r_mask = (int)q_mask[ge];
        }
    }
    // P1 of the agent phases as two pieces, so that the exact-shape per-step kernels can run them BEFORE the stage-in barrier
    // (kEarly, below): intent (:825-834, everything that needs only the agent's own record and action) and the occupant
    // of the target cell (a cross-lane exchange of the agents' positions).
    struct Intent { int a, st, tg0, tx0, ty0, occ_w; };
    constexpr int KNX = kRegAG ? Cfg::kN : 1;
    auto intent_of = [&](bool stepping, int a, int x, int y, int d) -> Intent {
        if (RW_RARE(stepping && (unsigned)a > 4u)) atomicOr(p.status, STATUS_INVALID_ACTION);  // Action(a) raises (:814); runs as NOOP
        a = (stepping && (unsigned)a <= 4u) ? a : (int)ACT_NOOP;
        const int fwd = (a == ACT_FORWARD) ? 1 : 0;
        const int dx = fwd & ((d == DIR_RIGHT) ? 1 : 0), dxn = fwd & ((d == DIR_LEFT) ? 1 : 0);
        const int dy = fwd & ((d == DIR_DOWN) ? 1 : 0), dyn = fwd & ((d == DIR_UP) ? 1 : 0);
        const int tx0 = min(max(x + dx - dxn, 0), W - 1);  // clamped at the walls (:105-112)
        const int ty0 = min(max(y + dy - dyn, 0), H - 1);
        return Intent{a, y * W + x, ty0 * W + tx0, tx0, ty0, -1};
    };
    // who stands on my target cell, and is it loaded: every agent announces (cell | loaded << 16 | index << 20)
    auto occupant_of = [&](const Intent &in, int carry, int a_idx, int lane_base) -> int {
        int pkv[KNX];
        env_gather<KNX>(in.st | (carry ? 0x10000 : 0) | (a_idx << 20), lane_base, pkv);
        int occ_w = -1;
#pragma unroll
        for (int k = 0; k < KNX; ++k) occ_w = ((pkv[k] & 0xffff) == in.tg0) ? pkv[k] : occ_w;
        return occ_w;
    };
    // kEarly (exact-shape per-step kernels): the agent wavefront does not take part in the stage-in DMA; its record loads
    // were the first thing it issued, so they are back while the other wavefronts' DMA is still in flight — it computes
    // intent and occupant in that window, before the barrier that everything else of the agent phases has to wait for.
    constexpr bool kEarly = kDirect && !kRollout && Cfg::kT >= 128 && !kPipe;
    Intent early{ACT_NOOP, 0, 0, 0, 0, -1};
    // builds that stage the agents through LDS: the DMA put the chunk's packed records into the `ax` slot; every thread
    // unpacks its agents in place (reads its own slot before it overwrites it) into the five per-agent LDS arrays
    auto unpack_records = [&]() {
        for (int i = tid; i < nea; i += T) {
            const uint32_t r = (uint32_t)s_ax[i];
            const int c = rec_cell(r), y = c / W;
            s_ax[i] = c - y * W; s_ay[i] = y; s_dir[i] = rec_dir(r); s_carry[i] = rec_carry(r); s_deliv[i] = rec_deliv(r);
        }
    };
    if constexpr (!kPipe) {
    clear_scratch();  // before the DMA: hipcc orders any later LDS write behind an in-flight LDS-DMA (vmcnt) — and the clear is not on the
                      // critical path anyway (DMA first, clear underneath through stores the compiler does not see: measured, round 4, +-0)
    RW_MARK(TL_ZEROED);
    if constexpr (Cfg::kE != 0) {
        // Static build: every DMA destination is contiguous in LDS and every source chunk is a whole
        // number of 16-byte pieces, so the chunk is ONE linear stream — thread t moves LDS piece t;
        // its HBM source is picked from the segment table (all pointers fetched in one scalar batch).
        static_assert((Cfg::kE * Cfg::kN) % 4 == 0 && (Cfg::kE * Cfg::kQcap) % 4 == 0 && Cfg::kE % 4 == 0, "chunk not 16-byte granular");
        static_assert(!kMsg || Cfg::kN == 0 || Cfg::kM != 0, "an exact-shape _MSG build needs its communication bits at compile time");
        static_assert((Cfg::kE * Cfg::kH * Cfg::kW * (int)sizeof(CellT)) % 16 == 0, "shelf chunk not 16-byte granular");
        const RW_GLOBAL char *src[12] = {
            as_bytes(g_shadow + (size_t)e0 * HW),
            // the chunk's packed records go to the `ax` slot and are unpacked in place behind the barrier (unpack_records);
            // the ay / dir / carry / deliv slots have no DMA source any more (entries 2..5 are skipped below)
            as_bytes(q_rec + (size_t)e0 * N), nullptr, nullptr, nullptr, nullptr,
            op == OP_STEP ? as_bytes(as_global(la.actions) + (size_t)e0 * N * AM) : as_bytes(q_rec + (size_t)e0 * N),
            as_bytes(q_queue + (size_t)e0 * Q), as_bytes(q_hw),
            as_bytes(q_cnt + e0), nullptr,   // (entry 10: the old second counter array — the record is one stream)
            as_bytes(q_mask + e0)};
        const int seg[13] = {lo.gs, lo.ax, lo.ay, lo.dir, lo.carry, lo.deliv, lo.act, lo.queue, lo.hw,
                             lo.dcnt, lo.dflag, lo.dflag, lo.dma_end};  // dword offsets, all multiples of 4 (entry 10 is empty)
        if constexpr (Cfg::kN != 0) {
            // One DMA instruction moves up to 64 pieces of ONE segment (LDS base + lane * 16), so the source
            // pick is scalar; the (compile-time) list of such instructions is dealt round-robin to the waves.
            // Per wave that is 3-4 instructions of ~3 VALU ops each — the phase is VALU-issue bound otherwise.
            // (kEarly: dealt to wavefronts 1.. only — wavefront 0 must not have a DMA of its own to wait for)
            const int wave_s = uniform(wave) - (kEarly ? 1 : 0), dma_w = nw - (kEarly ? 1 : 0);
            int job = 0;
#pragma unroll
            for (int k = 0; k < 12; ++k) {  // (fully unrolled: the shapes are compile-time constants — keep every `continue` a compile-time one)
                if ((k >= 2 && k <= 5) || k == 10) continue;  // (filled by unpack_records, not by DMA; no source)
                if (kDirect && ((k >= 1 && k <= 6) || k >= 9)) continue;  // agent records, actions, counters, flags: in registers
                const int pieces = (seg[k + 1] - seg[k]) >> 2;
                // (run-time queue length: the slot holds 2 N entries per env, the chunk in HBM is [E][Q] — contiguous, E * Q / 4 pieces)
                const int have = (Cfg::kQrt && k == 7) ? (E * Q) >> 2 : pieces;
                const bool wanted = k != 11 || op == OP_RESET;  // (scalar: the reset mask only matters to OP_RESET)
#pragma unroll
                for (int c = 0; c < pieces; c += 64, ++job)
                    if (job % dma_w == wave_s && c + lane < have && wanted)
                        lds_dma_b128(src[k] + (size_t)(c + lane) * 16, smem + seg[k] + 4 * c);
            }
            if constexpr (kMsg && !kDirect) {  // the agents' stored messages: a 13th array, outside the contiguous block
                const int pieces = (Cfg::kE * Cfg::kN) >> 2;
                for (int c = 0; c < pieces; c += 64, ++job)
                    if (job % dma_w == wave_s && c + lane < pieces)
                        lds_dma_b128(as_bytes(as_global(p.amsg) + (size_t)e0 * N) + (size_t)(c + lane) * 16, smem + lo.msg + 4 * c);
            }
            if constexpr (kEarly) {
                constexpr int KN = Cfg::kN, KG = 64 / KN;
                if (uniform(wave) * KG < Cfg::kE) {  // wave-uniform: a wavefront that runs agent phases
                    const int g = lane / KN, a_idx = lane - g * KN;
                    const bool mine = (g < KG) && (wave * KG + g < Cfg::kE);
                    unpack_own();
                    early = intent_of((op == OP_STEP) && mine && !r_flag, r_act, r_x, r_y, r_d);
                    if constexpr (!kCell) early.occ_w = occupant_of(early, r_carry, a_idx, (g < KG ? g : KG - 1) * KN);
                    keep_vgpr(early.tg0, early.occ_w);  // (materialised here, not sunk below the barrier)
                }
            }
        } else {  // N, Q are run-time values: thread t moves LDS piece t, its source picked per lane
            const int pieces = (lo.dma_end - lo.gs) >> 2;
            for (int b = wave * 64; b < pieces; b += nw * 64) {  // wave-uniform
                const int t = b + lane;
                const RW_GLOBAL char *g = src[0] + (size_t)t * 16;
#pragma unroll
                for (int k = 1; k < 12; ++k)
                    if (t >= ((seg[k] - seg[0]) >> 2)) g = src[k] + (size_t)(t - ((seg[k] - seg[0]) >> 2)) * 16;
                const bool no_src = (t >= ((seg[2] - seg[0]) >> 2) && t < ((seg[6] - seg[0]) >> 2)) ||   // ay .. deliv slots: unpack_records
                                    (t >= ((seg[11] - seg[0]) >> 2) && op != OP_RESET);                  // the reset mask: OP_RESET only
                if (t < pieces && !no_src) lds_dma_b128(g, smem + lo.gs + 4 * b);
            }
        }
        // (rounding pieces at the tail of hw / dcnt / dflag read a few bytes past the logical end of their source:
        //  the bitmap is allocated rounded up to 16 bytes, the records sit in the padded slab, the mask buffer has +64 bytes)
        RW_MARK(TL_DMA_ISSUED);
        RW_MARK(TL_ENV_LOADED);
        dma_wait();       // the stage-in DMA this wavefront issued has landed (explicit: see rware_cdna4.h) ...
        __syncthreads();  // ... and everybody else's: the one full barrier
        if constexpr (!kDirect) {  // (kDirect: the leader lane of each env publishes these from its registers, in AG)
            const uint8_t *s_dflag = reinterpret_cast<const uint8_t *>(smem + lo.dflag);
            for (int e = tid; e < ne; e += T) {
                int32_t *ev = s_envi + e * ENVI_W;
                const int cx = smem[lo.dcnt + 2 * e];
                const int rs = flag_of(cx, op == OP_RESET ? (int)s_dflag[e] : 0);
                ev[ENVI_STEPS] = cx & 0x7fffffff;
                ev[ENVI_INACTIVE] = smem[lo.dcnt + 2 * e + 1];
                ev[ENVI_RESET] = rs;
                ev[ENVI_SKIP] = rs;
                ev[ENVI_DONE] = 0; ev[ENVI_QDIRTY] = 0;
                if (rs) atomicOr(&s_misc[0], 1);
            }
            unpack_records();
            lds_barrier();
        }
    } else {
        dma_in(smem + lo.gs, (const RW_GLOBAL int32_t *)(g_shadow + (size_t)e0 * HW), (ne * HW * (int)sizeof(CellT) + 3) >> 2, tid, T);
        dma_in(s_ax, (const RW_GLOBAL int32_t *)(q_rec + (size_t)e0 * N), nea, tid, T);  // packed records -> the `ax` slot
        dma_in(s_queue, q_queue + (size_t)e0 * Q, ne * Q, tid, T);
        if (op == OP_STEP) dma_in(s_act, as_global(la.actions) + (size_t)e0 * N * AM, nea * AM, tid, T);
        if (kMsg) dma_in(s_msg, as_global(p.amsg) + (size_t)e0 * N, nea, tid, T);
        dma_in(smem + lo.hw, (const RW_GLOBAL int32_t *)q_hw, (HW + 31) / 32, tid, T);
        RW_MARK(TL_DMA_ISSUED);
        lds_barrier();  // orders the s_misc clear above before the flag writes below
        for (int e = tid; e < ne; e += T) {
            int32_t *ev = s_envi + e * ENVI_W;
            const Cnt c = cnt_load(e0 + e);
            const int rs = flag_of(c.x, op == OP_RESET ? (int)q_mask[e0 + e] : 0);
            ev[ENVI_STEPS] = c.x & 0x7fffffff;
            ev[ENVI_INACTIVE] = c.y;
            ev[ENVI_RESET] = rs;
            ev[ENVI_SKIP] = rs;  // an env that resets in this call does not step
            ev[ENVI_DONE] = 0; ev[ENVI_QDIRTY] = 0;
            if (rs) atomicOr(&s_misc[0], 1);
        }
        RW_MARK(TL_ENV_LOADED);
        dma_wait();
        __syncthreads();  // the one full barrier (the DMA has been waited for)
        unpack_records();
        lds_barrier();
    }
    keep_sgpr(k_reward_type, k_max_inactivity, k_max_steps, k_autoreset, k_n_goals, k_goal0, k_goal1, k_normalised, k_nt);
    if constexpr (Cfg::kQrt) keep_sgpr(Q);
    RW_MARK(TL_LOADED);
    }

    // kRollout == false is the single-step kernel (rw_step / rw_reset / rw_refresh_obs): no loop at all.
    const int n_steps = (kRollout && op == OP_STEP) ? la.n_steps : 1;
    // Rollout: when every agent lane owns exactly one (env, agent) for the whole launch, the NEXT step's
    // action is fetched into a register one step ahead, so its HBM latency hides under the current step.
    const bool act_prefetch = kRollout && !kMsg && (ne <= nw * (Cfg::kN ? 64 / (Cfg::kN ? Cfg::kN : 1) : p.groups_per_wave));
    int a_pref = ACT_NOOP;
