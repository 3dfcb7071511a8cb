// rware_phase_write_back.h — part of rw::rware_step_kernel (rware_kernels.h), included INSIDE the kernel body: WB — state write-back in three roles (counter record + queue; agent records + rewards; shelf-shadow patches)
// A textual unit, not a function: the phases share ~60 locals (LDS pointers, shapes, the agent lanes' registers), and every
// way of passing them that was tried — lambdas, always_inline or not — reschedules the kernels around it (round 5: +-10
// instructions per kernel, two 13/14-agent builds over a register cliff).  Splitting the text keeps every build's ISA.
    // ---------------------------------------------------------------- WB: state write-back, one role per wavefront
    // (envs flagged for reset were written by RS).  Where it runs is a measured choice.  Without the split below
    // (fewer than 4 wavefronts, or a large observation chunk), one role per wavefront:
    //   single step    before the observation: its small stores then drain underneath P7; issued after the
    //                  18 MB observation stream they queue behind it and hold every wavefront ~0.8 us longer
    //   fused rollout  after the observation stores: the next step's compute hides them, and the stream
    //                  starts 0.4 us earlier (5.62 -> 5.44 us per step)
    // With 4 wavefronts the workgroup splits after the agent phases ("split"): wavefront 3 is the service wave — it
    // writes the self bits while wavefronts 0..2 gather the window rows, and after the barrier it does ALL the state
    // write-back while wavefronts 0..2 expand and store the observation.  The observation stream — what the step
    // ends with — then starts one write-back earlier, and the small state stores go out beside its head instead of
    // behind its tail.
    auto write_back = [&](int first_role, int role_step) {
    for (int role = first_role; role < 3; role += role_step) {  // wave-uniform
        if (role == 0) {  // per-env counters and flags, request queue
            if (op == OP_STEP)
                for (int e = lane; e < ne; e += 64) {
                    const int32_t *ev = s_envi + e * ENVI_W;
                    if (ev[ENVI_RESET]) continue;
                    const int ge = e0 + e;
                    // (fused rollout: only the launch's last step stores the pending-reset bit — the reset at the top of the
                    //  following step consumes it from LDS)
                    const int pend = (ev[ENVI_DONE] && k_autoreset == AR_NEXT_STEP && (!kRollout || t + 1 == n_steps)) ? (int)0x80000000 : 0;
                    cnt_store(ge, ev[ENVI_STEPS] | pend, ev[ENVI_INACTIVE]);  // ONE 8-byte store: the env's counter record
                    term_t[ge] = (uint8_t)ev[ENVI_DONE];
                    // Only what changed: RW_BUF_TRUNCATED is zero for the engine's lifetime (the reference never truncates,
                    // :942); the queue changes only on a delivery.  Every store stream a step does not issue is ~0.1 us of it
                    // (DESIGN.md ablations).
                    if (ev[ENVI_QDIRTY])
                        for (int k = 0; k < Q; ++k) q_queue[(size_t)ge * Q + k] = s_queue[e * Q + k];
                }
        } else if (role == 1) {  // agent records and rewards: the chunk is contiguous in both [B][N] arrays
            if (RW_RARE(stats_on)) count_events(false, lane, 64);
            if (op == OP_STEP)
                for (int i = lane; i < nea; i += 64) {
                    if (s_envi[rw_div18(i, mN) * ENVI_W + ENVI_RESET]) continue;
                    const size_t gi = (size_t)e0 * N + i;
                    q_rec[gi] = rec_pack(s_ay[i] * W + s_ax[i], s_dir[i], s_deliv[i], s_carry[i]);  // one store stream, not five
                    rew_t[gi] = s_rew[i];
                    if (kMsg) as_global(p.amsg)[gi] = s_msg[i];
                }
        } else if (role == 2) {  // patch the shelf shadow at the two cells a LOADED mover changed
            // (The exported int32 grid, RW_BUF_GRID, is NOT patched here any more: it is rebuilt from the shadow and the agent
            //  coordinates when somebody asks for it — rw_refresh_grid.  Its scattered 4-byte patches were partial-line
            //  writes; once a batch outgrows the Infinity Cache each of them is a read-modify-write in HBM: 15 % of the
            //  step at B = 262144, measured by ablation.)
            if (op == OP_STEP)
                for (int i = lane; i < nea; i += 64) {
                    const int mv = s_mv[i], carry = s_carry[i];
                    if (mv < 0 || !carry) continue;
                    const int e = rw_div18(i, mN);
                    if (s_envi[e * ENVI_W + ENVI_RESET]) continue;
                    const int st = mv & 0xffff, tg = mv >> 16;
                    const size_t ge = (size_t)(e0 + e);
                    // the cell it left: cleared unless a loaded follower stepped onto it — the follower then writes that
                    // cell itself (as its `tg`), so every shadow cell has exactly one writer
                    if (s_gs[e * HW + st] == 0) g_shadow[ge * HW + st] = 0;
                    g_shadow[ge * HW + tg] = (CellT)carry;
                }
        }
    }
    };
