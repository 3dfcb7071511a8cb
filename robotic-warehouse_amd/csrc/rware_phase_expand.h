// rware_phase_expand.h — part of rw::rware_step_kernel (rware_kernels.h), included INSIDE the kernel body: ST — the bit string expanded to float32 and stored (single pass with byte-sized coordinates; two-pass for fractions; IMAGE)
// A textual unit, not a function: the phases share ~60 locals (LDS pointers, shapes, the agent lanes' registers), and every
// way of passing them that was tried — lambdas, always_inline or not — reschedules the kernels around it (round 5: +-10
// instructions per kernel, two 13/14-agent builds over a register cliff).  Splitting the text keeps every build's ISA.
    // ---------------------------------------------------------------- ST: obs, float4 #q == nibble #q
    // The chunk's first float sits on a 16-byte boundary whenever the chunk size is a multiple of 4 envs — every specialised build (a
    // compile-time fact there) and the default geometries of the generic kernel.  An odd run-time geometry (envs_per_workgroup 3, 5, ...) may
    // start a chunk on a 4- or 8-byte boundary: the hardware's dwordx4 stores do not mind, but a float4 store through a misaligned pointer
    // is undefined in C++ (UBSan on the host-thread build, round 6) — such a chunk takes the scalar tail loops below for all of its floats.
    constexpr bool kChunkAligned = Cfg::kE != 0 && Cfg::kE % 4 == 0;
    if constexpr (!kImage) {
        const int nf = nea * L;
        float *out = obs_t + (size_t)e0 * N * L;  // 16-byte aligned when e0 is a multiple of 4 (kChunkAligned), else checked
        const int nf4 = (kChunkAligned || (reinterpret_cast<uintptr_t>(out) & 15u) == 0) ? nf >> 2 : 0;
        float4 *out4 = reinterpret_cast<float4 *>(out);
        auto spread = [&](uint32_t nib) -> float4 {  // 4 bits -> 4 floats
            // one multiply spreads the bits into 4 bytes (0 or 1 each); hidden from the optimiser so that each
            // byte converts with ONE v_cvt_f32_ubyteN instead of a shift/and/convert chain
            const uint32_t b = opaque((nib * 0x00204081u) & 0x01010101u);
            float4 v;
            v.x = (float)(b & 0xFFu);
            v.y = (float)((b >> 8) & 0xFFu);
            v.z = (float)((b >> 16) & 0xFFu);
            v.w = (float)(b >> 24);
            return v;
        };
        // 16-byte store at (uniform) out + a per-lane byte offset the optimiser cannot take apart: it then keeps
        // the scalar-base form of the store instead of rebuilding a 64-bit per-lane address for every pass.
        // (Not inline asm: hipcc must see the store to respect the write-data hazard of 128-bit stores.)
        // `nt` (a std::bool_constant tag): store with the non-temporal hint.  Whole 128-byte lines written exactly once are a
        // pure stream; with the hint they no longer displace the state the next launch reads back, and at the headline batch
        // the step goes 7.13 -> 6.17 us, past the Infinity Cache 70.2 -> 60.2 us (round 3; round 1 measured the opposite on
        // the old two-pass expansion, whose second pass re-touched lines).  Per engine, by Params::nt_obs (rw_create's rule).
        auto store4 = [&](auto nt, uint32_t byte_off, float4 v) {
            float4 *dst = reinterpret_cast<float4 *>(reinterpret_cast<char *>(out) + (size_t)opaque(byte_off));
            if constexpr (decltype(nt)::value) store_f4_nt(dst, v); else *dst = v;
        };
        auto expand = [&](int q4) -> float4 {  // float4 #q4 of the chunk == nibble #q4 of the bit string
            return spread((s_obits[q4 >> 3] >> ((q4 & 7) << 2)) & 0xFu);
        };
        // ONE pass over the chunk's float4s when the coordinates are plain cell indices (not normalised): thread t takes
        // float4 t, t + T, ...: its nibble position inside the word (t & 7) and its word column (t >> 3) never change
        // (T % 8 == 0), and m = (4 q + 3) mod L — the float4 holds a coordinate slot iff m < 5 — and the agent index
        // (4 q + 3) div L advance by constants.  The four bits become four BYTES (0 / 1) that convert with one
        // v_cvt_f32_ubyteN each; a coordinate is a small integer and converts the same way, so the agent's (x | y << 8)
        // word is simply OR-ed into the byte lanes of its slots (which are 0 in the bit string).  Every float4 is written
        // exactly once and in order: whole 128-byte lines, no second scattered pass (7.84 -> 7.5 us per step at the
        // headline batch, -7 % at the cache-exceeding batches).
        // (not in the fused rollout: its steps are bound by instruction issue, not by the store stream, and the single pass
        //  costs ~10 more VALU operations per float4: 4.16 -> 4.78 us per step there)
        // (the coordinates travel as bytes: layouts wider or taller than 256 cells take the two-pass form as well — a
        //  compile-time fact in the exact-shape and size-static builds)
        const bool xy_bytes = !kRollout && !k_normalised && W <= 256 && H <= 256;  // workgroup-uniform
        // (round 5 A/B: two passes for sensor_range >= 2 — half the expansion's VALU work, 4 % of the float4s written by the second pass —
        //  is slower: config 5's shard 36.2 -> 36.8 us, past the Infinity Cache 76 -> 96: the partial lines cost more than the instructions)
        auto single_pass = [&](auto nt) {
            // (the thread index through an opaque copy: otherwise the address arithmetic of BOTH copies of the pass is hoisted in
            //  front of the branch that picks one — large-16ag r=2: 102 VGPRs instead of 60, 4 workgroups per CU instead of 7)
            // (only in the builds that hold both copies: with one copy the hoisting is wanted — the address arithmetic then runs
            //  while the workgroup waits at the bit-string barrier)
            int tq = x_tid;
            if constexpr (Cfg::kNT < 0) asm volatile("" : "+v"(tq));
            const int shift = (tq & 7) << 2, words_per_pass = x_TW >> 3;  // (x_TW % 8 == 0)
            const uint32_t *wp = s_obits + (tq >> 3);
            const int dm = (4 * x_TW) % L, di = (4 * x_TW) / L;
            int m = (4 * tq + 3) % L, ai = (4 * tq + 3) / L;
            const int passes = (nf4 + x_TW - 1) / x_TW;  // a compile-time constant in the specialised builds (full unroll)
            // in groups of 8 passes: first the LDS reads of all 8 in one unconditional batch (a read past the string still
            // lands inside the workgroup's LDS; the agent index is clamped), then the 8 expansions
            for (int k0 = 0; k0 < passes; k0 += 8) {
                uint32_t wv[8], xyv[8];
                int mv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    mv[j] = m;
                    wv[j] = (k0 + j < passes) ? wp[(k0 + j) * words_per_pass] : 0u;
                    xyv[j] = (k0 + j < passes) ? (uint32_t)s_xy[min(ai, nea - 1)] : 0u;
                    m += dm;
                    const bool wrap = m >= L;
                    m = wrap ? m - L : m;
                    ai += di + (wrap ? 1 : 0);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (k0 + j >= passes) break;
                    const int q4 = tq + (k0 + j) * x_TW;
                    if (q4 < nf4) {
                        const uint32_t bits = (((wv[j] >> shift) & 0xFu) * 0x00204081u) & 0x01010101u;
                        // x goes to byte 3 - m, y to byte 4 - m of this float4 (m == 4: x was the last float of the one before)
                        const uint32_t xy = (mv[j] <= 3) ? (xyv[j] << ((24 - 8 * mv[j]) & 31)) : ((mv[j] == 4) ? (xyv[j] >> 8) : 0u);
                        const uint32_t b = opaque(bits | xy);
                        float4 v;
                        v.x = (float)(b & 0xFFu);
                        v.y = (float)((b >> 8) & 0xFFu);
                        v.z = (float)((b >> 16) & 0xFFu);
                        v.w = (float)(b >> 24);
                        store4(nt, (uint32_t)q4 << 4, v);
                    }
                }
            }
        };
        if (x_worker && xy_bytes) {
            if constexpr (Cfg::kNT == 1) single_pass(yes_t{});
            else if constexpr (Cfg::kNT == 0) single_pass(no_t{});
            else {  // (two copies of the pass, one taken: a scalar branch on a workgroup-uniform flag)
                if (k_nt) single_pass(yes_t{}); else single_pass(no_t{});
            }
        }
        // normalised coordinates are fractions: bulk pass over every float4 that holds no coordinate slot (all but ~2 in
        // 18), then a second pass for the coordinate slots
        if (x_worker && !xy_bytes) {
            const int shift = (x_tid & 7) << 2, words_per_pass = x_TW >> 3;  // (x_TW % 8 == 0)
            const uint32_t *wp = s_obits + (x_tid >> 3);
            const int dm = (4 * x_TW) % L;
            int m = (4 * x_tid + 3) % L;
            const int passes = (nf4 + x_TW - 1) / x_TW;
            for (int k0 = 0; k0 < passes; k0 += 8) {
                uint32_t wv[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) wv[j] = (k0 + j < passes) ? wp[(k0 + j) * words_per_pass] : 0u;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    if (k0 + j >= passes) break;
                    const int q4 = x_tid + (k0 + j) * x_TW;
                    if (q4 < nf4 && m >= 5) store4(no_t{}, (uint32_t)q4 << 4, spread((wv[j] >> shift) & 0xFu));
                    m += dm;
                    m = (m >= L) ? m - L : m;
                }
            }
        }
        // coordinate pass: per agent, the one or two float4s that hold its x (element i*L) and y (i*L + 1)
        if (x_worker && !xy_bytes)
        for (int i = x_tid; i < nea; i += x_TW) {
            const int g = i * L, q4 = g >> 2, pos = g & 3;
            const float fx = s_fx[i], fy = s_fy[i];
            if (q4 < nf4) {
                float4 v = expand(q4);
                if (pos == 0) { v.x = fx; v.y = fy; }
                else if (pos == 1) { v.y = fx; v.z = fy; }
                else if (pos == 2) { v.z = fx; v.w = fy; }
                else { v.w = fx; }
                out4[q4] = v;
            }
            if (pos == 3 && q4 + 1 < nf4) {
                float4 v = expand(q4 + 1);
                v.x = fy;
                out4[q4 + 1] = v;
            }
        }
        if (x_worker)
        for (int g = (nf4 << 2) + x_tid; g < nf; g += x_TW) {  // < 4 leftover floats (partial last workgroup)
            const int i = g / L, k = g - i * L;
            out[g] = (k >= 2) ? (((s_obits[g >> 5] >> (g & 31)) & 1u) ? 1.0f : 0.0f)
                              : (k == 0 ? s_fx[i] : s_fy[i]);
        }
    }
    else {  // IMAGE: every element is a bit of the string; no coordinate slots
        const int Limg = k_n_layers * CELLS;
        const int nf = nea * Limg;
        float *out = obs_t + (size_t)e0 * N * Limg;  // 16-byte aligned when e0 is a multiple of 4 (kChunkAligned), else checked
        const int nf4 = (kChunkAligned || (reinterpret_cast<uintptr_t>(out) & 15u) == 0) ? nf >> 2 : 0;
        if (worker) {
            const int shift = (tid & 7) << 2, words_per_pass = TW >> 3;  // (see the FLATTENED bulk pass)
            const uint32_t *wp = s_obits + (tid >> 3);
            const int passes = (nf4 + TW - 1) / TW;
            for (int k = 0; k < passes; ++k) {
                const int q4 = tid + k * TW;
                if (q4 >= nf4) break;
                const uint32_t b = opaque((((wp[k * words_per_pass] >> shift) & 0xFu) * 0x00204081u) & 0x01010101u);  // 4 bits -> 4 bytes
                float4 v;
                v.x = (float)(b & 0xFFu);
                v.y = (float)((b >> 8) & 0xFFu);
                v.z = (float)((b >> 16) & 0xFFu);
                v.w = (float)(b >> 24);
                float4 *dst = reinterpret_cast<float4 *>(reinterpret_cast<char *>(out) + (size_t)opaque((uint32_t)q4 << 4));
                // (AGENT_DIRECTION patches its cells afterwards: cached)
                const bool nt = Cfg::kNT == 1 ? true : Cfg::kNT == 0 ? false : (k_nt != 0);
                if (nt && !(k_transposed & 1)) store_f4_nt(dst, v); else *dst = v;
            }
        }
        if (worker)
        for (int g = (nf4 << 2) + tid; g < nf; g += TW) out[g] = ((s_obits[g >> 5] >> (g & 31)) & 1u) ? 1.0f : 0.0f;
        if (k_transposed & 1) {
            // AGENT_DIRECTION (:547-552): the marked cells hold dir + 1, not 1.  Patched after every 0/1 store of
            // the workgroup has completed (full barrier: vmcnt), one thread per (agent, image row) as in P7.
            dma_wait();
            __syncthreads();
            if (worker)
            for (int w = tid; w < nea * WIN; w += TW) {
                const int i = w / WIN, r = w - i * WIN;
                const int e = rw_div18(i, mN);
                const int ax = s_ax[i], ay = s_ay[i], d = k_directional ? s_dir[i] : DIR_UP;
                for (int cc = 0; cc < WIN; ++cc) {
                    int wr = r, wc = cc;
                    if (d == DIR_DOWN) { wr = WIN - 1 - r; wc = WIN - 1 - cc; }
                    else if (d == DIR_LEFT) { wr = WIN - 1 - cc; wc = r; }
                    else if (d == DIR_RIGHT) { wr = cc; wc = WIN - 1 - r; }
                    const int y = ay - R + wr, x = ax - R + wc;
                    if ((unsigned)x >= (unsigned)W || (unsigned)y >= (unsigned)H || x >= H || y >= W) continue;
                    const int ida = s_ga[e * HW + x * W + y] & 0x7f;
                    if (!ida) continue;
                    const float v = (float)(s_dir[e * N + ida - 1] + 1);
#pragma unroll
                    for (int l = 0; l < 8; ++l)
                        if (l < k_n_layers && k_layer[l] == LAYER_AGENT_DIRECTION) out[(size_t)i * Limg + (l * WIN + r) * WIN + cc] = v;
                }
            }
        }
    }
