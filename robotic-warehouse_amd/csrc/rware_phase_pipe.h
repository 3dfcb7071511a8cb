// rware_phase_pipe.h — part of rw::rware_step_kernel (rware_kernels.h), included INSIDE the kernel body: PIPE — the pipelined flow's chunk bookkeeping, its stage-in (LDS-DMA by the service wavefront), and its prologue
// A textual unit, not a function: the phases share ~60 locals (LDS pointers, shapes, the agent lanes' registers), and every
// way of passing them that was tried — lambdas, always_inline or not — reschedules the kernels around it (round 5: +-10
// instructions per kernel, two 13/14-agent builds over a register cliff).  Splitting the text keeps every build's ISA.
    // ---------------------------------------------------------------- PIPE: the workgroup's chunks, the two buffers, the stage-in
    // chunk `it` of this workgroup is chunk blockIdx + it * gridDim of the batch and lives in LDS buffer it & 1
    const int pipe_wave = kPipe ? uniform(tid >> 6) : 0;  // (scalar: the roles below are scalar branches)
    const int pipe_chunks = kPipe ? (B / E - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x : 0;
    auto pipe_e0 = [&](int it_) RW_INLINE -> int { return ((int)blockIdx.x + it_ * (int)gridDim.x) * E; };
    // stage the chunk whose first env is ce0 into the buffer at `base`: shelf layer, packed records, actions, queue, highway bitmap,
    // counter records — ONE wavefront issues the whole list (LDS-DMA, 1 KiB per instruction), nothing is waited for here
    auto pipe_stage_in = [&](int ce0, int32_t *base) RW_INLINE {
        if constexpr (kPipe) {
            static_assert(!kPipe || ((Cfg::kE * Cfg::kN) % 4 == 0 && (Cfg::kE * Cfg::kQcap) % 4 == 0 && Cfg::kE % 4 == 0 &&
                                     (Cfg::kE * Cfg::kH * Cfg::kW * (int)sizeof(CellT)) % 16 == 0), "chunk not 16-byte granular");
            const RW_GLOBAL char *src[6] = {as_bytes(g_shadow + (size_t)ce0 * HW), as_bytes(q_rec + (size_t)ce0 * N),
                                            as_bytes(as_global(la.actions) + (size_t)ce0 * N * AM), as_bytes(q_queue + (size_t)ce0 * Q),
                                            as_bytes(q_hw), as_bytes(q_cnt + ce0)};
            const int seg[6] = {lo.gs, lo.ax, lo.act, lo.queue, lo.hw, lo.dcnt}, end[6] = {lo.ax, lo.ay, lo.queue, lo.hw, lo.dcnt, lo.dflag};
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const int pieces = (end[k] - seg[k]) >> 2;
                const int have = (Cfg::kQrt && k == 3) ? (E * Q) >> 2 : pieces;  // (run-time queue length: [E][Q] in HBM, contiguous)
#pragma unroll
                for (int c = 0; c < pieces; c += 64)
                    if (c + lane < have) lds_dma_b128(src[k] + (size_t)(c + lane) * 16, base + seg[k] + 4 * c);
            }
        }
    };
    // zero [lo_, hi_) dwords of a buffer, one wavefront, 16 bytes per lane
    auto pipe_zero = [&](int32_t *base, int lo_, int hi_) RW_INLINE {
        for (int i = lane; i < ((hi_ - lo_) >> 2); i += 64) reinterpret_cast<int4 *>(base + lo_)[i] = int4{0, 0, 0, 0};
    };
    // the agent lanes' own record, action and counter record out of the staged chunk (what the classic flow fetches from HBM
    // at the top of the kernel): the rest of the kDirect path is shared
    auto load_own_lds = [&]() RW_INLINE {
        if constexpr (kPipe) {
            constexpr int KN = Cfg::kN, KG = 64 / KN;
            const int g = lane / KN, a_idx = lane - g * KN;
            if (g < KG && g < Cfg::kE) {
                const int i = g * KN + a_idx;
                r_rec = (uint32_t)s_ax[i];
                if (op == OP_STEP) r_act = s_act[i * AM];
                r_cx = sm[lo.dcnt + 2 * g];
                r_inact = sm[lo.dcnt + 2 * g + 1];
            }
        }
    };
    if constexpr (kPipe) {
        int4 *z = reinterpret_cast<int4 *>(smem);
        for (int i = tid; i < (2 * lo.total) >> 2; i += T) z[i] = int4{0, 0, 0, 0};
        lds_barrier();  // (the clear is another wavefront's: in front of the DMA into the same buffers)
        if (pipe_wave == 3) {
            pipe_stage_in(pipe_e0(0), smem);
            if (pipe_chunks > 1) pipe_stage_in(pipe_e0(1), smem + lo.total);
            dma_wait();
        }
        lds_barrier();
        RW_MARK(TL_LOADED);
    }
