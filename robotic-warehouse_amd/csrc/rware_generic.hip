// rware_generic.hip — instantiates the generic (DynamicCfg) step kernels of ONE sensor range, RW_GENERIC_R.
// Compiled five times (see the Makefile); the exact-shape builds live in rware_capi.hip.
#include <hip/hip_runtime.h>

// The generic kernels carry the event-counter code (RW_STATS_ON; dead and compiled out of the exact-shape builds — rware_kernels.h):
// an engine that is asked for counters runs on these, or on a run-time compiled exact-shape build made with the same switch.
#ifndef RW_STATS_BUILD
#define RW_STATS_BUILD 1
#endif
#include "rware_kernel_table.h"

#ifndef RW_GENERIC_R
#define RW_GENERIC_R 1  // a plain `hipcc -c rware_generic.hip` builds the sensor_range 1 object
#endif
#define RW_CAT2(a, b) a##b
#define RW_CAT(a, b) RW_CAT2(a, b)

namespace rw_tab {
namespace {

template <bool kRollout, typename CellT>
step_kernel_t pick(bool image, bool msg) {
    constexpr int R = RW_GENERIC_R;
    if (msg && image) return (step_kernel_t)rw::rware_step_kernel<R, CellT, rw::DynamicCfg, kRollout, rw::OBS_IMAGE_MSG>;
    if (msg) return (step_kernel_t)rw::rware_step_kernel<R, CellT, rw::DynamicCfg, kRollout, rw::OBS_FLATTENED_MSG>;
    if (image) return (step_kernel_t)rw::rware_step_kernel<R, CellT, rw::DynamicCfg, kRollout, rw::OBS_IMAGE>;
    return (step_kernel_t)rw::rware_step_kernel<R, CellT, rw::DynamicCfg, kRollout>;
}

}  // namespace

step_kernel_t RW_CAT(generic_r, RW_GENERIC_R)(bool rollout, bool wide, bool image, bool msg) {
    if (rollout) return wide ? pick<true, uint16_t>(image, msg) : pick<true, uint8_t>(image, msg);
    return wide ? pick<false, uint16_t>(image, msg) : pick<false, uint8_t>(image, msg);
}

#if RW_GENERIC_R == 1
bool generic_has_stats() { return RW_STATS_BUILD != 0; }
#endif

}  // namespace rw_tab
#undef RW_CAT
#undef RW_CAT2
