// The generic (DynamicCfg) step kernels, one translation unit per sensor range so that the library builds in
// parallel (rware_generic.hip compiled five times with -DRW_GENERIC_R=1..5).
#pragma once
#include "rware_kernels.h"

namespace rw_tab {

using step_kernel_t = void (*)(RW_LAUNCH_PARAMS_TYPES);

// the kernel for sensor range R (1..5): single step or fused rollout; uint16 shelf ids ("wide") when S > 255;
// FLATTENED / IMAGE observations, with or without communication bits
step_kernel_t generic_r1(bool rollout, bool wide, bool image, bool msg);
step_kernel_t generic_r2(bool rollout, bool wide, bool image, bool msg);
step_kernel_t generic_r3(bool rollout, bool wide, bool image, bool msg);
step_kernel_t generic_r4(bool rollout, bool wide, bool image, bool msg);
step_kernel_t generic_r5(bool rollout, bool wide, bool image, bool msg);
// were the generic / the ahead-of-time specialised kernels compiled with the event-counter code (RW_STATS_BUILD, rware_kernels.h)?
// The library's own build: generic yes, specialised no (the host-thread emulation build of the tests: both).
bool generic_has_stats();
bool static_has_stats();

}  // namespace rw_tab
