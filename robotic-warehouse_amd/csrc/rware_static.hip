// rware_static.hip — instantiates ONE group of the specialised (rw::StaticCfg) step kernels, RW_STATIC_GROUP, and hands
// its table out.  Compiled once per group (see the Makefile and rware_static_table.h).
#include <hip/hip_runtime.h>

#ifndef RW_STATIC_GROUP
#define RW_STATIC_GROUP 0  // a plain `hipcc -c rware_static.hip` builds the BASELINE group
#endif
#include "rware_static_table.h"

#define RW_CAT2(a, b) a##b
#define RW_CAT(a, b) RW_CAT2(a, b)
namespace rw_tab {
const StaticEntry *RW_CAT(static_group_, RW_STATIC_GROUP)(int *n) {
    *n = (int)(sizeof(kEntries) / sizeof(kEntries[0]));
    return kEntries;
}
#if RW_STATIC_GROUP == 0
bool static_has_stats() { return RW_STATS_BUILD != 0; }  // (the same for every group: one set of compiler flags builds them all)
#endif
}  // namespace rw_tab
#undef RW_CAT
#undef RW_CAT2
