// rware_jit.h — run-time specialisation of the step kernel (hipRTC).
//
// rw_create asks for an exact-shape build (rw::StaticCfg with every shape folded in) of a task that has no ahead-of-time
// entry in rware_static_table.h — a `layout=` string, column_height != 8, sensor_range 2..5, more than 19 agents, ... (the
// reference's constructor arguments, rware/warehouse.py:146-170, and the ids of rware/__init__.py:83-175): the device headers
// travel inside the library (rware_jit_sources.inc), hipRTC compiles `rware_step_kernel<R, CellT, StaticCfg<...>>` for
// gfx950, and the code object is cached on disk keyed by the sources, the options and the shape.  hipRTC is loaded with
// dlopen: a box without it simply keeps the ahead-of-time builds (generic kernel for such shapes).
#pragma once
#include <stddef.h>
#include <stdint.h>

#include <string>
#include <vector>

namespace rw_jit {

struct Shape {
    int R, H, W, N, Q, S, E, T, M;   // StaticCfg's shape arguments (Q: the exact queue length)
    int wide;                        // 1: uint16 shelf ids
    int obs;                         // rw::OBS_FLATTENED / OBS_IMAGE / OBS_FLATTENED_MSG (kObs)
    int NL;                          // IMAGE: layers baked in (0 = FLATTENED)
    uint32_t layers;                 // 4 bits per layer id, first layer lowest
    int directional;                 // IMAGE: image_observation_directional (else -1)
    int nt;                          // per-step kernel: 1 = non-temporal observation stores
    int stats;                       // 1: compiled with RW_STATS_BUILD — the event counters of RW_STATS_ON (rware_kernels.h)
};

struct Result {
    std::vector<char> code;          // the code object (hsaco)
    std::string step_name, rollout_name;   // lowered (mangled) kernel names inside it
    bool from_cache = false;
    double compile_seconds = 0.0;
    std::string log;                 // what happened (reason of a failure, hipRTC's log, cache path)
};

// where the code object of a shape is (or would be) cached: "" without a cache directory (no HOME / RWARE_JIT_CACHE) — for diagnostics and
// for the tests that corrupt the file on purpose (oracle/sanitize.sh)
std::string cache_file(const Shape &s, const char *arch);

// false: no build (hipRTC missing, compile error, ...) — `out->log` says why; the caller keeps its ahead-of-time kernel
bool compile(const Shape &s, const char *arch, Result *out);

}  // namespace rw_jit
