// rware_hooks.h — the library's test / A-B hooks.
//
// A handful of decisions rw_create makes by a measured rule (observation-store mode, start stagger, which kernel build, run-time
// specialisation, launcher threads) can be moved from the environment — for the same-box A/B runs under profiles/tools and for tests
// that have to reach a path the rule would not take.  None of it is API: callers use rw_config.stream_flags.  So the variables are
// honoured ONLY when RWARE_HOOKS=1 is set as well — a production process that happens to inherit one of them runs the rules.
//   RWARE_OBS_STORES=cached|stream   RWARE_STAGGER_TICKS=n   RWARE_PREFER_QRT=1   RWARE_JIT=off|force   RWARE_JIT_LIBRARY=path
//   RWARE_JIT_NO_CACHE=1   RWARE_PIPE=0|1   RWARE_PIPE_E=n   RWARE_PIPE_WGS_PER_CU=n   RWARE_PIPE_GRID=n   RWARE_MULTI_THREADS=0|1
//   RWARE_SELFTEST_BREAK=1   RWARE_PRIO=0|1   RWARE_PRIO_ROLLOUT=0|1   RWARE_WIDE_E4=0|1
// (RWARE_JIT_CACHE — where compiled code objects are kept — is configuration, not a hook, and is read unconditionally.)
// A hook variable that is set while RWARE_HOOKS is not is IGNORED — and says so once per variable on stderr, so that a script written
// against an older library (which read e.g. RWARE_JIT=off unconditionally) does not change behaviour silently.
#pragma once
#include <atomic>
#include <cstdio>
#include <cstdlib>
#include <cstring>

inline const char *rw_hook(const char *name) {
    const char *on = getenv("RWARE_HOOKS");
    if (on && on[0] == '1' && on[1] == '\0') return getenv(name);
    const char *v = getenv(name);
    if (v && *v) {
        // one note per variable and process (a small fixed table: the hooks are the dozen names above)
        static std::atomic<const char *> noted[16];
        for (auto &slot : noted) {
            const char *seen = slot.load(std::memory_order_acquire);
            if (seen && strcmp(seen, name) == 0) break;
            const char *expect = nullptr;
            if (!seen && slot.compare_exchange_strong(expect, name, std::memory_order_acq_rel)) {
                fprintf(stderr, "librware_hip: %s=%s is ignored: the test / A-B hooks are honoured only with RWARE_HOOKS=1 (csrc/rware_hooks.h)\n", name, v);
                break;
            }
            if (expect && strcmp(expect, name) == 0) break;
        }
    }
    return nullptr;
}
