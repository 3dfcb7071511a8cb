#!/bin/bash
# TEST INFRASTRUCTURE (sanitizers).  CPU only (GPU AddressSanitizer is not available on the pool).
#   bash oracle/sanitize.sh > profiles/rNN_sanitizers.txt 2>&1
#  1. the oracle under AddressSanitizer + UBSan over every golden trace, the numpy RNG pins and the live reference check;
#  2. the product sources built for host threads (tests/emu: every workgroup = blockDim.x OS threads, LDS atomics = __atomic builtins)
#     under ThreadSanitizer: an exact-shape build on a golden trace, the generic kernel, a per-cell (kCell) agent-phase build with 10 and
#     16 agents in crowded warehouses, a chunk-pipelined (PIPE=1) build walking several chunks per workgroup, and rw_multi's launcher
#     threads: 8 engines, >= 1000 rounds, create / destroy cycles (the sleep / wake handshake);
#  2b. the WHOLE emulation library under AddressSanitizer + UBSan (heap redzones around every device buffer, the LDS behind a launch's dynamic
#     shared memory poisoned): the emulated engine tests, the event counters and the reference's KATs;
#  3. rware_jit.cpp under AddressSanitizer + UBSan: no hipRTC library, a library without the entry points, corrupt / truncated / foreign
#     cache files, a cache directory that is not private.
set -u
cd "$(dirname "$0")/.."
OUT=/tmp/rware_san; mkdir -p $OUT
CSRC=robotic-warehouse_amd/csrc
echo "== 1. oracle/rware_oracle.c under -fsanitize=address,undefined: golden traces + RNG vs numpy + live reference check"
gcc -O1 -g -std=c11 -shared -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -o $OUT/librware_oracle_asan.so oracle/rware_oracle.c || exit 1
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
  RWARE_ORACLE_SO=$OUT/librware_oracle_asan.so timeout 2400 python -m pytest tests/test_oracle_golden.py tests/test_oracle_rng_numpy.py tests/test_oracle_vs_reference.py tests/test_event_counters.py -k "not emulated" -m "not gpu" -q -x -p no:cacheprovider 2>&1 | tail -5

echo "== 2. tests/emu (the engine's kernels + C-ABI host code on host threads) under -fsanitize=thread"
FLAGS="-DRW_NO_JIT -DRW_WITH_PIPE=1 -DRW_STATS_BUILD=1 -O1 -g -std=c++17 -fPIC -pthread -Itests/emu -Wno-unknown-pragmas -fsanitize=thread"
BUILT_GROUPS="0 10 16 17 18"   # BASELINE shapes | small 9..14 agents (kCell) | large 9..14, 15..19 agents (kCell) | the pipelined builds
for r in 1 2; do g++ $FLAGS -DRW_GENERIC_R=$r -c -x c++ $CSRC/rware_generic.hip -o $OUT/g$r.o 2>/dev/null & done
g++ $FLAGS -c -x c++ $CSRC/rware_capi.hip -o $OUT/capi.o 2>/dev/null &
g++ $FLAGS -c tests/emu/emu_globals.cpp -o $OUT/glob.o &
g++ $FLAGS -c -x c++ $CSRC/rware_selftest.hip -o $OUT/selftest.o 2>/dev/null &
wait
for g in $BUILT_GROUPS; do g++ $FLAGS -DRW_STATIC_GROUP=$g -c -x c++ $CSRC/rware_static.hip -o $OUT/s$g.o 2>/dev/null & done
wait
# (what the cases below do not need is stubbed: sensor ranges 3..5 of the generic kernel, the other groups of the static table)
{
  echo '#include <hip/hip_runtime.h>'
  echo '#include "rware_static_table.h"'
  echo 'namespace rw_tab { step_kernel_t generic_r3(bool, bool, bool, bool) { return nullptr; } step_kernel_t generic_r4(bool, bool, bool, bool) { return nullptr; } step_kernel_t generic_r5(bool, bool, bool, bool) { return nullptr; }'
  for g in $(seq 0 18); do case " $BUILT_GROUPS " in *" $g "*) ;; *) echo "const StaticEntry *static_group_$g(int *n) { *n = 0; return nullptr; }";; esac; done
  echo '}'
} > $OUT/stub.cpp
g++ $FLAGS -I$CSRC -c $OUT/stub.cpp -o $OUT/stub.o 2> $OUT/stub.err || { cat $OUT/stub.err; exit 1; }
OBJS="$OUT/capi.o $OUT/selftest.o $OUT/g1.o $OUT/g2.o $OUT/glob.o $OUT/stub.o"; for g in $BUILT_GROUPS; do OBJS="$OBJS $OUT/s$g.o"; done
g++ -shared -pthread -fsanitize=thread -o $OUT/librware_emu_tsan.so $OBJS || exit 1
LD_PRELOAD=$(gcc -print-file-name=libtsan.so) TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:history_size=2" RWARE_HOOKS=1 timeout 5400 python - <<'EOP' 2>&1 | grep -E "WARNING: ThreadSanitizer|SUMMARY|tsan-run|Error|Traceback|assert" | sort | uniq -c | head -40
import os, sys
sys.path[:0] = [".", "oracle", "tests"]
import numpy as np
import golden_util as gu
from engine_backend import EngineBackend
import rware_amd
from rware_amd import _capi
from rware_oracle import OracleVecEnv
LIB = "/tmp/rware_san/librware_emu_tsan.so"

def against_oracle(label, B, steps, ctor, kw, seed=3, p=(.1, .55, .1, .1, .15), want=None):
    # (stats=True: the event counters — global atomics from the service wavefront's lanes — run in every case and are checked at the end)
    env = rware_amd.WarehouseVecEnv(B, library=LIB, stats=True, **ctor, **kw)
    info = env.engines[0].info
    if want:
        want(info)
    orc = OracleVecEnv(B, **kw)
    assert np.array_equal(env.reset(seed=seed)[0], orc.reset(seed=seed))
    rng = np.random.default_rng(0)
    for t in range(steps):
        a = rng.choice(5, size=(B, kw["n_agents"]), p=list(p)).astype(np.int32)
        o, r, d, _, _ = env.step(a)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        assert np.array_equal(o, o2) and np.array_equal(r, r2) and np.array_equal(d, d2.astype(bool)), (label, t)
    st, so = env.get_state(), orc.get_state()
    assert all(np.array_equal(st[k], so[k]) for k in so), label
    c = env.event_counters()
    assert np.array_equal(c["deliveries"], orc.stat_deliveries) and np.array_equal(c["failed_moves"], orc.stat_failed_moves), label
    env.close()
    print(f"tsan-run {label}: {steps} steps bit-exact (build kind {int(info.build_kind)}, E {int(info.envs_per_workgroup)}, pipe workgroups {int(info.pipe_workgroups)})", flush=True)

meta, z = gu.load_fixture("small-4ag")
be = EngineBackend(meta["E"], library=LIB, envs_per_workgroup=16, threads_per_workgroup=256, tile=4, **gu.ctor_kwargs(meta))
print("tsan-run golden small-4ag on the exact-shape build:", gu.replay(be, meta, z, steps=40), "steps bit-exact", flush=True)
be.env.close()
kw = rware_amd.env_kwargs("rware-medium-6ag-hard-v1"); kw["reward_type"] = 1; kw["max_steps"] = 12
against_oracle("generic kernel (medium-6ag-hard, 5 envs, 128 threads)", 5, 30, dict(envs_per_workgroup=4, threads_per_workgroup=128), kw)
# per-cell agent phases (kCell): the LDS atomicMax chain walks, the four-neighbour winner test, four-ballot goal flags — crowded on purpose
os.environ["RWARE_WIDE_E4"] = "0"   # (the 8-env builds of these shapes; rw_create's rule would give so small a batch the 4-env ones: next two cases)
kw = rware_amd.env_kwargs("rware-small-10ag-v1"); kw["reward_type"] = 1; kw["max_steps"] = 15
against_oracle("kCell small-10ag (agent-count-static, E 8)", 16, 40, {}, kw, want=lambda i: (int(i.build_kind) == 2 and int(i.envs_per_workgroup) == 8) or sys.exit("not the kCell build"))
kw = rware_amd.env_kwargs("rware-large-16ag-v1"); kw["reward_type"] = 2; kw["max_steps"] = 15
against_oracle("kCell large-16ag (agent-count-static, E 8, TWO_STAGE)", 16, 30, {}, kw, want=lambda i: (int(i.build_kind) == 2 and int(i.envs_per_workgroup) == 8) or sys.exit("not the 8-env kCell build"))
del os.environ["RWARE_WIDE_E4"]
# round 6: the same shapes as rw_create launches them at this batch — per-step kernel on 4-env workgroups (and, 16 agents, the fused rollout on 8-env ones)
kw = rware_amd.env_kwargs("rware-small-10ag-v1"); kw["reward_type"] = 1; kw["max_steps"] = 15
against_oracle("kCell small-10ag on the 4-env build (rw_create's rule)", 16, 25, {}, kw, want=lambda i: int(i.envs_per_workgroup) == 4 or sys.exit("not the 4-env build"))
kw = rware_amd.env_kwargs("rware-large-16ag-v1"); kw["reward_type"] = 1; kw["max_steps"] = 15
against_oracle("kCell large-16ag on the 4-env build (rw_create's rule)", 16, 20, {}, kw, want=lambda i: int(i.envs_per_workgroup) == 4 or sys.exit("not the 4-env build"))
kw = rware_amd.env_kwargs("rware-large-16ag-v1"); kw["sensor_range"] = 2; kw["reward_type"] = 1; kw["max_steps"] = 12
against_oracle("BASELINE config 5's kernel (large-16ag, sensor_range 2, exact, E 4)", 8, 25, {}, kw, want=lambda i: int(i.build_kind) == 1 or sys.exit("not the exact build"))
# the chunk-pipelined persistent builds (make PIPE=1): 2 persistent workgroups walk 4 chunks each, two LDS chunk buffers
os.environ["RWARE_PIPE_GRID"] = "2"
kw = rware_amd.env_kwargs("rware-small-4ag-v1"); kw["reward_type"] = 1; kw["max_steps"] = 12
against_oracle("PIPE small-4ag (8 chunks on 2 workgroups)", 128, 30, dict(pipe=True), kw, want=lambda i: int(i.pipe_workgroups) == 2 or sys.exit("not the pipelined build"))
kw = rware_amd.env_kwargs("rware-small-10ag-v1"); kw["reward_type"] = 1; kw["max_steps"] = 12
against_oracle("PIPE small-10ag (kCell inside the pipelined flow, 6 chunks on 2 workgroups)", 24, 25, dict(pipe=True), kw, want=lambda i: int(i.pipe_workgroups) == 2 or sys.exit("not the pipelined build"))
del os.environ["RWARE_PIPE_GRID"]
# rw_multi: a launcher thread per engine (RWARE_MULTI_THREADS=1 forces the thread mode on one device), sleep / wake handshake, create / destroy cycles
os.environ["RWARE_MULTI_THREADS"] = "1"
kw = rware_amd.env_kwargs("rware-tiny-2ag-v1"); kw["reward_type"] = 1
rounds = 0
import time
for cycle in range(6):
    env = rware_amd.WarehouseVecEnv(64, library=LIB, devices=[0] * 8, **kw)
    orc = OracleVecEnv(64, **kw)
    assert np.array_equal(env.reset(seed=cycle)[0], orc.reset(seed=cycle))
    multi = _capi.MultiEngine(env.engines)
    rng = np.random.default_rng(cycle)
    for t in range(180):
        a = rng.integers(0, 5, size=(64, 2)).astype(np.int32)
        bufs = [np.ascontiguousarray(a[lo:hi]) for lo, hi in env.shard_bounds]
        multi.step_device([b.ctypes.data for b in bufs])
        for e in env.engines:
            e.sync()
        rounds += 1
        if t % 45 == 44:
            time.sleep(0.05)   # (long enough for the launcher threads to go to sleep: the next round has to wake them)
        if t % 20 == 19 or t == 179:
            o2 = None
            # (the oracle steps every round; the engine's observations are compared every 20th)
        o2, r2, d2 = orc.step_autoreset(a, "next_step")
        if t % 20 == 19 or t == 179:
            o = env.observations()
            assert np.array_equal(o, o2), (cycle, t)
    multi.close(); env.close()
print(f"tsan-run rw_multi, 8 engines, launcher threads: {rounds} rounds over 6 create / destroy cycles bit-exact", flush=True)
EOP
echo "(a ThreadSanitizer WARNING line above = a reported race; none = clean)"

echo "== 2b. the whole emulation library (every table group, all generic kernels, the C-ABI host code) under -fsanitize=address,undefined: the emulated engine tests + the reference's KATs"
echo "   (device buffers are heap blocks with redzones; the LDS behind each launch's dynamic shared memory is poisoned: an out-of-bounds index in a kernel or in the host code faults)"
AFLAGS="-DRW_NO_JIT -DRW_WITH_PIPE=1 -DRW_STATS_BUILD=1 -O1 -g -std=c++17 -fPIC -pthread -Itests/emu -Wno-unknown-pragmas -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer"
A=$OUT/asan; mkdir -p $A
( for g in $(seq 0 18); do echo "g++ $AFLAGS -DRW_STATIC_GROUP=$g -c -x c++ $CSRC/rware_static.hip -o $A/s$g.o"; done
  for r in 1 2 3 4 5; do echo "g++ $AFLAGS -DRW_GENERIC_R=$r -c -x c++ $CSRC/rware_generic.hip -o $A/g$r.o"; done
  echo "g++ $AFLAGS -c -x c++ $CSRC/rware_capi.hip -o $A/capi.o"; echo "g++ $AFLAGS -c -x c++ $CSRC/rware_selftest.hip -o $A/selftest.o"
  echo "g++ $AFLAGS -c tests/emu/emu_globals.cpp -o $A/glob.o" ) | xargs -P 8 -I{} sh -c "{} 2>/dev/null"
g++ -shared -pthread -fsanitize=address,undefined -o $A/librware_emu_asan.so $A/*.o || exit 1
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1:allow_user_poisoning=1 UBSAN_OPTIONS=print_stacktrace=1 RWARE_ALLOW_STALE_PMC=1 \
  RWARE_EMU_LIB=$A/librware_emu_asan.so timeout 5400 python -m pytest tests/test_engine_emulated.py tests/test_reference_kats.py tests/test_gymnasium_boundary.py tests/test_event_counters.py -m "not gpu" -q -x -p no:cacheprovider -n 6 \
  --deselect tests/test_engine_emulated.py::test_rw_multi_launcher_threads_overlap_the_enqueues 2>&1 | tail -6
echo "   (deselected: the one wall-clock comparison of the suite — launcher threads against the in-call loop — which a sanitizer build distorts)"

echo "== 3. rware_jit.cpp under -fsanitize=address,undefined: no hipRTC, a library without the entry points, corrupt / truncated / foreign cache files, a cache directory that is not private"
make -s -C $CSRC rware_jit_sources.inc   # (the device headers as string literals: rware_jit.cpp includes them)
cat > $OUT/jit_driver.cpp <<'EOS'
#include <sys/stat.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include "rware_jit.h"
static const rw_jit::Shape kShape = {1, 11, 10, 3, 3, 32, 16, 256, 0, 0, 0 /* OBS_FLATTENED */, 0, 0u, -1, 1};
static int run(const char *what) {
    rw_jit::Result res;
    const bool ok = rw_jit::compile(kShape, "gfx950", &res);
    printf("jit-run %-52s -> %s | %.170s\n", what, ok ? "code object handed back" : "no code object (the engine keeps its ahead-of-time kernel)", res.log.c_str());
    return ok ? 1 : 0;
}
int main(int argc, char **argv) {
    const std::string dir = argv[1];
    setenv("RWARE_HOOKS", "1", 1);
    setenv("RWARE_JIT_CACHE", dir.c_str(), 1);
    mkdir(dir.c_str(), 0700);
    const std::string file = rw_jit::cache_file(kShape, "gfx950");
    if (file.empty() || file.compare(0, dir.size(), dir) != 0) { printf("jit-run unexpected cache file '%s'\n", file.c_str()); return 1; }
    setenv("RWARE_JIT_LIBRARY", "/nonexistent/libhiprtc.so", 1);
    run("no hipRTC library, empty cache");
    const char *cases[][2] = {{"", "cache file: empty"}, {"RWJIT1\n", "cache file: magic only"}, {"RWJIT1\nstep\n", "cache file: truncated header"},
                              {"RWJIT1\nstep\nroll\n", "cache file: header without a code object"}, {"RWJIT9\nstep\nroll\nxxxx", "cache file: foreign magic"},
                              {"\n\n\n\n\n\n\n\n", "cache file: newlines only"}};
    for (auto &c : cases) {
        FILE *f = fopen(file.c_str(), "wb");
        fwrite(c[0], 1, strlen(c[0]), f);
        fclose(f);
        run(c[1]);
    }
    { FILE *f = fopen(file.c_str(), "wb"); std::string big = "RWJIT1\nstep\nroll\n" + std::string(1 << 20, '\x7f'); fwrite(big.data(), 1, big.size(), f); fclose(f); }
    // (a well-formed header is all rware_jit checks: the bytes behind it go to hipModuleLoadData in rw_create, which rejects junk and falls back)
    run("cache file: well-formed header + 1 MiB of junk");
    chmod(dir.c_str(), 0777);
    run("the same file, cache directory group / world writable");
    chmod(dir.c_str(), 0700);
    unsetenv("RWARE_JIT_CACHE"); unsetenv("HOME");
    run("no HOME, no RWARE_JIT_CACHE");
    return 0;
}
EOS
g++ -O1 -g -std=c++17 -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -I$CSRC -o $OUT/jit_driver $OUT/jit_driver.cpp $CSRC/rware_jit.cpp -ldl 2>&1 | head -5
rm -rf $OUT/jitcache; ASAN_OPTIONS=detect_leaks=1:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 $OUT/jit_driver $OUT/jitcache 2>&1 | grep -v "^librware_hip:" | tail -15
echo "(an AddressSanitizer / runtime error line above = a finding; none = clean)"
