#!/bin/bash
# TEST INFRASTRUCTURE (sanitizers): the oracle under AddressSanitizer + UBSan, and the product sources built
# for host threads (tests/emu) under ThreadSanitizer.  CPU only.   bash oracle/sanitize.sh > profiles/rNN_sanitizers.txt
set -u
cd "$(dirname "$0")/.."
OUT=/tmp/rware_san; mkdir -p $OUT
echo "== oracle/rware_oracle.c under -fsanitize=address,undefined: golden traces + RNG vs numpy + live reference check"
gcc -O1 -g -std=c11 -shared -fPIC -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer -o $OUT/librware_oracle_asan.so oracle/rware_oracle.c || exit 1
LD_PRELOAD=$(gcc -print-file-name=libasan.so) ASAN_OPTIONS=detect_leaks=0:abort_on_error=1 UBSAN_OPTIONS=print_stacktrace=1 \
  RWARE_ORACLE_SO=$OUT/librware_oracle_asan.so timeout 1800 python -m pytest tests/test_oracle_golden.py tests/test_oracle_rng_numpy.py tests/test_oracle_vs_reference.py -q -x -p no:cacheprovider 2>&1 | tail -5
echo "== tests/emu (the engine's kernel + C-ABI host code on host threads) under -fsanitize=thread: one golden replay on an exact-shape build, one oracle comparison on the generic kernel"
CSRC=robotic-warehouse_amd/csrc
FLAGS="-O1 -g -std=c++17 -fPIC -pthread -Itests/emu -Wno-unknown-pragmas -fsanitize=thread"
for r in 1 2; do g++ $FLAGS -DRW_GENERIC_R=$r -c -x c++ $CSRC/rware_generic.hip -o $OUT/g$r.o 2>/dev/null & done
g++ $FLAGS -c -x c++ $CSRC/rware_capi.hip -o $OUT/capi.o 2>/dev/null &
g++ $FLAGS -c tests/emu/emu_globals.cpp -o $OUT/glob.o &
wait
# (sensor ranges 3..5 are not needed by the two cases below: their pick functions are stubbed)
cat > $OUT/stub.cpp <<'EOS'
#include <hip/hip_runtime.h>
#include "rware_kernel_table.h"
namespace rw_tab { step_kernel_t generic_r3(bool, bool, bool, bool) { return nullptr; } step_kernel_t generic_r4(bool, bool, bool, bool) { return nullptr; } step_kernel_t generic_r5(bool, bool, bool, bool) { return nullptr; } }
EOS
g++ $FLAGS -I$CSRC -c $OUT/stub.cpp -o $OUT/stub.o 2>/dev/null
g++ -shared -pthread -fsanitize=thread -o $OUT/librware_emu_tsan.so $OUT/capi.o $OUT/g1.o $OUT/g2.o $OUT/glob.o $OUT/stub.o || exit 1
LD_PRELOAD=$(gcc -print-file-name=libtsan.so) TSAN_OPTIONS="halt_on_error=0:report_signal_unsafe=0:history_size=2" timeout 3000 python - <<'EOP' 2>&1 | grep -E "WARNING: ThreadSanitizer|SUMMARY|tsan-run" | sort | uniq -c | head -20
import sys
sys.path[:0] = [".", "oracle", "tests"]
import numpy as np
import golden_util as gu
from engine_backend import EngineBackend
import rware_amd
from rware_oracle import OracleVecEnv
LIB = "/tmp/rware_san/librware_emu_tsan.so"
meta, z = gu.load_fixture("small-4ag")
be = EngineBackend(meta["E"], library=LIB, envs_per_workgroup=16, threads_per_workgroup=256, tile=4, **gu.ctor_kwargs(meta))
print("tsan-run golden small-4ag on the exact-shape build:", gu.replay(be, meta, z, steps=40), "steps bit-exact")
be.env.close()
kw = rware_amd.env_kwargs("rware-medium-6ag-hard-v1"); kw["reward_type"] = 1; kw["max_steps"] = 12
env = rware_amd.WarehouseVecEnv(5, library=LIB, envs_per_workgroup=4, threads_per_workgroup=128, **kw)
orc = OracleVecEnv(5, **kw)
assert np.array_equal(env.reset(seed=3)[0], orc.reset(seed=3))
rng = np.random.default_rng(0)
for t in range(30):
    a = rng.choice(5, size=(5, 6), p=[.1, .55, .1, .1, .15]).astype(np.int32)
    o, r, d, _, _ = env.step(a)
    o2, r2, d2 = orc.step_autoreset(a, "next_step")
    assert np.array_equal(o, o2) and np.array_equal(r, r2)
print("tsan-run generic kernel vs oracle: 30 steps bit-exact")
EOP
echo "(a ThreadSanitizer WARNING line above = a reported race; none = clean)"
