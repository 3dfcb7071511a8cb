#!/bin/bash
# One rocprofv3 --pmc pass of bench.py (GPU box, repo root):  bash profiles/tools/pmc_pass.sh <tag> "<counters>" <bench.py args...>
# Writes gpurun_out/<tag>_pmc.txt (appends).  Counter sets must fit one pass (8 SQ slots; see MI355X_MICROARCH.md).
set -u
TAG=$1; COUNTERS=$2; shift 2
ROOT=$(pwd); OUT=$ROOT/gpurun_out/pmc_$TAG; mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
N=$(ls -d "$OUT"/pass* 2>/dev/null | wc -l)
rocprofv3 --pmc $COUNTERS -d "$OUT/pass$N" -o bench -- python "$ROOT/bench.py" --no-cpu-baseline --no-hbm-regime --no-api-loop --no-fused-extra --no-submit-modes --no-sustained --no-soak --steps 30 --warmup 10 "$@" > "$OUT/pass$N.out" 2> "$OUT/pass$N.err"
cd "$ROOT"
python profiles/tools/summarize_rocpd.py "$OUT" > "gpurun_out/${TAG}_pmc.txt" 2>&1
find "$OUT" -name '*.db' -size +4M -delete
