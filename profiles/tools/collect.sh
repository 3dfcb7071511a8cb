#!/bin/bash
# Collects the rocprofv3 evidence kept under profiles/ (run on the GPU box from the repo root):
#   bash profiles/tools/collect.sh r01_v7   ->  gpurun_out/r01_v7_kernels.txt (+ the raw rocpd databases)
# Kernel trace and each PMC counter set run as separate passes (never --pmc together with API traces).
set -u
TAG=${1:-profile}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/prof_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
cd /tmp
STEP="python $ROOT/bench.py --no-cpu-baseline --no-hbm-regime --no-api-loop --no-fused-extra --no-submit-modes"
FUSED="python $ROOT/bench.py --no-cpu-baseline --no-hbm-regime --no-api-loop --no-fused-extra --no-submit-modes --many 64 --steps 1024 --warmup 64"
rocprofv3 --kernel-trace --stats -d "$OUT/trace_step" -o bench -- $STEP --steps 1000 --warmup 100 > "$OUT/trace_step.log" 2>&1
rocprofv3 --kernel-trace --stats -d "$OUT/trace_fused" -o bench -- $FUSED > "$OUT/trace_fused.log" 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --pmc $c -d "$OUT/pmc_step_$c" -o bench -- $STEP --steps 50 --warmup 20 > "$OUT/pmc_step_$c.log" 2>&1
  rocprofv3 --pmc $c -d "$OUT/pmc_fused_$c" -o bench -- $FUSED > "$OUT/pmc_fused_$c.log" 2>&1
done
rocprofv3 --pmc SQ_WAVES SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM -d "$OUT/pmc_step_insts" -o bench -- $STEP --steps 50 --warmup 20 > "$OUT/pmc_step_insts.log" 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_INST_ANY SQ_WAIT_ANY -d "$OUT/pmc_step_cycles" -o bench -- $STEP --steps 50 --warmup 20 > "$OUT/pmc_step_cycles.log" 2>&1
cd "$ROOT"
python profiles/tools/summarize_rocpd.py "$OUT" > "gpurun_out/${TAG}_kernels.txt" 2>&1
# keep only the text summary and logs in what travels back (the databases can be large)
find "$OUT" -name '*.db' -size +8M -delete
tail -40 "gpurun_out/${TAG}_kernels.txt"
