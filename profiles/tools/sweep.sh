#!/bin/bash
# Collects the rocprofv3 evidence behind every number DESIGN.md quotes (run on the GPU box from the repo root):
#   bash profiles/tools/sweep.sh r02            -> gpurun_out/r02_sweep.json, gpurun_out/r02_sweep.txt, gpurun_out/pmc_traffic.json
#   bash profiles/tools/sweep.sh r02 headline   -> only the configs whose name contains "headline"
# Per config three separate passes of the same bench.py command: kernel trace (--kernel-trace --stats), then FETCH_SIZE,
# then WRITE_SIZE (never --pmc together with an API trace; FETCH_SIZE and WRITE_SIZE do not fit one pass).
set -u
TAG=${1:-sweep}
ONLY=${2:-}
ROOT=$(pwd)
OUT=$ROOT/gpurun_out/sweep_$TAG
mkdir -p "$OUT"
export TMPDIR=/tmp
COMMON="--no-cpu-baseline --no-hbm-regime --no-api-loop --no-fused-extra --no-submit-modes --no-sustained --no-soak"
# name | tape steps | trace steps | bench.py arguments
# (small kernels get long trace runs: rocprofv3 slows the host's first few thousand launches down to ~10 us each, which
#  leaves idle gaps in front of a ~8 us kernel; sweep_collect.py reports the back-to-back dispatches separately)
CONFIGS=(
  "headline_small4_B16384|256|6000|"
  "cfg2_tiny2_B4096|256|6000|--env-id rware-tiny-2ag-v1 --batch 4096"
  "cfg4_medium6hard_B8192|256|6000|--env-id rware-medium-6ag-hard-v1 --batch 8192"
  "cfg5_large16_r2_B16384|64|400|--env-id rware-large-16ag-v1 --sensor-range 2 --batch 16384"
  "small4_B65536|64|400|--batch 65536"
  "hbm_small4_B262144|16|200|--batch 262144"
  "hbm_large16_r2_B32768|16|200|--env-id rware-large-16ag-v1 --sensor-range 2 --batch 32768"
  "cfg4_medium6hard_B65536|16|200|--env-id rware-medium-6ag-hard-v1 --batch 65536"
  "cfg5_large16_r2_B131072|8|100|--env-id rware-large-16ag-v1 --sensor-range 2 --batch 131072"
  "image_small4_B16384|256|6000|--observation-type 2"
  "image_tiny2_B4096|256|6000|--env-id rware-tiny-2ag-v1 --batch 4096 --observation-type 2"
  "msg2_small4_B16384|256|6000|--msg-bits 2"
  "small8_B16384|256|6000|--env-id rware-small-8ag-v1"
  "small10_B16384|256|3000|--env-id rware-small-10ag-v1"
  "large16_B16384|64|1500|--env-id rware-large-16ag-v1"
  "fused64_small4_B16384|256|1024|--many 64"
)
cd /tmp
# calibration of FETCH_SIZE / WRITE_SIZE on known byte counts (16 B / lane and 4 B / lane streams)
if [ -z "$ONLY" ] || [[ "calib" == *"$ONLY"* ]]; then
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O2 -o /tmp/copy_calib "$ROOT/profiles/tools/copy_calib.hip" 2> "$OUT/calib_build.log"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d "$OUT/calib/pmc_$c" -o calib -- /tmp/copy_calib > "$OUT/calib_$c.log" 2>&1
  done
fi
for entry in "${CONFIGS[@]}"; do
  IFS='|' read -r name tape steps args <<< "$entry"
  if [ -n "$ONLY" ] && [[ "$name" != *"$ONLY"* ]]; then continue; fi
  D="$OUT/$name"; mkdir -p "$D"
  export RWARE_BENCH_TAPE_STEPS=$tape
  echo "== $name" >&2
  # small kernels: the profiled host cannot issue single launches as fast as a ~7 us kernel retires them (a third of the
  # dispatches then start late, on an idle chip) — replay the same per-step launches from a HIP graph for the trace pass
  SUBMIT=""
  if [ "$steps" -ge 1000 ] && [[ "$args" != *"--many"* ]] && [ "${SWEEP_TRACE_SUBMIT:-graph}" = "graph" ]; then SUBMIT="--submit graph"; fi
  rocprofv3 --kernel-trace --stats -d "$D/trace" -o bench -- python "$ROOT/bench.py" $COMMON $args $SUBMIT --steps $steps --warmup 50 > "$D/trace.out" 2> "$D/trace.err"
  for c in FETCH_SIZE WRITE_SIZE; do
    rocprofv3 --pmc $c -d "$D/pmc_$c" -o bench -- python "$ROOT/bench.py" $COMMON $args --steps 30 --warmup 10 > "$D/pmc_$c.out" 2> "$D/pmc_$c.err"
  done
done
cd "$ROOT"
python profiles/tools/sweep_collect.py "$OUT" "$TAG"
# the per-config summaries that get committed under profiles/ (before the large databases are dropped)
for d in "$OUT"/*/; do n=$(basename "$d"); [ "$n" = calib ] && continue; python profiles/tools/summarize_rocpd.py "$d" > "gpurun_out/${TAG}_${n}_kernels.txt" 2>/dev/null; done
python profiles/tools/summarize_rocpd.py "$OUT/calib" > "gpurun_out/${TAG}_calibration_kernels.txt" 2>/dev/null
# keep only what is worth carrying back (the databases can be large)
find "$OUT" -name '*.db' -size +4M -delete
