#!/bin/bash
# The un-profiled bench.py line of every config of the sweep (GPU box, repo root) -> gpurun_out/<tag>_unprofiled.jsonl
TAG=${1:-r02}
OUT=gpurun_out/${TAG}_unprofiled.jsonl
: > $OUT
COMMON="--no-cpu-baseline --no-hbm-regime --no-api-loop --no-fused-extra --no-submit-modes --no-soak"
run() { name=$1; tape=$2; shift 2; RWARE_BENCH_TAPE_STEPS=$tape python bench.py $COMMON "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d['sweep_config']='$name'; print(json.dumps(d))" >> $OUT; }
run headline_small4_B16384 256 --steps 6000 --warmup 200
run cfg2_tiny2_B4096 256 --env-id rware-tiny-2ag-v1 --batch 4096 --steps 6000 --warmup 200
run cfg4_medium6hard_B8192 256 --env-id rware-medium-6ag-hard-v1 --batch 8192 --steps 6000 --warmup 200
run cfg5_large16_r2_B16384 64 --env-id rware-large-16ag-v1 --sensor-range 2 --batch 16384 --steps 600 --warmup 50
run small4_B65536 64 --batch 65536 --steps 1000 --warmup 100
run small4_B131072 32 --batch 131072 --steps 500 --warmup 50
run hbm_small4_B262144 16 --batch 262144 --steps 300 --warmup 30
run hbm_large16_r2_B32768 16 --env-id rware-large-16ag-v1 --sensor-range 2 --batch 32768 --steps 300 --warmup 30
run cfg4_medium6hard_B65536 16 --env-id rware-medium-6ag-hard-v1 --batch 65536 --steps 500 --warmup 50
run cfg5_large16_r2_B131072 8 --env-id rware-large-16ag-v1 --sensor-range 2 --batch 131072 --steps 150 --warmup 20
run image_small4_B16384 256 --observation-type 2 --steps 6000 --warmup 200
run image_tiny2_B4096 256 --env-id rware-tiny-2ag-v1 --batch 4096 --observation-type 2 --steps 6000 --warmup 200
run msg2_small4_B16384 256 --msg-bits 2 --steps 6000 --warmup 200
run small8_B16384 256 --env-id rware-small-8ag-v1 --steps 6000 --warmup 200
run small10_B16384 256 --env-id rware-small-10ag-v1 --steps 3000 --warmup 200
run large16_B16384 64 --env-id rware-large-16ag-v1 --steps 1500 --warmup 100
RWARE_BENCH_TAPE_STEPS=256 python bench.py --no-cpu-baseline --no-hbm-regime --no-api-loop --no-fused-extra --no-submit-modes --no-sustained --no-soak --many 64 --steps 4096 --warmup 64 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); d['sweep_config']='fused64_small4_B16384'; print(json.dumps(d))" >> $OUT
python - <<PY
import json
print("%-28s %10s %12s %12s" % ("config", "us/step", "sustained", "G agent-st/s"))
for l in open("$OUT"):
    d = json.loads(l)
    print("%-28s %10.3f %12s %12.2f" % (d["sweep_config"], d["ms_per_step"] * 1e3, ("%.3f" % (d["sustained"]["ms_per_step"] * 1e3)) if "sustained" in d else "-", d["value"] / 1e9))
PY
