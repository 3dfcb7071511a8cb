#!/bin/bash
# Same-box A/B of library builds (run on the GPU box from the repo root, through gpurun):
#   build the variants in the build container first:  mkdir -p scratch/ab; (build) ; cp robotic-warehouse_amd/csrc/librware_hip.so scratch/ab/<name>.so
#   (scratch/ is git-ignored but travels with the gpurun snapshot), then
#   gpurun -- 'bash profiles/tools/ab.sh "base variant" 16384:0:0:6000 262144:0:0:300'
# Each spec is <batch>:<envs per workgroup>:<threads per workgroup>:<steps> (0:0 = the engine's own geometry); optional
# extra bench.py arguments in AB_ARGS (e.g. AB_ARGS="--observation-type 2").  Every variant runs twice, alternating, so a
# drift of the box shows up as a difference between the two passes.  Run-to-run spread on one box: ±0.01-0.02 us.
VARS=$1; shift
for r in 1 2; do
  for v in $VARS; do
    cp ${AB_DIR:-scratch/ab}/$v.so robotic-warehouse_amd/csrc/librware_hip.so
    echo -n "$v: "
    for spec in "$@"; do
      IFS=':' read -r b e t n <<< "$spec"
      python bench.py --no-cpu-baseline --no-hbm-regime --no-api-loop --no-fused-extra --no-submit-modes --no-sustained --no-soak --batch $b --steps $n --warmup 50 --envs-per-wg $e --threads-per-wg $t ${AB_ARGS:-} 2>/dev/null \
        | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('B', d['config']['envs_per_gpu'], 'E', d['config']['envs_per_workgroup'], 'T', d['config']['threads_per_workgroup'], 'us/step %.3f' % (d['ms_per_step']*1e3), 'kernel %.3f |' % (d['roofline']['kernel_ms_per_launch']*1e3), end=' ')"
    done
    echo
  done
done
