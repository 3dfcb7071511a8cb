#!/bin/bash
# End-of-round evidence pass, ONE gpurun call (run on the GPU box from the repo root, ≈12 GPU-minutes):
#   /usr/local/graft/bin/gpurun --timeout 3000 -- 'bash profiles/tools/final_pass.sh r06'
# THE LAST SOURCE-TOUCHING ACT OF A ROUND: tests/test_host_layer.py fails while profiles/pmc_traffic.json and the kernel sources disagree.
# Order matters: the GPU suite first (a red suite voids everything after it), then the sweep (which writes
# gpurun_out/pmc_traffic.json), then THAT file into profiles/ BEFORE bench.py runs (bench.py reports `roofline.traffic`
# only when the file's kernel_sources_sha matches the sources it is running), then the un-profiled lines.
# Afterwards, in the build container: copy gpurun_out/{pmc_traffic.json, <tag>_*} into profiles/, regenerate the
# per-config summaries with summarize_rocpd.py (see the loop at the bottom), update DESIGN.md §6, commit.
set -u
TAG=${1:-rNN}
mkdir -p gpurun_out
# (-rs: the reason of every skip lands in the committed file — VERDICT r4 item 7: a skipped guard must be visible)
python -m pytest tests -m gpu -q --tb=short -rs --durations=10 2>&1 | tail -80 > gpurun_out/${TAG}_gpu_suite_full.txt; tail -3 gpurun_out/${TAG}_gpu_suite_full.txt | tee gpurun_out/${TAG}_gpu_suite.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -3 | tee gpurun_out/${TAG}_smoke.txt
bash profiles/tools/sweep.sh $TAG 2>&1 | tail -15
cp gpurun_out/pmc_traffic.json profiles/pmc_traffic.json
python bench.py > gpurun_out/${TAG}_bench_default.json 2> gpurun_out/${TAG}_bench_default.err
python bench.py --steps 20 --warmup 5 > gpurun_out/${TAG}_bench_k20.json 2> /dev/null
python profiles/tools/timeline_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_timeline.txt
python profiles/tools/timeline_probe.py rware-small-4ag-v1 262144 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_timeline_B262144.txt
TL_OBS_TYPE=2 python profiles/tools/timeline_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_timeline_image.txt
python profiles/tools/timeline_probe.py rware-medium-6ag-hard-v1 8192 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_timeline_medium6.txt
TL_SENSOR_RANGE=2 python profiles/tools/timeline_probe.py rware-large-16ag-v1 16384 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_timeline_large16_r2.txt
python profiles/tools/timeline_probe.py rware-small-8ag-v1 16384 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_timeline_small8.txt
python profiles/tools/timeline_probe.py rware-small-10ag-v1 16384 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_timeline_small10.txt
python profiles/tools/timeline_probe.py rware-tiny-2ag-v1 4096 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_timeline_tiny2.txt
python profiles/tools/grid_table.py 16384 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_grid_table_B16384.txt
# run-time specialised builds (hipRTC) against the generic kernel on unregistered shapes
python profiles/tools/measure.py layoutstr-3ag:16384:0:auto::off layoutstr-3ag:16384:0:auto::force small-4ag-colheight5:16384:0:auto::off \
  small-4ag-colheight5:16384:0:auto::force sr5-12ag-colheight5:8192:0:auto::off sr5-12ag-colheight5:8192:0:auto::force \
  rware-small-4ag-v1:16384:0:auto:2:off rware-small-4ag-v1:16384:0:auto:2:force rware-small-4ag-v1:16384:0:auto:3:off \
  rware-small-4ag-v1:16384:0:auto:3:force small-24ag:16384:0:auto::off small-24ag:16384:0:auto::force 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_jit_vs_generic.txt
# instruction mix and instruction-cache traffic of the headline kernel and two wide ones (separate PMC passes)
I1="SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_WAVES"
I2="SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY"
I3="SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE SQC_TC_INST_REQ SQC_DCACHE_REQ SQC_DCACHE_MISSES"
for c in "$I1" "$I2" "$I3"; do bash profiles/tools/pmc_pass.sh ${TAG}_headline "$c"; done
for c in "$I1" "$I2"; do bash profiles/tools/pmc_pass.sh ${TAG}_small10 "$c" --env-id rware-small-10ag-v1; bash profiles/tools/pmc_pass.sh ${TAG}_cfg5 "$c" --env-id rware-large-16ag-v1 --sensor-range 2; done
# (round 5's pipelined-build A/B and resident-mailbox probe are history: profiles/r05_pipe_*.txt, r05_resident_probe.txt; the pipelined kernels
#  are only in a `make PIPE=1` library since round 6)
python profiles/tools/k20_probe.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_k20_probe.txt
python profiles/tools/api_rates.py 2>&1 | grep -v amdgpu.ids > gpurun_out/${TAG}_api_rates.txt
bash profiles/tools/unprofiled.sh $TAG > gpurun_out/${TAG}_unprofiled.txt
cat gpurun_out/${TAG}_unprofiled.txt
python - <<PY
import json
for f in ("gpurun_out/${TAG}_bench_default.json", "gpurun_out/${TAG}_bench_k20.json"):
    d = json.loads(open(f).read().strip().splitlines()[-1])
    print(f, "G=%.3f" % (d["value"] / 1e9), "us/step=%.3f" % (d["ms_per_step"] * 1e3), "sustained=%.3f" % (d["sustained"]["ms_per_step"] * 1e3),
          "traffic=", d["roofline"]["traffic"], "frac=%.3f (%s, %s)" % (d["roofline"]["frac"], d["roofline"]["basis"], d["roofline"]["regime"]), "frac_algorithmic=%.3f" % d["roofline"]["frac_algorithmic"])
PY
# in the build container afterwards:
#   cp gpurun_out/pmc_traffic.json profiles/; cp gpurun_out/${TAG}_* profiles/
#   for d in gpurun_out/sweep_${TAG}/*/; do n=$(basename $d); python profiles/tools/summarize_rocpd.py $d > profiles/${TAG}_${n}_kernels.txt; done
