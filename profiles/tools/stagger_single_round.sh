#!/bin/bash
# Start stagger on launches that are resident at once (one round of workgroups): us per step against the stagger slot width
# (RWARE_STAGGER_TICKS x 10 ns; the k-th of the first eight workgroups a CU receives starts k slots late).  GPU box, repo root:
#   bash profiles/tools/stagger_single_round.sh > gpurun_out/r06_stagger_single_round.txt
SPECS="rware-small-10ag-v1:16384 rware-large-16ag-v1:16384 rware-large-16ag-v1:16384:0:auto:2 rware-small-8ag-v1:16384 rware-medium-6ag-hard-v1:8192 rware-small-4ag-v1:16384 rware-small-12ag-v1:16384 rware-small-19ag-v1:16384"
for r in 1 2; do
for t in ${STAGGER_LIST:-0 12 25 40 60 90}; do
  echo "== stagger ticks $t (pass $r)"
  RWARE_HOOKS=1 RWARE_STAGGER_TICKS=$t python profiles/tools/measure.py $SPECS 2>&1 | grep -v amdgpu.ids
done
done
