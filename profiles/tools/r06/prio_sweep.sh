#!/bin/bash
# RWARE_PRIO=0|1 (RWARE_HOOKS=1) on ONE library: us per step, two alternating passes, then the engine's own rule.
SPECS="rware-tiny-2ag-v1:4096 rware-tiny-2ag-v1:16384 rware-small-4ag-v1:4096 rware-small-4ag-v1:16384 rware-small-4ag-v1:32768 rware-small-4ag-v1:65536 rware-small-4ag-v1:131072 rware-small-4ag-v1:262144 rware-medium-6ag-hard-v1:8192 rware-medium-6ag-hard-v1:16384 rware-medium-6ag-hard-v1:65536 rware-small-8ag-v1:16384 rware-small-8ag-v1:65536 rware-small-10ag-v1:16384 rware-small-10ag-v1:65536 rware-small-12ag-v1:16384 rware-small-12ag-v1:65536 rware-medium-13ag-v1:16384 rware-small-14ag-v1:16384 rware-small-14ag-v1:65536 rware-tiny-14ag-v1:16384 rware-small-15ag-v1:16384 rware-large-16ag-v1:16384 rware-large-16ag-v1:32768 rware-large-16ag-v1:65536 rware-small-17ag-v1:16384 rware-small-19ag-v1:16384 rware-small-19ag-v1:65536 rware-large-16ag-v1:16384:0:auto:2 rware-large-16ag-v1:32768:0:auto:2 rware-large-16ag-v1:131072:0:auto:2 small-24ag:16384 layoutstr-3ag:16384 sr5-12ag-colheight5:4096"
for r in 1 2; do
  for p in 0 1; do
    echo "== prio $p (pass $r)"
    RWARE_HOOKS=1 RWARE_PRIO=$p timeout 600 python profiles/tools/measure.py $SPECS 2>&1 | grep -v amdgpu.ids
  done
done
echo "== rule"
timeout 600 python profiles/tools/measure.py $SPECS 2>&1 | grep -v amdgpu.ids
