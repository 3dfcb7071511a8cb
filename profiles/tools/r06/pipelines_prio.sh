#!/bin/bash
export RWARE_HOOKS=1
T="rware-small-4ag-v1 rware-medium-6ag-hard-v1 rware-small-8ag-v1 rware-small-10ag-v1 rware-small-12ag-v1 rware-medium-13ag-v1 rware-large-16ag-v1 rware-small-19ag-v1 rware-large-16ag-v1:2"
for r in 1 2; do for p in 0 1; do echo "== RWARE_PRIO=$p (pass $r)"; RWARE_PRIO=$p timeout 900 python profiles/tools/grid_pipelines.py 16384 $T 2>&1 | grep -v amdgpu.ids; done; done
echo "== RWARE_PRIO=0/1 at 32768"
for p in 0 1; do echo "== RWARE_PRIO=$p B 32768"; RWARE_PRIO=$p timeout 900 python profiles/tools/grid_pipelines.py 32768 rware-small-4ag-v1 rware-medium-6ag-hard-v1 rware-small-8ag-v1 2>&1 | grep -v amdgpu.ids; done
