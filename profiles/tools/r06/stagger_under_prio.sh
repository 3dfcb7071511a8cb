#!/bin/bash
m() { timeout 900 python profiles/tools/measure.py "$@" 2>&1 | grep -v amdgpu.ids; }
export RWARE_HOOKS=1
F="rware-small-4ag-v1:65536 rware-small-4ag-v1:131072 rware-medium-6ag-hard-v1:32768 rware-medium-6ag-hard-v1:65536 rware-small-6ag-v1:65536 rware-small-8ag-v1:32768 rware-small-8ag-v1:65536 rware-small-8ag-v1:131072 rware-small-10ag-v1:32768 rware-small-10ag-v1:65536 rware-small-12ag-v1:32768 rware-tiny-2ag-v1:65536 rware-small-2ag-v1:131072 rware-small-3ag-v1:65536 rware-small-5ag-v1:65536 rware-small-7ag-v1:65536 rware-large-8ag-v1:65536 rware-medium-4ag-v1:65536 rware-tiny-4ag-v1:131072 rware-small-9ag-v1:65536 rware-small-11ag-v1:32768"
for r in 1 2; do for st in 0 25; do echo "== prio rule, stagger $st (pass $r)"; RWARE_STAGGER_TICKS=$st m $F; done; done
