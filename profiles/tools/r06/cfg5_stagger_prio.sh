#!/bin/bash
m() { timeout 900 python profiles/tools/measure.py "$@" 2>&1 | grep -v amdgpu.ids; }
export RWARE_HOOKS=1
F="rware-large-16ag-v1:16384:0:auto:2 rware-large-16ag-v1:8192:0:auto:2 rware-large-16ag-v1:4096:0:auto:2 rware-large-16ag-v1:16384:8:auto:2 rware-large-16ag-v1:32768:0:auto:2 rware-medium-14ag-v1:16384:0:auto:2 rware-small-13ag-v1:16384:0:auto:2"
for r in 1 2; do for st in 0 20 40; do echo "== prio on (rule), stagger $st (pass $r)"; RWARE_STAGGER_TICKS=$st m $F; done; done
for r in 1 2; do echo "== prio off, stagger 40 (pass $r)"; RWARE_PRIO=0 RWARE_STAGGER_TICKS=40 m $F; done
