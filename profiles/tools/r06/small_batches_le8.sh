#!/bin/bash
m() { timeout 1800 python profiles/tools/measure.py "$@" 2>&1 | grep -v amdgpu.ids; }
A=""
for t in rware-small-4ag-v1 rware-tiny-4ag-v1 rware-medium-4ag-v1; do for b in 1024 2048 4096 8192; do A="$A $t:$b $t:$b:8"; done; done
for t in rware-small-2ag-v1 rware-tiny-2ag-v1; do for b in 1024 2048 4096 8192; do A="$A $t:$b $t:$b:8"; done; done
for t in rware-small-6ag-v1 rware-medium-6ag-hard-v1 rware-small-8ag-v1; do for b in 1024 2048 4096; do A="$A $t:$b $t:$b:16"; done; done
for r in 1 2; do echo "== pass $r"; m $A; done
