#!/bin/bash
m() { timeout 1800 python profiles/tools/measure.py "$@" 2>&1 | grep -v amdgpu.ids; }
A=""
for t in rware-small-9ag-v1 rware-small-10ag-v1 rware-small-11ag-v1 rware-small-12ag-v1 rware-medium-10ag-v1 rware-large-12ag-v1 rware-large-9ag-v1; do for b in 512 4096 6144 8192 12288; do A="$A $t:$b $t:$b:4"; done; done
for t in rware-small-17ag-v1 rware-small-19ag-v1 rware-large-18ag-easy-v1 rware-medium-17ag-v1; do for b in 512 1024 6144; do A="$A $t:$b $t:$b:4"; done; done
for r in 1 2; do echo "== pass $r"; m $A; done
echo "== fused rollouts"
F=""
for t in rware-small-10ag-v1 rware-small-12ag-v1 rware-small-9ag-v1 rware-small-17ag-v1 rware-small-19ag-v1; do for b in 2048 4096; do F="$F $t:$b:8:auto:::32 $t:$b:4:auto:::32"; done; done
m $F
