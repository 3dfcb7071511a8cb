#!/bin/bash
m() { timeout 1800 python profiles/tools/measure.py "$@" 2>&1 | grep -v amdgpu.ids; }
export RWARE_HOOKS=1
T="rware-large-16ag-v1 rware-small-16ag-v1 rware-small-15ag-v1 rware-large-15ag-v1 rware-small-14ag-v1 rware-large-14ag-v1 rware-medium-13ag-v1 rware-large-13ag-v1 rware-tiny-14ag-v1 rware-tiny-16ag-v1"
B=""
for t in $T; do for b in 2048 4096 8192 12288 16384 24576 32768 49152 65536; do B="$B $t:$b"; done; done
for r in 1 2; do
echo "== the rule before (RWARE_WIDE_E4=0, RWARE_PRIO=0: 8-env workgroups, stagger) (pass $r)"; RWARE_WIDE_E4=0 RWARE_PRIO=0 m $B
echo "== the rule as shipped (pass $r)"; m $B
done
echo "== fused rollouts under the shipped rule"
m rware-large-16ag-v1:4096:0:auto:::32 rware-large-16ag-v1:32768:0:auto:::32 rware-medium-13ag-v1:8192:0:auto:::32
