#!/bin/bash
m() { timeout 1800 python profiles/tools/measure.py "$@" 2>&1 | grep -v amdgpu.ids; }
export RWARE_HOOKS=1
T="rware-large-16ag-v1 rware-small-16ag-v1 rware-small-15ag-v1 rware-large-15ag-v1 rware-small-14ag-v1 rware-large-14ag-v1 rware-medium-13ag-v1 rware-large-13ag-v1"
A=""; B=""
for t in $T; do for b in 2048 4096 8192 12288 16384 24576 32768 49152 65536 98304; do A="$A $t:$b:4"; B="$B $t:$b"; done; done
echo "== E8 rule (pass 1)"; m $B
echo "== E8 s0 p0 (pass 1)"; RWARE_PRIO=0 RWARE_STAGGER_TICKS=0 m $B
echo "== E8 s0 p1 (pass 1)"; RWARE_PRIO=1 RWARE_STAGGER_TICKS=0 m $B
echo "== E4 s0 p1 (pass 1)"; RWARE_PRIO=1 RWARE_STAGGER_TICKS=0 m $A
echo "== E4 s0 p0 (pass 1)"; RWARE_PRIO=0 RWARE_STAGGER_TICKS=0 m $A
