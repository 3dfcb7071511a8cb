#!/bin/bash
m() { timeout 1800 python profiles/tools/measure.py "$@" 2>&1 | grep -v amdgpu.ids; }
export RWARE_HOOKS=1
F=""
for t in rware-large-16ag-v1 rware-medium-13ag-v1 rware-small-14ag-v1 rware-small-15ag-v1; do for b in 4096 8192 32768; do for e in 4 8; do F="$F $t:$b:$e:auto:::32"; done; done; done
for r in 1 2; do echo "== fused rollouts, explicit geometry, rollout priority on (pass $r)"; RWARE_STAGGER_TICKS=0 m $F; done
