#!/bin/bash
# Same-box A/B of the launch rules: round 6's first session (RWARE_PRIO=0 RWARE_PRIO_ROLLOUT=0: no wavefront priority, hence the start stagger of
# rounds 4 / 6a; 8-agent exact builds on 16-env workgroups beyond 16384 envs) against the rules as shipped.  ONE library; two alternating passes.
m() { timeout 900 python profiles/tools/measure.py "$@" 2>&1 | grep -v amdgpu.ids; }
export RWARE_HOOKS=1
S="rware-tiny-2ag-v1:4096 rware-small-4ag-v1:16384 rware-medium-6ag-hard-v1:8192 rware-medium-6ag-hard-v1:65536 rware-large-16ag-v1:16384:0:auto:2 rware-large-16ag-v1:131072:0:auto:2 rware-small-4ag-v1:32768 rware-small-4ag-v1:65536 rware-small-4ag-v1:131072 rware-small-4ag-v1:262144 rware-small-10ag-v1:16384 rware-small-12ag-v1:16384 rware-small-17ag-v1:16384 rware-small-19ag-v1:16384 rware-small-3ag-v1:65536 rware-small-5ag-v1:65536 rware-small-6ag-v1:65536 rware-small-7ag-v1:65536 rware-medium-13ag-v1:16384 rware-large-16ag-v1:16384"
OLD8="rware-small-8ag-v1:32768:16 rware-small-8ag-v1:65536:16 rware-tiny-8ag-v1:65536:16 rware-medium-8ag-v1:65536:16"
NEW8="rware-small-8ag-v1:32768 rware-small-8ag-v1:65536 rware-tiny-8ag-v1:65536 rware-medium-8ag-v1:65536"
F="rware-small-4ag-v1:16384:0:auto:::64 rware-medium-6ag-hard-v1:8192:0:auto:::64 rware-small-8ag-v1:16384:0:auto:::64 rware-small-10ag-v1:16384:0:auto:::32 rware-large-16ag-v1:16384:0:auto:2::32"
for r in 1 2; do
  echo "== first-session rules (pass $r)"; RWARE_PRIO=0 RWARE_PRIO_ROLLOUT=0 m $S $OLD8 $F
  echo "== shipped rules (pass $r)"; m $S $NEW8 $F
done
