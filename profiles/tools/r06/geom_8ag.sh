#!/bin/bash
m() { timeout 900 python profiles/tools/measure.py "$@" 2>&1 | grep -v amdgpu.ids; }
G=""
for t in rware-small-8ag-v1 rware-medium-8ag-v1 rware-tiny-8ag-v1 rware-large-8ag-v1; do for b in 32768 65536 131072; do for e in 8 16; do G="$G $t:$b:$e"; done; done; done
for t in rware-medium-6ag-hard-v1 rware-small-6ag-v1; do for b in 32768 65536 131072; do for e in 8 16; do G="$G $t:$b:$e"; done; done; done
for t in rware-small-10ag-v1 rware-small-12ag-v1 rware-small-9ag-v1; do for b in 32768 65536; do for e in 8 16; do G="$G $t:$b:$e"; done; done; done
for r in 1 2; do echo "== rule (pass $r)"; m $G; done
