#!/bin/bash
m() { timeout 900 python profiles/tools/measure.py "$@" 2>&1 | grep -v amdgpu.ids; }
G=""
for s in cached stream; do G="$G rware-large-16ag-v1:16384:0:$s:2 rware-large-16ag-v1:8192:0:$s:2 rware-large-16ag-v1:16384:0:$s rware-small-16ag-v1:16384:0:$s rware-small-17ag-v1:16384:0:$s rware-small-19ag-v1:16384:0:$s rware-small-12ag-v1:16384:0:$s rware-small-14ag-v1:16384:0:$s rware-small-4ag-v1:16384:0:$s rware-large-16ag-v1:32768:0:$s:2 rware-small-19ag-v1:32768:0:$s"; done
for t in rware-tiny-2ag-v1 rware-small-2ag-v1; do for b in 8192 16384 32768 65536; do for e in 16 32; do G="$G $t:$b:$e"; done; done; done
for t in rware-small-7ag-v1 rware-small-5ag-v1 rware-large-8ag-v1 rware-large-6ag-v1; do for b in 4096 8192; do for e in 8 16; do G="$G $t:$b:$e"; done; done; done
for r in 1 2; do echo "== rule (pass $r)"; m $G; done
