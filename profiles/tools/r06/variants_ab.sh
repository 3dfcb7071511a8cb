#!/bin/bash
# Same-box A/B of library variants through measure.py (un-profiled us per step), two alternating passes.
#   gpurun -- 'bash profiles/tools/r06/variants_ab.sh "base preH stH slot" > gpurun_out/r06_prio_ab.txt 2>&1'
VARS=$1
SPECS=${SPECS:-"rware-small-10ag-v1:16384 rware-large-16ag-v1:16384 rware-large-16ag-v1:16384:0:auto:2 rware-small-8ag-v1:16384 rware-small-4ag-v1:16384 rware-small-12ag-v1:16384 rware-small-19ag-v1:16384 rware-medium-13ag-v1:16384 rware-medium-6ag-hard-v1:8192 rware-small-4ag-v1:65536"}
cp robotic-warehouse_amd/csrc/librware_hip.so /tmp/keep.so
for r in 1 2; do
  for v in $VARS; do
    cp scratch/ab/$v.so robotic-warehouse_amd/csrc/librware_hip.so
    echo "== $v (pass $r)"
    timeout 300 python profiles/tools/measure.py $SPECS 2>&1 | grep -v amdgpu.ids
  done
done
cp /tmp/keep.so robotic-warehouse_amd/csrc/librware_hip.so
