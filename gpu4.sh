export TMPDIR=/tmp
R=$PWD
python profiles/tools/timeline_probe.py rware-small-4ag-v1 16384 16 256 2>&1 | grep -v amdgpu.ids | tee $R/gpurun_out/timeline_v2.log
python profiles/tools/timeline_probe.py rware-small-4ag-v1 16384 8 128 2>&1 | grep -v amdgpu.ids | tee -a $R/gpurun_out/timeline_v2.log
