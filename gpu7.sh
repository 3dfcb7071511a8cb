export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/pytest_gpu7.log 2>&1; echo "pytest rc=$?"; tail -3 $R/gpurun_out/pytest_gpu7.log
b() { timeout 120 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["roofline"]["kernel_ms_per_launch"]*1000,2),"us", round(d["value"]/1e9,3),"G agent-steps/s frac",round(d["roofline"]["frac"],3))'; }
echo "default:           $(b)"
echo "DEV_KERNARG=1:     $(HIP_FORCE_DEV_KERNARG=1 b)"
echo "many=64:           $(b --many 64)"
echo "B=65536:           $(b --batch 65536)"
echo "B=4096:            $(b --batch 4096)"
python profiles/tools/timeline_probe.py rware-small-4ag-v1 16384 16 256 2>&1 | grep -v amdgpu.ids | tee $R/gpurun_out/timeline_v4b.log
python profiles/tools/timeline_probe.py rware-small-4ag-v1 1024 16 256 2>&1 | grep -v amdgpu.ids | tee -a $R/gpurun_out/timeline_v4b.log
