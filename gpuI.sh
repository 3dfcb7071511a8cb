export TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests -m gpu -x -q > $R/gpurun_out/pytest_gpuI.log 2>&1; echo "pytest rc=$?"; tail -2 $R/gpurun_out/pytest_gpuI.log
b() { timeout 120 python bench.py --steps 1024 --warmup 64 --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["roofline"]["kernel_ms_per_launch"]*1000,2),"us", round(d["value"]/1e9,3),"G agent-steps/s frac",round(d["roofline"]["frac"],3), "spec", d["config"]["kernel_specialised"])'; }
echo "per-step:          $(b)"
echo "fused x64:         $(b --many 64)"
echo "generic E8 T128:   $(b --envs-per-wg 8 --threads-per-wg 128)"
python profiles/tools/timeline_probe.py rware-small-4ag-v1 16384 2>&1 | grep -v amdgpu.ids | tail -12
