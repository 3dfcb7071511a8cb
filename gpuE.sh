export TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests -m gpu -x -q > $R/gpurun_out/pytest_gpuE.log 2>&1; echo "pytest rc=$?"; tail -2 $R/gpurun_out/pytest_gpuE.log
b() { timeout 120 python bench.py --steps 1024 --warmup 64 --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["roofline"]["kernel_ms_per_launch"]*1000,2),"us", round(d["value"]/1e9,3),"G agent-steps/s frac",round(d["roofline"]["frac"],3), "spec", d["config"]["kernel_specialised"])'; }
echo "per-step:          $(b)"
echo "fused x8:          $(b --many 8)"
echo "fused x64:         $(b --many 64)"
echo "fused x256:        $(b --many 256)"
echo "fused x64 E8 T128 (generic): $(b --many 64 --envs-per-wg 8 --threads-per-wg 128)"
echo "fused x64 E4 T64 (generic):  $(b --many 64 --envs-per-wg 4 --threads-per-wg 64)"
