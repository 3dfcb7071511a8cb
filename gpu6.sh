export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
timeout 900 python -m pytest tests -m gpu -x -q > $R/gpurun_out/pytest_gpu6.log 2>&1; echo "pytest rc=$?"; tail -3 $R/gpurun_out/pytest_gpu6.log
for cfg in "16 256" "16 128" "16 64" "8 128" "8 64" "4 64" "32 256" "32 128" "64 256"; do
  set -- $cfg
  echo "E=$1 T=$2: $(timeout 120 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline --envs-per-wg $1 --threads-per-wg $2 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["roofline"]["kernel_ms_per_launch"]*1000,2),"us", round(d["value"]/1e9,3),"G agent-steps/s frac",round(d["roofline"]["frac"],3))')"
done 2>&1 | tee $R/gpurun_out/sweep6.log
python profiles/tools/timeline_probe.py rware-small-4ag-v1 16384 16 256 2>&1 | grep -v amdgpu.ids | tee $R/gpurun_out/timeline_v4.log
