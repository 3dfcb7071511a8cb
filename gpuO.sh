export TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests -m gpu -x -q > $R/gpurun_out/pytest_gpuO.log 2>&1; echo "pytest rc=$?"; tail -2 $R/gpurun_out/pytest_gpuO.log
b() { timeout 120 python bench.py --steps 1024 --warmup 64 --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["roofline"]["kernel_ms_per_launch"]*1000,2),"us", round(d["value"]/1e9,3),"G frac", round(d["roofline"]["frac"],3), "; fused", round(d.get("fused_rollout",{}).get("ms_per_step",0)*1000,2), "us spec", d["config"]["kernel_specialised"], d["config"]["envs_per_workgroup"], d["config"]["threads_per_workgroup"])'; }
echo "small-4ag (full static):   $(b)"
echo "small-3ag (size-static):   $(b --env-id rware-small-3ag-v1)"
echo "small-5ag (size-static):   $(b --env-id rware-small-5ag-v1)"
echo "small-5ag generic E16T128: $(b --env-id rware-small-5ag-v1 --envs-per-wg 16 --threads-per-wg 128)"
echo "small-8ag (size-static):   $(b --env-id rware-small-8ag-v1)"
echo "large-16ag r=1 (size-st):  $(b --env-id rware-large-16ag-v1)"
echo "tiny-4ag-easy (size-st):   $(b --env-id rware-tiny-4ag-easy-v1)"
