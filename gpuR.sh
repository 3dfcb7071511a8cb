export TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests -m gpu -x -q > $R/gpurun_out/pytest_gpuR.log 2>&1; echo "pytest rc=$?"; tail -3 $R/gpurun_out/pytest_gpuR.log
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -1
b() { timeout 120 python bench.py --steps 1024 --warmup 64 --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["roofline"]["kernel_ms_per_launch"]*1000,2),"us", round(d["value"]/1e9,3),"G frac", round(d["roofline"]["frac"],3), "; fused", round(d.get("fused_rollout",{}).get("ms_per_step",0)*1000,2), "us spec", d["config"]["kernel_specialised"])'; }
echo "small-4ag:   $(b)"
