export TMPDIR=/tmp
R=$PWD
timeout 1200 python -m pytest tests -m gpu -x -q -k "fused or golden" > $R/gpurun_out/pytest_gpuJ.log 2>&1; echo "pytest rc=$?"; tail -2 $R/gpurun_out/pytest_gpuJ.log
b() { timeout 120 python bench.py --steps 1024 --warmup 64 --no-cpu-baseline "$@" 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["roofline"]["kernel_ms_per_launch"]*1000,2),"us", round(d["value"]/1e9,3),"G agent-steps/s frac",round(d["roofline"]["frac"],3), "spec", d["config"]["kernel_specialised"])'; }
echo "per-step:          $(b)"
echo "fused x64:         $(b --many 64)"
echo "fused x256:        $(b --many 256)"
echo "B=65536 per-step:  $(b --batch 65536)"
echo "B=65536 fused x64: $(b --batch 65536 --many 64)"
echo "large16 sr2 8192:  $(b --env-id rware-large-16ag-v1 --sensor-range 2 --batch 8192)"
echo "large16 sr2 16384: $(b --env-id rware-large-16ag-v1 --sensor-range 2 --batch 16384)"
echo "medium6h 8192:     $(b --env-id rware-medium-6ag-hard-v1 --batch 8192)"
echo "tiny2 4096:        $(b --env-id rware-tiny-2ag-v1 --batch 4096)"
