export TMPDIR=/tmp
R=$PWD
P=$R/gpurun_out/prof_v5
mkdir -p $P
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $P/trace_step -o bench -- python $R/bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $P/trace_step.log 2>&1
timeout 300 rocprofv3 --kernel-trace --stats -d $P/trace_fused -o bench -- python $R/bench.py --steps 1024 --warmup 64 --many 64 --no-cpu-baseline > $P/trace_fused.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d $P/pmc_step_$c -o bench -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline > $P/pmc_step_$c.log 2>&1
  timeout 300 rocprofv3 --pmc $c -d $P/pmc_fused_$c -o bench -- python $R/bench.py --steps 64 --warmup 64 --many 64 --no-cpu-baseline > $P/pmc_fused_$c.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $P/pmc_step_sq -o bench -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline > $P/pmc_step_sq.log 2>&1
grep -h '"metric"' $P/trace_step.log $P/trace_fused.log | python -c 'import sys,json
for l in sys.stdin:
    d=json.loads(l); print(d["config"]["submit"][:30], round(d["roofline"]["kernel_ms_per_launch"]*1000,2), "us/launch-step", round(d["value"]/1e9,2), "G")'
python $R/profiles/tools/timeline_probe.py rware-small-4ag-v1 16384 2>&1 | grep -v amdgpu.ids > $P/timeline_small4ag.txt
du -sh $P
