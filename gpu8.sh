export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out
rocprofv3 -L 2>/dev/null | grep -i -E "icache|ifetch|SQ_INSTS_|SQ_WAVE|SQ_BUSY|SQC_" | head -60 > $R/gpurun_out/counters_list.txt
wc -l $R/gpurun_out/counters_list.txt
python profiles/tools/timeline_probe.py rware-small-4ag-v1 1024 16 256 2>&1 | grep -v amdgpu.ids | tee $R/gpurun_out/timeline_v4c.log
timeout 120 python bench.py --steps 1000 --warmup 50 --no-cpu-baseline 2>/dev/null | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["roofline"]["kernel_ms_per_launch"]*1000,2),"us")'
cd /tmp
timeout 300 rocprofv3 --pmc SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQC_ICACHE_MISSES_DUPLICATE -d $R/gpurun_out/prof2/pmc_icache -o bench -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline > $R/gpurun_out/pmc_icache.log 2>&1
tail -2 $R/gpurun_out/pmc_icache.log
