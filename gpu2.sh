set -x
export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/prof
timeout 300 python -m pytest tests -m gpu -x -q -k "invalid or torch or full_size" > $R/gpurun_out/pytest_gpu2.log 2>&1; echo "pytest rc=$?"; tail -5 $R/gpurun_out/pytest_gpu2.log
cd /tmp
timeout 300 rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof/trace -o bench -- python $R/bench.py --steps 1000 --warmup 100 --no-cpu-baseline > $R/gpurun_out/prof/trace.log 2>&1
tail -1 $R/gpurun_out/prof/trace.log
for c in FETCH_SIZE WRITE_SIZE; do
  timeout 300 rocprofv3 --pmc $c -d $R/gpurun_out/prof/pmc_$c -o bench -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof/pmc_$c.log 2>&1
  timeout 120 rocprofv3 --pmc $c -d $R/gpurun_out/prof/calib_$c -o calib -- $R/profiles/tools/copy_calib > $R/gpurun_out/prof/calib_$c.log 2>&1
done
timeout 300 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_VALU SQ_INSTS_LDS -d $R/gpurun_out/prof/pmc_sq -o bench -- python $R/bench.py --steps 60 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof/pmc_sq.log 2>&1
find $R/gpurun_out/prof -name "*.csv" | head -30
du -sh $R/gpurun_out/prof
