export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out/prof3
cd /tmp
timeout 300 rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU -d $R/gpurun_out/prof3/pmc_a -o bench -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof3/a.log 2>&1
timeout 300 rocprofv3 --pmc SQ_INSTS_SMEM SQ_INSTS_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_BRANCH SQ_WAIT_INST_LDS SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS -d $R/gpurun_out/prof3/pmc_b -o bench -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof3/b.log 2>&1
timeout 300 rocprofv3 --pmc SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_INST_CYCLES_SALU SQ_THREAD_CYCLES_VALU SQ_IFETCH SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE -d $R/gpurun_out/prof3/pmc_c -o bench -- python $R/bench.py --steps 40 --warmup 10 --no-cpu-baseline > $R/gpurun_out/prof3/c.log 2>&1
tail -1 $R/gpurun_out/prof3/c.log
