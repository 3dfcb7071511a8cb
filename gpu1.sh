set -x
mkdir -p gpurun_out
export TMPDIR=/tmp
rocminfo | grep -m2 -E "gfx|Marketing" 
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/smoke.log 2>&1; echo "smoke rc=$?"; tail -3 gpurun_out/smoke.log
timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -15 gpurun_out/pytest_gpu.log
timeout 300 python bench.py --steps 2000 --warmup 100 > gpurun_out/bench1.log 2>&1; echo "bench rc=$?"; tail -3 gpurun_out/bench1.log
timeout 300 python bench.py --steps 2000 --warmup 100 --many 64 --no-cpu-baseline > gpurun_out/bench_many.log 2>&1; tail -1 gpurun_out/bench_many.log
