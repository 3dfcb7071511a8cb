export TMPDIR=/tmp
R=$PWD
timeout 900 python -m pytest tests -m gpu -x -q -k "soak" > $R/gpurun_out/pytest_gpuS.log 2>&1; echo "pytest rc=$?"; tail -3 $R/gpurun_out/pytest_gpuS.log
