export TMPDIR=/tmp
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 500 --warmup 50 2>&1 | tail -3 | cut -c1-600
