#!/bin/bash
# GPU box: exercise bench.py's multi-rank path with two ranks sharing device 0 (the box has one GPU)
cd ${GRAFT_REPO_ROOT:-/root/repo}
export MASTER_ADDR=127.0.0.1 MASTER_PORT=29511 WORLD_SIZE=2 LOCAL_RANK=0
RANK=1 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 --scale 0.5 > /tmp/r1.log 2>&1 &
RANK=0 timeout 300 python bench.py --gpus 2 --steps 10 --warmup 2 --scale 0.5 2>&1 | tail -2 | cut -c1-900
wait
echo "rank1 log:"; tail -3 /tmp/r1.log | cut -c1-300
# and the launcher the driver uses, with one process
timeout 300 python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 1 --steps 10 --warmup 2 --no-cpu-baseline 2>&1 | tail -1 | cut -c1-300
