#!/bin/bash
# Two ranks on ONE GPU (gloo instead of RCCL): exercises bench.py's N > 1 code path (sharding, barrier,
# max-over-ranks timing, frame sum, gather to rank 0) where only a single-GPU box is available.
cd "$(dirname "$0")/.."
SMILEHIP_DIST_BACKEND=gloo python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 \
  --master-port 29517 bench.py --gpus 2 --steps 5 --warmup 2
