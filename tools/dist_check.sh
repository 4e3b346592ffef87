#!/bin/bash
# Exercises bench.py through torch.distributed.run exactly as the driver launches it (1 rank on a 1-GPU box).
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 2 --warmup 1 --no-cpu-baseline
