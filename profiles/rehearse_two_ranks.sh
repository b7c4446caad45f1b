#!/bin/bash
# bench.py's N > 1 code path (barriers, MAX of the elapsed time, counter all-reduce, rank-0 line) rehearsed with two
# ranks on the one GPU of the box: gloo instead of RCCL, both ranks on device 0, a 50 M-key index to keep it short.
DCN_BENCH_BACKEND=gloo DCN_BENCH_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 --reads 1000000 --index-keys 50000000
