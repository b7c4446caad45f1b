#!/bin/bash
# bench.py's N > 1 code path (CPU binding, key array shared through tmpfs, barriers, MAX of the elapsed time, counter
# all-reduce, per-rank PCIe-inclusive legs, rank-0 line) rehearsed with two ranks on the one GPU of the box: gloo instead
# of RCCL, both ranks on device 0, a 50 M-key index and 2 M-read batches to keep it short.
# usage: profiles/rehearse_two_ranks.sh [output.json]
out=${1:-/dev/stdout}
DCN_BENCH_BACKEND=gloo DCN_BENCH_SINGLE_DEVICE=1 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 \
  --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 2 --steps 6 --warmup 2 --reads 2000000 --index-keys 50000000 | grep "^{" > "$out"   # (gloo prints its connection banner on stdout)
