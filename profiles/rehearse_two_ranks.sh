#!/bin/bash
# bench.py's N > 1 code path rehearsed with two ranks on the one GPU of the box, started the way the driver starts the 1-GPU
# run -- a plain `python3 bench.py --gpus 2 ...`, which launches its own ranks (self_launch_if_needed) -- at the driver's sizes:
# CPU binding, key array shared through tmpfs, barriers, MAX of the elapsed time, counter all-reduce, per-rank PCIe-inclusive
# legs, rank 0's compact line.  gloo instead of RCCL and both ranks on device 0 (RCCL refuses two ranks on one device), so
# the C ABI's communicator is not part of it (tests/test_distributed.py covers that with one rank).
# usage: profiles/rehearse_two_ranks.sh [output.json] [extra bench.py args]
out=${1:-/dev/stdout}; shift
DCN_BENCH_BACKEND=gloo DCN_BENCH_SINGLE_DEVICE=1 python3 bench.py --gpus 2 --steps 20 --warmup 5 --detail gpurun_out/r4_rehearse2_detail.json "$@" | tail -n 1 > "$out"
