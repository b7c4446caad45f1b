#!/usr/bin/env python3
"""Does a second pipeline context per GPU pay?  The headline batches alternately through 1 and 2 contexts (own stream
and scratch each, same index): with two, pack + plan of one batch can overlap the scan of the other.
usage: python profiles/two_contexts_probe.py [short|long|mixed]
Round 3: for long reads the stages around the scan are not small any more (pack 0.37 + plan 0.12 in front, distinct 0.5
behind it, per 1.5 Gbp), and unlike the scan they are light kernels that can share the chip."""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import bench as B
import deacon_server_amd as dcn

dev = torch.device("cuda", 0)
genome = B.make_host_genome(64_000_000, 3, dev)
index, keys, hk, nr, _ = B.build_index(genome, 409_913_780, 0)
del keys
kind = sys.argv[1] if len(sys.argv) > 1 else "short"
batches = B.make_batches(kind, genome, 4_000_000 if kind == "short" else 10_000_000, 5, dev)
P = {"abs": 2, "rel": 0.01, "deplete": False}
for n_ctx in (1, 2, 3, 1, 2):
    procs = [dcn.FilterProcessor(index, max_batch_bases=max(b.n_bases for b in batches), max_batch_reads=max(b.n_reads for b in batches)) for _ in range(n_ctx)]
    if kind != "short":
        for p in procs:
            p.reserve_records(max(b.n_bases for b in batches) // 6)
    def step(i):
        b = batches[i % 3]
        procs[i % n_ctx].filter_batch_device(b.d_bases.data_ptr(), b.d_offsets.data_ptr(), b.n_reads, b.n_bases, b.d_keep.data_ptr(),
                                             b.d_hits.data_ptr(), b.d_total.data_ptr())
    for i in range(6):
        step(i)
    for p in procs:
        p.synchronize()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    K = 30
    for i in range(K):
        step(i)
    for p in procs:
        p.synchronize()
    dt = time.perf_counter() - t0
    print(f"{n_ctx} context(s): {dt / K * 1e3:.3f} ms/step = {K * batches[0].n_bases / dt / 1e9:.1f} Gbp/s")
    for p in procs:
        p.close()
