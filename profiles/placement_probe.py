#!/usr/bin/env python3
"""Run-to-run state of the probe rate (the same tree runs the headline scan in 3.13-3.46 ms from box to box
and flips inside one process).  Is it WHERE the 34 GB table lands?  The table is built, measured (uniformly random probes and
the headline's scan stage on one resident batch), freed, the allocator is perturbed with junk allocations of various sizes,
and the table is built again -- several times in one process.
usage: python profiles/placement_probe.py [trials]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
import bench as B
import deacon_server_amd as dcn

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 6
dev = torch.device("cuda", 0)
genome = B.make_host_genome(64_000_000, 3, dev)
index, keys, hk, nr, _ = B.build_index(genome, B.PANHUMAN_KEYS, 0)
batches = B.make_batches("short", genome, 10_000_000, 5, dev, rotate=2)
P = {"abs": 2, "rel": 0.01, "deplete": False}


def measure(idx, tag):
    rnd = idx.probe_ceiling(None, 1 << 27, reps=3)
    proc = dcn.FilterProcessor(idx, max_batch_bases=batches[0].n_bases, max_batch_reads=batches[0].n_reads)
    def step(i):
        b = batches[i % 2]
        proc.filter_batch_device(b.d_bases.data_ptr(), b.d_offsets.data_ptr(), b.n_reads, b.n_bases, b.d_keep.data_ptr(),
                                 b.d_hits.data_ptr(), b.d_total.data_ptr())
    for i in range(3):
        step(i)
    proc.synchronize()
    proc.set_profiling(2)
    for i in range(10):
        step(i)
    proc.synchronize()
    ms, nb = proc.profile()
    proc.close()
    print(f"{tag:34s} table at {idx.table_bytes / 2**30:.0f} GiB  random probes {rnd / 1e9:5.1f} G/s   scan {ms['scan'] / nb:.3f} ms", flush=True)


measure(index, "first table of the process")
junk = []
rng = np.random.default_rng(0)
for t in range(trials):
    index.close()
    # perturb: free whatever junk is held, take new junk of other sizes (torch caching allocator bypassed by empty_cache)
    junk.clear()
    torch.cuda.empty_cache()
    for _ in range(int(rng.integers(1, 6))):
        junk.append(torch.empty(int(rng.integers(1, 24)) << 30, dtype=torch.uint8, device=dev))
    if t % 2 == 1:
        junk.clear()
        torch.cuda.empty_cache()
    index = dcn.Index.from_keys(keys, B.K, B.W, device=0)
    measure(index, f"rebuilt, {sum(j.numel() for j in junk) >> 30} GiB of junk held")
