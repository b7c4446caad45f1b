#!/usr/bin/env python3
"""PCIe-inclusive rate of the host-buffer entry point dcn_filter_batch (pageable host memory -> pinned staging ->
hipMemcpyAsync -> kernels -> results back).  usage: python profiles/host_path_bench.py [reads] [index_keys]"""
import sys, time
import numpy as np
import torch  # noqa: F401  (load torch's HIP runtime first)
sys.path.insert(0, ".")
import deacon_server_amd as dcn

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
n_keys = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
rng = np.random.default_rng(1)
keys = rng.integers(1, 2**63, n_keys, dtype=np.uint64)
idx = dcn.Index.from_keys(keys, 31, 15)
bases = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n_reads * 150)]
offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(150)
proc = dcn.FilterProcessor(idx, max_batch_bases=n_reads * 150, max_batch_reads=n_reads)
def run(label, b, o):
    for _ in range(2):
        proc.filter_batch(b, o)
    R = 5
    t = time.perf_counter()
    for _ in range(R):
        keep, hits, total = proc.filter_batch(b, o)
    dt = (time.perf_counter() - t) / R
    print(f"host path ({label}): {n_reads} reads x 150 bp per call, {dt*1e3:.1f} ms/call = "
          f"{n_reads*150/dt/1e9:.2f} Gbp/s (ASCII over PCIe, results copied back)")
    return keep


import os
k0 = run(f"pageable, DCN_HOST_THREADS={os.environ.get('DCN_HOST_THREADS', 'default')}", bases, offsets)
pb = dcn.PinnedBuffer(len(bases), np.uint8)
po = dcn.PinnedBuffer(len(offsets), np.uint64)
pb.array[:] = bases
po.array[:] = offsets
k1 = run("page-locked buffers from dcn_host_alloc", pb.array, po.array)
assert (k0 == k1).all()
