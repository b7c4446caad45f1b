#!/usr/bin/env python3
"""Does the ORDER of allocations in a fresh process decide which state the scan runs in?  (profiles/placement_probe.py: the
same table runs the same batches in 3.19 or 3.35-3.45 ms depending on what else was allocated when it was built.)
  A: index, then the batches (many temporaries through torch's allocator), then the filter context   (bench.py's order)
  B: index, then the filter context, then the batches                                                 (a server's order)
usage: python profiles/placement_order.py A|B        (one fresh process per call; run alternately)"""
import sys

import torch

sys.path.insert(0, ".")
import bench as B
import deacon_server_amd as dcn

order = sys.argv[1]
dev = torch.device("cuda", 0)
genome = B.make_host_genome(64_000_000, 3, dev)
index, keys, hk, nr, _ = B.build_index(genome, B.PANHUMAN_KEYS, 0)
del keys
n_reads = 10_000_000
proc = None
if order == "B":
    proc = dcn.FilterProcessor(index, max_batch_bases=n_reads * B.READ_LEN, max_batch_reads=n_reads)
batches = B.make_batches("short", genome, n_reads, 5, dev)
torch.cuda.synchronize()
if proc is None:
    proc = dcn.FilterProcessor(index, max_batch_bases=n_reads * B.READ_LEN, max_batch_reads=n_reads)


def step(i):
    b = batches[i % 3]
    proc.filter_batch_device(b.d_bases.data_ptr(), b.d_offsets.data_ptr(), b.n_reads, b.n_bases, b.d_keep.data_ptr(),
                             b.d_hits.data_ptr(), b.d_total.data_ptr())


for i in range(3):
    step(i)
proc.synchronize()
proc.set_profiling(2)
for i in range(12):
    step(i)
proc.synchronize()
ms, nb = proc.profile()
rnd = index.probe_ceiling(None, 1 << 27, reps=3)
print(f"order {order}: scan {ms['scan'] / nb:.3f} ms   random probes {rnd / 1e9:.1f} G/s", flush=True)
