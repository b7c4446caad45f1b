#!/usr/bin/env python3
"""Which allocation carries the scan kernel's levels (profiles/placement_probe.py: 3.10 / 3.17-3.20 / 3.27 / 3.43 ms for the
same table contents rebuilt in one process)?  Three parts, one process: (A) the table and the batches stay, only the filter
context (its scratch and work lists) is created anew; (B) table and context stay, the batches are generated anew; (C) context and
batches stay... which cannot be (a context belongs to its index), so: the table is rebuilt and a context with it, as in the
first probe, for comparison.  Junk allocations of changing sizes perturb the allocator before each trial.
usage: python profiles/placement_probe2.py [trials]"""
import sys

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
rng = np.random.default_rng(1)
junk = []


def perturb(t):
    junk.clear()
    torch.cuda.empty_cache()
    for _ in range(int(rng.integers(1, 6))):
        junk.append(torch.empty(int(rng.integers(1, 24)) << 30, dtype=torch.uint8, device=dev))
    if t % 2 == 1:
        junk.clear()
        torch.cuda.empty_cache()
    return sum(j.numel() for j in junk) >> 30


def new_proc(idx):
    return dcn.FilterProcessor(idx, max_batch_bases=batches[0].n_bases, max_batch_reads=batches[0].n_reads)


def scan_ms(proc, bs):
    def step(i):
        b = bs[i % 2]
        proc.filter_batch_device(b.d_bases.data_ptr(), b.d_offsets.data_ptr(), b.n_reads, b.n_bases, b.d_keep.data_ptr(),
                                 b.d_hits.data_ptr(), b.d_total.data_ptr())
    torch.cuda.synchronize()  # the batches were generated on torch's stream: order it before the context's
    for i in range(3):
        step(i)
    proc.synchronize()
    proc.set_profiling(2)
    for i in range(10):
        step(i)
    proc.synchronize()
    ms, nb = proc.profile()
    proc.set_profiling(0)
    return ms["scan"] / nb


print("(A) same table, same batches, a new context each time", flush=True)
for t in range(trials):
    held = perturb(t)
    proc = new_proc(index)
    print(f"  context {t}: {held:3d} GiB of junk held   scan {scan_ms(proc, batches):.3f} ms", flush=True)
    proc.close()
junk.clear()
torch.cuda.empty_cache()
print("(B) same table, same context, new batches each time", flush=True)
proc = new_proc(index)
print(f"  the batches so far                         scan {scan_ms(proc, batches):.3f} ms", flush=True)
for t in range(trials):
    del batches
    held = perturb(t)
    batches = B.make_batches("short", genome, 10_000_000, 5, dev, rotate=2)
    print(f"  batches {t}: {held:3d} GiB of junk held   scan {scan_ms(proc, batches):.3f} ms", flush=True)
proc.close()
junk.clear()
torch.cuda.empty_cache()
print("(C) the table rebuilt (and a context with it), same batches", flush=True)
for t in range(trials):
    index.close()
    held = perturb(t)
    index = dcn.Index.from_keys(keys, B.K, B.W, device=0)
    proc = new_proc(index)
    print(f"  table {t}: {held:3d} GiB of junk held   scan {scan_ms(proc, batches):.3f} ms", flush=True)
    proc.close()
