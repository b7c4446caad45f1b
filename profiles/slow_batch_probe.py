#!/usr/bin/env python3
"""VERDICT r3 weak 9, second half: one of the three rotated batches runs the scan kernel 3 % slower than the other two, every
time it comes round (profiles/r04_scan_dispatches.txt).  Is it what the batch CONTAINS or where its buffers ARE?

Per-batch scan time (HIP events around the scan kernel, one step at a time, 8 repetitions each, after warm-up), then the same
after the CONTENTS of the batches have been rotated through the SAME buffers (batch i's ASCII copied into batch (i+1)%3's buffer):
if the slow one follows the contents it is the reads, if it stays with the buffer it is placement.  Prints one JSON line."""
import json
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import deacon_server_amd as dcn  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    g = bench.make_host_genome(64_000_000, 3, dev)
    index, keys, host_keys, n_rand, _ = bench.build_index(g, bench.PANHUMAN_KEYS, 0)
    del keys
    batches = bench.make_batches("short", g, 10_000_000, 5, dev)
    proc = dcn.FilterProcessor(index, max_batch_bases=batches[0].n_bases, max_batch_reads=batches[0].n_reads)

    def step(b):
        proc.filter_batch_device(b.d_bases.data_ptr(), b.d_offsets.data_ptr(), b.n_reads, b.n_bases, b.d_keep.data_ptr(),
                                 b.d_hits.data_ptr(), b.d_total.data_ptr(), n_units=b.n_units)

    def per_batch(reps=8):
        out = []
        for b in batches:
            ms = []
            for _ in range(reps):
                proc.set_profiling(2)
                step(b)
                proc.synchronize()
                t, n = proc.profile()
                proc.set_profiling(False)
                ms.append(t["scan"] / max(n, 1))
            out.append({"scan_ms_median": float(np.median(ms)), "scan_ms_min": float(min(ms)), "minimizers": int(b.d_total.sum(dtype=torch.int64).item()),
                        "bases_ptr_mod_2MB": int(b.d_bases.data_ptr() % (2 << 20)), "keep_ptr": hex(b.d_keep.data_ptr())})
        return out

    for i in range(9):  # warm-up, rotated as bench.py does
        step(batches[i % 3])
    proc.synchronize()
    res = {"as_allocated": per_batch()}
    # in rotation (what the bench does): per-slot means over 30 steps
    ms = [[], [], []]
    for i in range(30):
        proc.set_profiling(2)
        step(batches[i % 3])
        proc.synchronize()
        t, n = proc.profile()
        proc.set_profiling(False)
        ms[i % 3].append(t["scan"] / max(n, 1))
    res["rotating"] = [float(np.median(m)) for m in ms]
    # rotate the contents through the buffers: buffer j now holds what buffer (j-1)%3 held
    saved = [b.d_bases.clone() for b in batches]
    for j in range(3):
        batches[j].d_bases.copy_(saved[(j - 1) % 3])
    torch.cuda.synchronize()
    res["contents_rotated_by_one_buffer"] = per_batch()
    print(json.dumps(res))


if __name__ == "__main__":
    main()
