#!/usr/bin/env python3
"""f1 measurement: index build (sequence -> index-side minimizers -> device set) and index file load, GPU vs the
CPU oracle on the same synthetic genome.  usage: python profiles/index_build_bench.py [genome_bases] [cpu_bases]"""
import os, sys, tempfile, time
import numpy as np
import torch  # noqa: F401
sys.path.insert(0, ".")
import deacon_server_amd as dcn
from oracle import oracle as O

n = int(sys.argv[1]) if len(sys.argv) > 1 else 512_000_000
n_cpu = int(sys.argv[2]) if len(sys.argv) > 2 else 64_000_000
rng = np.random.default_rng(1)
chroms = []
left = n
while left > 0:  # "chromosomes" of up to 128 Mbp with a few N runs
    ln = min(left, 128_000_000)
    s = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, ln)].copy()
    for _ in range(20):
        a = int(rng.integers(0, ln - 2000)); s[a:a + int(rng.integers(1, 2000))] = ord("N")
    chroms.append(s)
    left -= ln
dcn.Index.build([chroms[0][:1_000_000]])  # warm up (module load, first allocations)
t = time.perf_counter()
idx = dcn.Index.build(chroms, 31, 15)
dt = time.perf_counter() - t
print(f"GPU index build: {n/1e6:.0f} Mbp -> {idx.n_keys:,} minimizers in {dt:.2f} s = {n/dt/1e6:.0f} Mbp/s (host memory in, device set out)")
t = time.perf_counter()
o = O.Index.build([chroms[0][:n_cpu]])
dtc = time.perf_counter() - t
print(f"CPU oracle build (1 thread): {n_cpu/1e6:.0f} Mbp -> {len(o):,} minimizers in {dtc:.2f} s = {n_cpu/dtc/1e6:.1f} Mbp/s")
g = dcn.Index.build([chroms[0][:n_cpu]])
print("same key set on the CPU sample:", sorted(g.keys().tolist()) == sorted(o.keys().tolist()))
with tempfile.TemporaryDirectory() as d:
    path = os.path.join(d, "bench.idx")
    t = time.perf_counter(); idx.write(path); dtw = time.perf_counter() - t
    size = os.path.getsize(path)
    t = time.perf_counter(); idx2 = dcn.Index.from_file(path); dtl = time.perf_counter() - t
    print(f"index file: {size/1e6:.0f} MB; write {dtw:.2f} s ({size/dtw/1e6:.0f} MB/s); load to device set {dtl:.2f} s "
          f"({size/dtl/1e6:.0f} MB/s, {idx2.n_keys/dtl/1e6:.1f} M keys/s)")
    assert idx2.n_keys == idx.n_keys
