import sys, numpy as np
sys.path.insert(0,'.'); sys.path.insert(0,'tests')
import torch
import deacon_server_amd as dcn
from oracle import oracle as O
from conftest import random_reads
genome = random_reads(np.random.default_rng(1), 1, 200_000, 200_000)[0]
oidx = O.Index.build([genome], k=31, w=15)
gidx = dcn.Index.from_keys(oidx.keys(), 31, 15)
rng = np.random.default_rng(39)
n = 400_000
g = np.frombuffer(genome, dtype=np.uint8)
starts = rng.integers(0, len(genome) - 150, n)
host = rng.random(n) < 0.5
mat = g[starts[:, None] + np.arange(150)[None, :]].copy()
rnd = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n, 150))]
mat[~host] = rnd[~host]
bases = mat.reshape(-1)
offsets = (np.arange(n + 1, dtype=np.uint64) * np.uint64(150))
proc = dcn.FilterProcessor(gidx, max_batch_bases=n * 150, max_batch_reads=n)
res = [proc.filter_batch(bases, offsets) for _ in range(4)]
want = O.filter_batch(oidx, bases, offsets, threads=8)
for i, r in enumerate(res):
    for j, nm in enumerate(("keep", "hits", "total")):
        d = np.nonzero(r[j] != want[j])[0]
        print("call", i, nm, "diffs", len(d), d[:10], (d // 1).min() if len(d) else None, d.max() if len(d) else None)
        if len(d): print("   got", r[j][d[:10]], "want", want[j][d[:10]])
