#!/usr/bin/env python3
"""How fast can the HBM-resident set be probed in isolation?  (one 32-byte group read per key, full occupancy)
usage: python profiles/probe_microbench.py [index_keys] [probes]"""
import sys, time
import numpy as np
import torch
sys.path.insert(0, ".")
import deacon_server_amd as dcn

n_keys = int(sys.argv[1]) if len(sys.argv) > 1 else 409_913_780
n_probe = int(sys.argv[2]) if len(sys.argv) > 2 else 64_000_000
rng = np.random.default_rng(1)
keys = rng.integers(1, 2**63, n_keys, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
idx = dcn.Index.from_keys(keys, 31, 15)
dev = torch.device("cuda:0")
probe = torch.from_numpy(keys[:n_probe // 2].view(np.int64)).to(dev)
miss = torch.randint(0, 2**62, (n_probe - probe.numel(),), device=dev, dtype=torch.int64) * 2  # even: never present
q = torch.cat([probe, miss])[torch.randperm(n_probe, device=dev)]
out = torch.zeros(n_probe, dtype=torch.uint8, device=dev)
torch.cuda.synchronize()
for it in range(3):
    idx.contains_device(q.data_ptr(), n_probe, out.data_ptr())
torch.cuda.synchronize()
t = time.perf_counter()
R = 10
for it in range(R):
    idx.contains_device(q.data_ptr(), n_probe, out.data_ptr())
torch.cuda.synchronize()
dt = (time.perf_counter() - t) / R
print(f"keys={idx.n_keys:,} probes={n_probe:,} hit_frac={out.float().mean().item():.3f} "
      f"time={dt*1e3:.3f} ms  {n_probe/dt/1e9:.1f} G probes/s  {n_probe*64/dt/1e12:.2f} TB/s @64B  {n_probe*8/dt/1e9:.0f} GB/s algorithmic")
