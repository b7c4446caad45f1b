#!/usr/bin/env python3
"""PCIe-inclusive rate of the host entry points (dcn_filter_batch / _submit / _wait / _packed*) on a small index,
for tuning the pipeline itself: chunk size (DCN_CHUNK_BASES), host threads (DCN_HOST_THREADS), transports.
usage: python profiles/host_path_bench.py [reads] [index_keys] [calls]
The driver-run numbers come from bench.py's host_path block (panhuman-sized index); this script is the quick A/B."""
import ctypes as C
import os
import sys
import time

import numpy as np
import torch  # noqa: F401  (load torch's HIP runtime first)

sys.path.insert(0, ".")
import deacon_server_amd as dcn  # noqa: E402

torch.set_num_threads(min(16, torch.get_num_threads()))  # (the CPU quota of a 1-GPU job, not the machine's 256 threads)

n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 2_000_000
n_keys = int(sys.argv[2]) if len(sys.argv) > 2 else 50_000_000
calls = int(sys.argv[3]) if len(sys.argv) > 3 else 12
rng = np.random.default_rng(1)
keys = rng.integers(1, 2**63, n_keys, dtype=np.uint64)
idx = dcn.Index.from_keys(keys, 31, 15)
n_bases = n_reads * 150
host = [np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n_bases)] for _ in range(2)]
offsets = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(150)
proc = dcn.FilterProcessor(idx, max_batch_bases=n_bases, max_batch_reads=n_reads)
lib, P = dcn._native.lib(), proc._params()
G = (n_bases + 31) // 32
pins = []
for h in host:
    pb = dcn.PinnedBuffer(n_bases, np.uint8)
    pb.array[:] = h
    pp, pm = dcn.PinnedBuffer(2 * G, np.uint32), dcn.PinnedBuffer(G, np.uint32)
    dcn._native.check(lib.dcn_pack_ascii(h.ctypes.data, n_bases, pp.array.ctypes.data, pm.array.ctypes.data, None))
    pins.append((pb, pp, pm))
poff = dcn.PinnedBuffer(n_reads + 1, np.uint64)
poff.array[:] = offsets
pkeep = [dcn.PinnedBuffer(n_reads, np.uint8) for _ in range(2)]
keep = [np.zeros(n_reads, np.uint8) for _ in range(2)]


def call(kind, i, kp, submit):
    pb, pp, pm = pins[i % 2]
    t = C.c_uint64()
    if kind == "pageable":
        a = (proc._h, host[i % 2].ctypes.data, offsets.ctypes.data, None, n_reads, C.byref(P), kp, None, None)
        rc = lib.dcn_filter_batch_submit(*a, C.byref(t)) if submit else lib.dcn_filter_batch(*a)
    elif kind == "pinned":
        a = (proc._h, pb.array.ctypes.data, poff.array.ctypes.data, None, n_reads, C.byref(P), kp, None, None)
        rc = lib.dcn_filter_batch_submit(*a, C.byref(t)) if submit else lib.dcn_filter_batch(*a)
    else:
        a = (proc._h, pp.array.ctypes.data, pm.array.ctypes.data, poff.array.ctypes.data, None, n_reads, C.byref(P), kp, None, None)
        rc = lib.dcn_filter_batch_packed_submit(*a, C.byref(t)) if submit else lib.dcn_filter_batch_packed(*a)
    dcn._native.check(rc)
    return t.value


ref = None
print(f"{n_reads} reads x 150 bp per call, DCN_CHUNK_BASES={os.environ.get('DCN_CHUNK_BASES', 'default')}, "
      f"DCN_HOST_THREADS={os.environ.get('DCN_HOST_THREADS', 'default')}")
for kind in ("pageable", "pinned", "packed"):
    kb = keep if kind == "pageable" else [k.array for k in pkeep]
    kptr = [k.ctypes.data for k in kb]
    for i in range(2):
        call(kind, i, kptr[0], False)
    t0 = time.perf_counter()
    for i in range(calls):
        call(kind, i, kptr[0], False)
    dt_b = (time.perf_counter() - t0) / calls
    for i in range(2):  # slot 1 is allocated on first use
        tk = [call(kind, 0, kptr[0], True), call(kind, 1, kptr[1], True)]
        for x in tk:
            dcn._native.check(lib.dcn_filter_batch_wait(proc._h, x))
    t0 = time.perf_counter()
    fly = []
    t_submit = 0.0
    for i in range(calls):
        if len(fly) == 2:
            dcn._native.check(lib.dcn_filter_batch_wait(proc._h, fly.pop(0)))
        ts = time.perf_counter()
        fly.append(call(kind, i, kptr[i % 2], True))
        t_submit += time.perf_counter() - ts
    for x in fly:
        dcn._native.check(lib.dcn_filter_batch_wait(proc._h, x))
    dt_p = (time.perf_counter() - t0) / calls
    k_ = kb[(calls - 1) % 2].copy()
    if ref is None:
        ref = k_
    assert (k_ == ref).all() or calls % 2 == 0
    print(f"  {kind:9s} blocking {dt_b * 1e3:6.2f} ms/call = {n_bases / dt_b / 1e9:6.1f} Gbp/s | two in flight {dt_p * 1e3:6.2f} ms/call = "
          f"{n_bases / dt_p / 1e9:6.1f} Gbp/s (submit itself {t_submit / calls * 1e3:.2f} ms/call)")
