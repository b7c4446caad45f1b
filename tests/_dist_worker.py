"""Worker of tests/test_distributed.py (launched with torch.distributed.run, backend gloo).

Without a GPU (argv[2] absent) the per-batch filter engine is the CPU oracle standing in for
FilterProcessor.filter_batch (same inputs, same outputs): what is under test is the N>1 host logic of
deacon-server_amd/distributed.py.  With argv[2] == "gpu" (the -m gpu test) every rank runs the REAL engine -- its
own index replica and FilterProcessor on GPU 0 of the one-GPU box -- and takes its counters from dcn_ctx_stats;
the collectives stay on gloo (RCCL refuses two ranks on one device), which is the only thing that differs from
bench.py's N>1 run."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import deacon_server_amd as dcn  # noqa: E402
from conftest import mutate, random_reads  # noqa: E402
from oracle import oracle as O  # noqa: E402


def main():
    out_path = sys.argv[1]
    on_gpu = len(sys.argv) > 2 and sys.argv[2] == "gpu"
    if on_gpu:
        import torch  # noqa: F401  (its HIP runtime first, as everywhere)
    dist.init_process_group(backend="gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    rng = np.random.default_rng(123)  # same data on every rank
    genome = random_reads(rng, 1, 30_000, 30_000)[0]
    idx = O.Index.build([genome])
    reads = []
    for i in range(3001):  # odd: the last pair is a singleton unit
        ln = int(rng.integers(60, 160))
        if i % 3:
            s = int(rng.integers(0, len(genome) - ln))
            reads.append(mutate(rng, genome[s:s + ln], 0.01))
        else:
            reads.append(random_reads(rng, 1, ln, ln)[0])
    uid = (np.arange(len(reads)) // 2).astype(np.uint32)
    n_units = int(uid[-1]) + 1
    batch_units = 100
    batches = dcn.distributed.split_units(n_units, batch_units)
    mine = dcn.distributed.batches_of_rank(len(batches), rank, world)
    stats = {n: 0 for n in dcn._native.STAT_NAMES}
    local = {}
    proc = None
    if on_gpu:
        gidx = dcn.Index.from_keys(idx.keys(), 31, 15, device=0)  # this rank's replica
        proc = dcn.FilterProcessor(gidx, deplete=True, max_batch_bases=1 << 20, max_batch_reads=1 << 12)
    for seq in mine:
        a, b = batches[seq]
        sel = [r for r in range(len(reads)) if a <= uid[r] < b]
        sub = [reads[r] for r in sel]
        bases, offsets = O.concat_reads(sub)
        if on_gpu:
            keep, hits, total = proc.filter_batch(bases, offsets, (uid[sel] - a).astype(np.uint32))
            local[seq] = keep
            continue
        keep, hits, total = O.filter_batch(idx, bases, offsets, (uid[sel] - a).astype(np.uint32), deplete=True)
        local[seq] = keep
        for u in range(b - a):
            rs = [r for r in sel if uid[r] - a == u]
            bp = sum(len(reads[r]) for r in rs)
            stats["total_seqs"] += len(rs)
            stats["total_bp"] += bp
            if keep[u]:
                stats["output_bp"] += bp
                stats["output_seq_counter"] += len(rs)
            else:
                stats["filtered_seqs"] += len(rs)
                stats["filtered_bp"] += bp
    if on_gpu:
        stats = proc.stats()  # the engine's own six counters
    total_stats = dcn.distributed.allreduce_counters(stats)
    keep_all = dcn.distributed.gather_keep_in_order(local, n_units, batch_units)
    if rank == 0:
        bases, offsets = O.concat_reads(reads)
        want_keep, _, _ = O.filter_batch(idx, bases, offsets, uid, deplete=True)
        lens = np.array([len(r) for r in reads])
        ulen = np.bincount(uid, weights=lens).astype(np.int64)
        ucnt = np.bincount(uid)
        want = {"total_seqs": int(len(reads)), "total_bp": int(lens.sum()),
                "output_bp": int(ulen[want_keep].sum()), "filtered_bp": int(ulen[~want_keep].sum()),
                "filtered_seqs": int(ucnt[~want_keep].sum()), "output_seq_counter": int(ucnt[want_keep].sum())}
        json.dump({"world": world, "stats": total_stats, "want": want,
                   "keep_equal": bool((keep_all == want_keep).all()), "n_batches": len(batches),
                   "mine": mine}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
