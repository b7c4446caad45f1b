"""N>1 host logic: one process per GPU, index replicated, units sharded by batch, one collective at the end.

The path has no data-path exchange (units are independent given the immutable index); the only
collectives are the sum of the six ProcessingStats counters (src/local_filter.rs:388-396 merges them under a mutex;
across ranks it is one all-reduce: RCCL with backend "nccl", gloo on CPU) and, when a caller wants the decisions of
the whole job in input order, a gather of the per-batch keep bitmaps by batch sequence number."""
import numpy as np

from ._native import STAT_NAMES


def batches_of_rank(n_batches, rank, world_size):
    """Batch sequence numbers handled by `rank`: contiguous batches dealt round-robin.  A batch holds whole units
    (a pair is never split), so any batch->rank map keeps decisions identical to the single-process run."""
    return list(range(rank, n_batches, world_size))


def split_units(n_units, batch_units):
    """[(first_unit, last_unit_exclusive)] of consecutive batches of at most batch_units units."""
    return [(a, min(n_units, a + batch_units)) for a in range(0, n_units, batch_units)]


def _collective_device(device, group):
    """Device a collective's tensor must live on: an RCCL ("nccl") group has no CPU backend, so host arrays are
    moved to this rank's GPU first; gloo takes CPU tensors."""
    import torch
    import torch.distributed as dist
    if device is not None:
        return torch.device(device)
    if dist.is_available() and dist.is_initialized() and "nccl" in str(dist.get_backend(group)).lower():
        return torch.device("cuda", torch.cuda.current_device())
    return torch.device("cpu")


def allreduce_counters(stats, device=None, group=None):
    """Sum the six counters over all ranks; `stats` is FilterProcessor.stats().  Returns a dict."""
    import torch
    import torch.distributed as dist
    t = torch.tensor([int(stats[n]) for n in STAT_NAMES], dtype=torch.int64, device=_collective_device(device, group))
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)
    return dict(zip(STAT_NAMES, (int(x) for x in t.cpu().tolist())))


def gather_keep_in_order(local_batches, n_units, batch_units, group=None, device=None):
    """local_batches: {batch_seq: bool array} of this rank -> keep bitmap of the whole job in unit order
    (identical on every rank).  Host-side ordered merge by batch sequence number."""
    import torch
    import torch.distributed as dist
    keep = np.zeros(n_units, np.uint8)
    for (seq, k), (a, b) in ((item, split_units(n_units, batch_units)[item[0]]) for item in local_batches.items()):
        keep[a:b] = np.asarray(k, dtype=np.uint8)
    if dist.is_available() and dist.is_initialized() and dist.get_world_size(group) > 1:
        t = torch.from_numpy(keep).to(_collective_device(device, group))
        dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)  # every unit is written by exactly one rank
        keep = t.cpu().numpy()
    return keep.astype(bool)
