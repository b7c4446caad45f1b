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


class Comm:
    """RCCL communicator of the C ABI (dcn_comm_*): what a host that is not Python uses for the path's one collective.
    `exchange(id_bytes_or_None) -> id_bytes` hands rank 0's 128-byte id to every rank (torch.distributed.broadcast_object_list,
    a file, MPI ...: the library does not care).  Creating it is a collective call."""

    def __init__(self, world_size, rank, device, exchange):
        import ctypes as C

        from . import _native as N
        self._N, self._h = N, None
        N.check(N.lib().dcn_comm_available())
        ident = (C.c_uint8 * 128)()
        if rank == 0:
            N.check(N.lib().dcn_comm_unique_id(ident))
        raw = exchange(bytes(ident) if rank == 0 else None)
        if len(raw) != 128:
            raise ValueError("the communicator id is 128 bytes")
        ident = (C.c_uint8 * 128).from_buffer_copy(raw)
        h = C.c_void_p()
        N.check(N.lib().dcn_comm_create(ident, world_size, rank, device, C.byref(h)))
        self._h = h

    def allreduce_counters(self, processors):
        """sum of the six counters over these FilterProcessors and over all ranks -> dict"""
        import ctypes as C
        N = self._N
        ctxs = (C.c_void_p * max(len(processors), 1))(*[p._h for p in processors])
        out = (C.c_uint64 * len(STAT_NAMES))()
        N.check(N.lib().dcn_stats_allreduce_rccl(self._h, ctxs, len(processors), out))
        return dict(zip(STAT_NAMES, (int(x) for x in out)))

    def close(self):
        if self._h is not None:
            self._N.lib().dcn_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


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


# ---- host threads next to the rank's GPU ------------------------------------------------------------------------------
def parse_cpulist(text):
    """'0-3,8,10-11' -> [0, 1, 2, 3, 8, 10, 11]"""
    out = []
    for part in text.strip().split(","):
        if not part:
            continue
        a, _, b = part.partition("-")
        out.extend(range(int(a), int(b or a) + 1))
    return out


def format_cpulist(cpus):
    """[0, 1, 2, 3, 8, 10, 11] -> '0-3,8,10-11'"""
    out, cpus = [], sorted(cpus)
    i = 0
    while i < len(cpus):
        j = i
        while j + 1 < len(cpus) and cpus[j + 1] == cpus[j] + 1:
            j += 1
        out.append(str(cpus[i]) if i == j else f"{cpus[i]}-{cpus[j]}")
        i = j + 1
    return ",".join(out)


def gpu_numa_topology(sysfs="/sys"):
    """[(numa_node, [cpus])] per GPU in HIP ordinal order, read from sysfs only (no GPU call: the affinity has to be
    set before the runtime and the library's host threads exist).  KFD lists its nodes in the order HIP enumerates
    them; the ones with SIMDs are GPUs, and `drm_render_minor` names each one's DRM device, whose `numa_node` /
    `local_cpulist` say which cores sit on its socket."""
    import glob
    import os
    import re
    gpus = []
    nodes = sorted(glob.glob(os.path.join(sysfs, "class/kfd/kfd/topology/nodes/*")), key=lambda p: int(os.path.basename(p)))
    for nd in nodes:
        try:
            props = dict(line.split(None, 1) for line in open(os.path.join(nd, "properties")).read().splitlines() if " " in line)
        except OSError:
            continue
        if int(props.get("simd_count", "0")) == 0:
            continue
        dev = os.path.join(sysfs, "class/drm", "renderD%d" % int(props.get("drm_render_minor", "-1")), "device")
        try:
            numa = int(open(os.path.join(dev, "numa_node")).read())
            cpus = parse_cpulist(open(os.path.join(dev, "local_cpulist")).read())
        except (OSError, ValueError):
            numa, cpus = -1, []
        gpus.append((numa, cpus))
    vis = os.environ.get("HIP_VISIBLE_DEVICES") or os.environ.get("ROCR_VISIBLE_DEVICES")
    if vis and re.fullmatch(r"[0-9]+(,[0-9]+)*", vis):  # ordinals are positions in the visible list
        gpus = [gpus[int(i)] for i in vis.split(",") if int(i) < len(gpus)]
    return gpus


def cpus_for_rank(local_rank, local_world, allowed, topology, gpu_of_rank=None):
    """CPUs this rank's host threads should run on: the allowed CPUs local to its GPU, shared out evenly between the
    ranks whose GPUs sit on the same NUMA node (counterpart of the reference's one reader feeding N workers,
    src/local_filter.rs:696-709: here every rank has its own packer threads, and 8 of them share one host).
    gpu_of_rank: GPU ordinal of every local rank (default: rank r drives GPU r; a rehearsal puts all ranks on GPU 0).
    Returns (cpus, note); cpus is None when there is nothing sensible to bind to."""
    allowed = sorted(allowed)
    gpu_of_rank = list(range(local_world)) if gpu_of_rank is None else list(gpu_of_rank)
    gpu = gpu_of_rank[local_rank] if local_rank < len(gpu_of_rank) else local_rank
    if gpu >= len(topology):
        return None, f"GPU {gpu} not found in the KFD topology ({len(topology)} GPUs listed)"
    numa, local = topology[gpu]
    near = [c for c in local if c in set(allowed)]
    if not near:
        return None, f"no allowed CPU is local to GPU {gpu} (NUMA node {numa})"
    peers = [r for r in range(len(gpu_of_rank)) if gpu_of_rank[r] < len(topology) and topology[gpu_of_rank[r]][0] == numa]
    i, n = peers.index(local_rank), len(peers)
    share = near[len(near) * i // n:len(near) * (i + 1) // n]
    if not share:
        return None, f"fewer allowed CPUs ({len(near)}) than ranks ({n}) on NUMA node {numa}"
    return share, f"NUMA node {numa}: {len(share)} of its {len(near)} allowed CPUs (rank {i + 1} of {n} on that node)"


def bind_rank_to_gpu_cpus(local_rank, local_world, sysfs="/sys", gpu_of_rank=None):
    """sched_setaffinity of the calling process (threads created later inherit it).  Call before any GPU call."""
    import os
    try:
        topo = gpu_numa_topology(sysfs)
        cpus, note = cpus_for_rank(local_rank, local_world, os.sched_getaffinity(0), topo, gpu_of_rank)
        if cpus:
            os.sched_setaffinity(0, cpus)
        return {"cpus": cpus, "cpulist": format_cpulist(cpus) if cpus else None, "note": note}
    except Exception as ex:  # never fatal: an unbound rank is slower, not wrong
        return {"cpus": None, "cpulist": None, "note": f"not bound: {ex!r}"}
