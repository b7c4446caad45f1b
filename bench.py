#!/usr/bin/env python3
"""Headline benchmark: Mbp/s filtered (k=31, w=15) against a panhuman-1-sized index on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1 without a launcher: this file starts `python -m torch.distributed.run --nnodes=1 --nproc-per-node N ...` of itself
   as a CHILD process -- before anything here has touched the GPU -- relays the child's output and exits with its code;
   under torch.distributed.run (RANK / WORLD_SIZE set) it is one rank of that job, as the task contract launches it.)

One "step" = one pass of the hot path (pack -> plan -> scan/hash/probe/distinct -> finish) over one batch of
synthetic 150 bp reads that is already resident in HBM as ASCII + offsets (BASELINE.json configs[1]).  Three distinct
batches are rotated, so no step re-probes the table lines the previous step left in L2 / Infinity Cache.  The index
is a device table of 409,913,780 synthetic u64 keys (panhuman-1's size): the minimizers of a synthetic "host" genome
plus pseudo-random keys mix64(1..n); half of the reads are drawn from the host genome (0.5 % substitutions, 0.1 % N),
half are random.  Weak scaling: every rank holds a full index replica and filters its own batches; the only
collective is the all-reduce of the six summary counters (RCCL) at the end of the timed region.

Output.  The LAST stdout line (rank 0) is the contract's JSON, kept small (compact_line(): < 4 KB, never above 8 KB):
value = whole-job Mbp/s of that workload, plus
  roofline      dominant kernel (scan): algorithmic HBM bytes / HIP-event time on the kernel's own stream; `traffic` = HBM bytes
                per launch from PMC counters, measured after the timed region by child runs of this file under rocprofv3 --pmc
  cpu_baseline  the CPU oracle (oracle/, "port") on a bounded sample of the same reads, all host cores
and one scalar per extra leg measured at N = 1 after the timed region (none of it enters `value`):
  legs.{long,paired,union950m,...}           BASELINE configs[2], [3], [4]-sized table: Mbp/s + the oracle check
  host_path.{pageable,pinned,packed}         the PCIe-inclusive rate of dcn_filter_batch* from host memory
  cli.{search50,deplete95,deplete95_gz,paired} `deacon-hip filter` file to file (deplete95_gz: from one gzip stream)
Everything else that is measured (per-stage times, repetitions, counters, samples checked) goes to bench_detail.json next
to this file (--detail PATH) and, as one line, to stderr.
"""
import argparse
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))


def _flag_value(argv, name, default):
    """value of `--name V` / `--name=V` in argv (the launcher parent reads --gpus without building the whole parser)"""
    for i, a in enumerate(argv):
        if a == name and i + 1 < len(argv):
            return argv[i + 1]
        if a.startswith(name + "="):
            return a.split("=", 1)[1]
    return default


def self_launch_if_needed(argv):
    """`python bench.py --gpus N` (N > 1) outside torch.distributed.run: start the N-rank job as a child process, relay
    its stdout (rank 0's final JSON line is the last thing it prints) and return its exit code.  Runs before torch or the
    library is imported, so this process has made no GPU call; it never replaces itself (no exec), it waits for the child.
    The reference's counterpart is one command starting all of its workers (src/local_filter.rs:696-709)."""
    try:
        n = int(_flag_value(argv, "--gpus", "1"))
    except ValueError:
        return None
    if n <= 1 or "WORLD_SIZE" in os.environ or "RANK" in os.environ:
        return None
    import socket
    with socket.socket() as s_:
        s_.bind(("127.0.0.1", 0))
        port = s_.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__), *argv]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC only on this pool: RCCL needs it
    env.setdefault("OMP_NUM_THREADS", "1")             # (torchrun sets it anyway, and says so on stderr)
    print(f"[bench] --gpus {n} without a launcher: starting {' '.join(cmd[1:8])} ... as a child process", file=sys.stderr, flush=True)
    child = subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, text=True, bufsize=1)
    try:
        for line in child.stdout:
            sys.stdout.write(line)
            sys.stdout.flush()
        return child.wait()
    except BaseException:
        child.kill()
        child.wait()
        raise


if __name__ == "__main__":
    _rc = self_launch_if_needed(sys.argv[1:])
    if _rc is not None:
        sys.exit(_rc)

import numpy as np  # noqa: E402
import torch  # noqa: E402  first: libdeacon_hip.so must bind to the HIP runtime torch already loaded
import torch.distributed as dist  # noqa: E402

sys.path.insert(0, ROOT)
import deacon_server_amd as dcn  # noqa: E402

K, W = 31, 15
READ_LEN = 150
PANHUMAN_KEYS = 409_913_780  # README.md:52 of the reference
UNION_KEYS = 950_000_000     # panmouse-1a u panhuman-1 (BASELINE.json configs[4])
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
PCIE_PEAK_GBS = 64.0         # PCIe Gen5 x16, one direction
SCATTER_CEILING = 46e9  # random 16-byte reads/s of a 2^31-slot table on MI355X, measured (profiles/r01_probe_patterns_17GB.txt)
ROTATE = 3
T0 = time.time()
# Extras that follow the headline at N = 1 by default (the whole run stays near a minute), and every one there is
DEFAULT_EXTRAS = "long,paired,host_path,pmc,union950m,cli"
ALL_EXTRAS = "long,paired,host_path,cli,pmc,host1g,host95,union950m,union950m_paired,pmc_legs"


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print(f"[bench {time.time() - T0:6.1f}s]", *a, file=sys.stderr, flush=True)


# ---- synthetic index keys: host-genome minimizers + mix64(1..n) ---------------------------------------------------------
_M1, _M2 = 0xBF58476D1CE4E5B9, 0x94D049BB133111EB


def _signed(x):
    return x - (1 << 64) if x >= (1 << 63) else x


def mix64_device(first, count, device):
    """splitmix64's finalizer of first .. first+count-1 (a bijection on u64) as int64 bit patterns, on the GPU"""
    def lsr(x, s):
        return (x >> s) & ((1 << (64 - s)) - 1)
    z = torch.arange(first, first + count, dtype=torch.int64, device=device)
    z = (z ^ lsr(z, 30)) * _signed(_M1)
    z = (z ^ lsr(z, 27)) * _signed(_M2)
    return z ^ lsr(z, 31)


def unmix64(h):
    """inverse of mix64 on a numpy uint64 array: membership of a hash in {mix64(i) : 1 <= i <= n} is unmix64(h) in 1..n"""
    h = np.asarray(h, dtype=np.uint64)

    def inv_xs(y, s):
        x = y.copy()
        for _ in range(64 // s + 1):
            x = y ^ (x >> np.uint64(s))
        return x
    with np.errstate(over="ignore"):
        z = inv_xs(h, 31) * np.uint64(pow(_M2, -1, 1 << 64))
        z = inv_xs(z, 27) * np.uint64(pow(_M1, -1, 1 << 64))
        return inv_xs(z, 30)


def make_host_genome(n, seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    alpha = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    return alpha[torch.randint(0, 4, (n,), generator=g, device=device)]


def host_minimizer_keys(genome_dev, device_index):
    """Distinct minimizer hashes of the host genome, computed by the product path itself (dcn_index_build + key
    export): for ACGT-only sequence the index-side and filter-side rules coincide."""
    idx = dcn.Index.build([genome_dev.cpu().numpy()], K, W, device=device_index)
    keys = idx.keys()
    idx.close()
    return np.sort(keys)


def build_index(genome_dev, n_keys, local_rank, world=1, rank=0):
    """Device table of the synthetic index.  The key array (3.3 GB for panhuman-1's size) is generated ONCE per node:
    at world > 1 rank 0 writes it to tmpfs and the other ranks map it (one copy in the page cache instead of one
    generation and one 3.3 GB array per rank); every rank then builds its own replica from it."""
    t0 = time.time()
    shared = None
    if world > 1:
        shared = os.path.join("/dev/shm" if os.path.isdir("/dev/shm") else "/tmp",
                              f"dcn_bench_keys_{os.environ.get('MASTER_PORT', '0')}_{n_keys}.u64")
    host_keys, n_rand = None, 0
    if rank == 0:
        host_keys = host_minimizer_keys(genome_dev, local_rank)
        n_rand = max(0, n_keys - len(host_keys))
        n_all = len(host_keys) + n_rand
        keys = np.lib.format.open_memmap(shared, mode="w+", dtype=np.uint64, shape=(n_all,)) if shared else np.empty(n_all, np.uint64)
        keys[:len(host_keys)] = host_keys
        step = 1 << 27
        for a in range(0, n_rand, step):
            m = min(step, n_rand - a)
            keys[len(host_keys) + a:len(host_keys) + a + m] = mix64_device(1 + a, m, genome_dev.device).cpu().numpy().view(np.uint64)
        if shared:
            keys.flush()
    if world > 1:
        dist.barrier()
        if rank != 0:
            keys = np.load(shared, mmap_mode="r")
    t1 = time.time()
    index = dcn.Index.from_keys(keys, K, W, device=local_rank)
    build_s = time.time() - t1
    if world > 1:
        dist.barrier()
        if rank == 0:
            os.unlink(shared)
    if rank == 0:
        log(f"index: {len(host_keys):,} host-genome keys + {n_rand:,} mix64 keys generated in {t1 - t0:.1f} s"
            f"{' (once for the node, shared through ' + os.path.dirname(shared) + ')' if shared else ''}; device table "
            f"of {index.n_keys:,} distinct keys built in {build_s:.1f} s")
    return index, keys, host_keys, n_rand, build_s


# ---- synthetic reads, generated on the device ---------------------------------------------------------------------
def make_reads(genome_dev, n_reads, seed, device, host_frac=0.5, sub=0.005, p_n=0.001):
    """n_reads x READ_LEN ASCII on the device: host-derived (with substitutions / N) or uniform random."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    alpha = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    out = torch.empty((n_reads, READ_LEN), dtype=torch.uint8, device=device)
    ar = torch.arange(READ_LEN, device=device)
    chunk = 1 << 19
    for a in range(0, n_reads, chunk):
        m = min(chunk, n_reads - a)
        is_host = torch.rand(m, generator=g, device=device) < host_frac
        starts = torch.randint(0, genome_dev.numel() - READ_LEN, (m,), generator=g, device=device)
        host = genome_dev[starts[:, None] + ar[None, :]]
        rnd = alpha[torch.randint(0, 4, (m, READ_LEN), generator=g, device=device)]
        mut = torch.rand((m, READ_LEN), generator=g, device=device) < sub
        host = torch.where(mut, rnd, host)
        nmask = torch.rand((m, READ_LEN), generator=g, device=device) < p_n
        host = torch.where(nmask, torch.full_like(host, ord("N")), host)
        out[a:a + m] = torch.where(is_host[:, None], host, rnd)
    return out.reshape(-1)


def _revcomp_ascii(t):
    """reverse complement of a (n, L) uint8 ACGT/N tensor"""
    comp = torch.arange(256, dtype=torch.uint8, device=t.device)
    for a, b in (("A", "T"), ("C", "G"), ("G", "C"), ("T", "A")):
        comp[ord(a)] = ord(b)
    return comp[t.long()].flip(1)


def make_pairs(genome_dev, n_pairs, seed, device, host_frac=0.5, sub=0.005, p_n=0.001):
    """2 x READ_LEN pairs (mate 2 reverse-complemented, insert 350 +- 50): rows 2i, 2i+1 are the mates of pair i."""
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    alpha = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    out = torch.empty((n_pairs, 2, READ_LEN), dtype=torch.uint8, device=device)
    ar = torch.arange(READ_LEN, device=device)
    chunk = 1 << 18
    for a in range(0, n_pairs, chunk):
        m = min(chunk, n_pairs - a)
        is_host = torch.rand(m, generator=g, device=device) < host_frac
        ins = (350 + 50 * torch.randn(m, generator=g, device=device)).clamp(READ_LEN, 600).long()
        starts = torch.randint(0, genome_dev.numel() - 700, (m,), generator=g, device=device)
        m1 = genome_dev[starts[:, None] + ar[None, :]]
        m2 = _revcomp_ascii(genome_dev[(starts + ins - READ_LEN)[:, None] + ar[None, :]])
        both = torch.stack([m1, m2], 1)
        rnd = alpha[torch.randint(0, 4, (m, 2, READ_LEN), generator=g, device=device)]
        mut = torch.rand((m, 2, READ_LEN), generator=g, device=device) < sub
        both = torch.where(mut, rnd, both)
        nmask = torch.rand((m, 2, READ_LEN), generator=g, device=device) < p_n
        both = torch.where(nmask, torch.full_like(both, ord("N")), both)
        out[a:a + m] = torch.where(is_host[:, None, None], both, rnd)
    return out.reshape(-1)


def make_long_reads(genome_dev, total_bases, seed, device, host_frac=0.5, sub=0.05):
    """ONT-style reads: LogNormal(mu=8.8903, sigma=0.8) lengths (mean 10 kbp) clamped to [200, 500000]; host-derived
    reads carry 5 % substitutions.  Returns (bases u8[], offsets int64[n+1])."""
    rng = np.random.default_rng(seed)
    lens = []
    tot = 0
    while tot < total_bases:
        ln = int(min(500_000, max(200, rng.lognormal(8.8903, 0.8))))
        ln = min(ln, genome_dev.numel() - 1)
        lens.append(ln)
        tot += ln
    lens = np.array(lens, dtype=np.int64)
    offsets = np.zeros(len(lens) + 1, np.int64)
    np.cumsum(lens, out=offsets[1:])
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    alpha = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    out = torch.empty(int(offsets[-1]), dtype=torch.uint8, device=device)
    is_host = rng.random(len(lens)) < host_frac
    starts = (rng.random(len(lens)) * (genome_dev.numel() - lens)).astype(np.int64)
    # chunks of reads holding <= 32 M bases: per-base source index = start[read] + position in read
    a = 0
    while a < len(lens):
        b = a
        while b < len(lens) and offsets[b + 1] - offsets[a] <= (1 << 25):
            b += 1
        b = max(b, a + 1)
        n = int(offsets[b] - offsets[a])
        ln_t = torch.from_numpy(lens[a:b]).to(device)
        rid = torch.repeat_interleave(torch.arange(b - a, device=device), ln_t)
        pos = torch.arange(n, device=device) - torch.from_numpy(offsets[a:b] - offsets[a]).to(device)[rid]
        src = genome_dev[torch.from_numpy(starts[a:b]).to(device)[rid] + pos]
        rnd = alpha[torch.randint(0, 4, (n,), generator=g, device=device)]
        mut = torch.rand(n, generator=g, device=device) < sub
        src = torch.where(mut, rnd, src)
        host = torch.from_numpy(is_host[a:b]).to(device)[rid]
        out[int(offsets[a]):int(offsets[b])] = torch.where(host, src, rnd)
        a = b
    return out, torch.from_numpy(offsets).to(device)


def make_mixed_reads(genome_dev, total_bases, seed, device):
    """BASELINE configs[4]'s stream shape: half of the bases in ONT-style long reads (as configs[2]),
    half in 150 bp reads (as configs[1]), interleaved in one stream -- every long read is followed by as many short reads as
    it has bases / 150.  Returns (bases u8[], offsets int64[n+1])."""
    lb, lo = make_long_reads(genome_dev, total_bases // 2, seed, device)
    lo_np = lo.cpu().numpy()
    llen = np.diff(lo_np)
    n_short_after = np.maximum(1, np.rint(llen / READ_LEN).astype(np.int64))
    n_short = int(n_short_after.sum())
    sb = make_reads(genome_dev, n_short, seed + 1, device)
    # lengths in stream order: long_0, short x c_0, long_1, short x c_1, ...
    n_long = len(llen)
    pos_long = np.arange(n_long) + np.concatenate([[0], np.cumsum(n_short_after)[:-1]])  # index of long read i in the stream
    n_all = n_long + n_short
    lens = np.full(n_all, READ_LEN, np.int64)
    lens[pos_long] = llen
    offsets = np.zeros(n_all + 1, np.int64)
    np.cumsum(lens, out=offsets[1:])
    is_long = np.zeros(n_all, bool)
    is_long[pos_long] = True
    out = torch.empty(int(offsets[-1]), dtype=torch.uint8, device=device)
    # long reads: out[dest_start_i + j] = lb[lo_i + j]; short reads likewise from the (n_short, 150) block -- copied in
    # slices of <= 64 M bases of source
    def scatter(src, src_off, dst_off):
        a = 0
        while a < len(src_off) - 1:
            b = int(np.searchsorted(src_off, src_off[a] + (1 << 26), side="right")) - 1
            b = min(max(b, a + 1), len(src_off) - 1)
            ln = torch.from_numpy(np.diff(src_off[a:b + 1])).to(device)
            shift = torch.from_numpy(dst_off[a:b] - src_off[a:b]).to(device)
            idx = torch.arange(int(src_off[a]), int(src_off[b]), device=device)
            out[idx + torch.repeat_interleave(shift, ln)] = src[int(src_off[a]):int(src_off[b])]
            a = b
    scatter(lb, lo_np, offsets[:-1][is_long])
    scatter(sb, np.arange(n_short + 1, dtype=np.int64) * READ_LEN, offsets[:-1][~is_long])
    return out, torch.from_numpy(offsets).to(device)


class Batch:
    """one batch resident in HBM: ASCII + offsets (+ unit ids) and its result arrays"""

    def __init__(self, d_bases, d_offsets, d_unit_id=None):
        self.d_bases, self.d_offsets, self.d_unit_id = d_bases, d_offsets, d_unit_id
        self.n_reads = d_offsets.numel() - 1
        self.n_units = self.n_reads if d_unit_id is None else self.n_reads // 2
        self.n_bases = int(d_bases.numel())
        dev = d_bases.device
        self.d_keep = torch.zeros(self.n_units, dtype=torch.uint8, device=dev)
        self.d_hits = torch.zeros(self.n_units, dtype=torch.int32, device=dev)
        self.d_total = torch.zeros(self.n_units, dtype=torch.int32, device=dev)
        self.d_keep2 = torch.zeros(self.n_units, dtype=torch.uint8, device=dev)


def make_batches(kind, genome_dev, reads, seed, device, rotate=ROTATE, host_frac=0.5):
    out = []
    for i in range(rotate):
        if kind == "short":
            b = make_reads(genome_dev, reads, seed + 100 * i, device, host_frac=host_frac)
            out.append(Batch(b, torch.arange(reads + 1, dtype=torch.int64, device=device) * READ_LEN))
        elif kind == "paired":
            n = reads // 2 * 2
            b = make_pairs(genome_dev, n // 2, seed + 100 * i, device)
            out.append(Batch(b, torch.arange(n + 1, dtype=torch.int64, device=device) * READ_LEN,
                             (torch.arange(n, dtype=torch.int32, device=device) // 2).contiguous()))
        elif kind == "long":
            b, o = make_long_reads(genome_dev, reads * READ_LEN, seed + 100 * i, device)
            out.append(Batch(b, o))
        elif kind == "mixed":
            b, o = make_mixed_reads(genome_dev, reads * READ_LEN, seed + 100 * i, device)
            out.append(Batch(b, o))
        else:
            raise ValueError(kind)
    return out


def host_cores():
    """Cores this process may really use: the affinity mask capped by the cgroup CPU quota (the GPU box gives a
    1-GPU job 16 of its 256 hardware threads)."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return n


# ---- the device-resident measurement ---------------------------------------------------------------------------------
def run_device_workload(index, batches, params, steps, warmup, world, device, reserve_long=False, coll_device=None):
    """K steps over the rotating batches, counting mode (keep + distinct hits + totals), then the same K steps asking
    for decisions only.  Returns (result dict, counters of the timed region)."""
    coll_device = device if coll_device is None else coll_device
    max_bases = max(b.n_bases for b in batches)
    max_reads = max(b.n_reads for b in batches)
    proc = dcn.FilterProcessor(index, abs_threshold=params["abs"], rel_threshold=params["rel"], deplete=params["deplete"],
                               max_batch_bases=max_bases, max_batch_reads=max_reads)
    if reserve_long:
        proc.reserve_records(max_bases // 6)  # ~1 hit record per 16 bp when half the reads are host-derived

    def step(i, counts=True):
        b = batches[i % len(batches)]
        proc.filter_batch_device(b.d_bases.data_ptr(), b.d_offsets.data_ptr(), b.n_reads, b.n_bases,
                                 (b.d_keep if counts else b.d_keep2).data_ptr(),
                                 b.d_hits.data_ptr() if counts else None, b.d_total.data_ptr() if counts else None,
                                 d_unit_id=b.d_unit_id.data_ptr() if b.d_unit_id is not None else None, n_units=b.n_units)

    def stage_breakdown(counts):
        """per-stage device times from a few UNTIMED steps with events around every stage (six marker packets per
        step cost ~2.5 % of it: the timed region only carries the two around the scan kernel)"""
        proc.set_profiling(1)
        for i in range(3):
            step(i, counts)
        proc.synchronize()
        ms, nb = proc.profile()
        proc.set_profiling(False)
        return {k_: v / max(nb, 1) for k_, v in ms.items()}

    def timed(counts):
        for i in range(warmup):
            step(i, counts)
        proc.synchronize()
        stage = stage_breakdown(counts)
        proc.reset_stats()
        proc.set_profiling(2)  # HIP events around the scan kernel of every timed step, on the stream it runs on
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for i in range(steps):
            step(i, counts)
        proc.synchronize()
        local = proc.stats()
        # RCCL all-reduce of the six counters: the path's only collective
        counters = dcn.distributed.allreduce_counters(local, device=coll_device)
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        elapsed = time.perf_counter() - t0
        t = torch.tensor([elapsed], dtype=torch.float64, device=coll_device)
        if world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
        ms, nb = proc.profile()
        proc.set_profiling(False)
        stage["scan"] = ms["scan"] / max(nb, 1)  # the live measurement over the timed region replaces the sampled one
        return float(t.item()), counters, stage

    elapsed, counters, stage = timed(True)
    elapsed2, _, stage2 = timed(False)
    proc.close()
    bases_done = sum(batches[i % len(batches)].n_bases for i in range(steps))  # this rank
    total_bp = counters["total_bp"]
    n_min = [int(b.d_total.sum(dtype=torch.int64).item()) for b in batches[:min(len(batches), steps)]]
    used = [batches[i % len(batches)] for i in range(steps)]
    algo = float(np.mean([0.375 * b.n_bases for b in used])) + 8.0 * float(np.mean([n_min[i % len(batches)] for i in range(steps)]))
    scan_ms = stage["scan"]
    achieved = algo / (scan_ms * 1e-3) / 1e9
    mins = float(np.mean([n_min[i % len(batches)] for i in range(steps)]))
    res = {
        "value": total_bp / elapsed / 1e6, "unit": "Mbp/s", "ms_per_step": elapsed / steps * 1e3,
        "bases_per_batch": int(np.mean([b.n_bases for b in batches])), "reads_per_batch": int(np.mean([b.n_reads for b in batches])),
        "batches_rotated": len(batches),
        "roofline": {
            "bound": "hbm", "kernel": "scan_kernel<15> (scan+hash+probe+distinct)",
            "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
            "algorithmic_bytes_per_launch": algo, "avg_launch_ms": scan_ms, "minimizers_per_launch": int(mins),
            # second ceiling, reported beside the contract's: every minimizer is one scattered 16-byte request, and the
            # chip serves ~46 G of those per second from a 17-34 GB table however they are issued (profiles/r01_probe_patterns_*)
            "scattered_probes_per_s": mins / (scan_ms * 1e-3), "scattered_ceiling_per_s": SCATTER_CEILING,
            "frac_of_scattered_ceiling": mins / (scan_ms * 1e-3) / SCATTER_CEILING,
        },
        "stage_ms_per_launch": stage,
        "kept_fraction": float(np.mean([float(b.d_keep.float().mean().item()) for b in batches[:min(len(batches), steps)]])),
        # not the headline: same batches, same K steps, caller passes no hits/total arrays (what the CLI does outside
        # --debug); reads whose decision is fixed after abs_threshold distinct hits are not probed further
        "decisions_only": {"value": bases_done * world / elapsed2 / 1e6, "unit": "Mbp/s", "ms_per_step": elapsed2 / steps * 1e3,
                           "scan_ms_per_launch": stage2["scan"], "stage_ms_per_launch": stage2,
                           "decisions_identical_to_counting_mode": all(bool(torch.equal(b.d_keep, b.d_keep2)) for b in batches[:min(len(batches), steps)])},
    }
    return res, counters, elapsed, bases_done


def c_abi_rccl_check(index, batch, params, world, device, coll_device, timeout_s=90.0):
    """The LAST thing the ranks do together at N > 1, after everything the line needs has been measured: the six counters of a
    small run reduced by the C ABI's own RCCL communicator (dcn_comm_* / dcn_stats_allreduce_rccl: what a host that is not
    Python would call) and compared with torch.distributed's all-reduce of the same counters.  Every rank first says whether it
    can load RCCL at all (a torch collective), so that no rank enters the communicator's collective set-up alone; the id
    travels by broadcast_object_list.  The C calls run on a helper thread that is given `timeout_s`: a set-up that never
    returns costs the line one `timed_out` entry, not the run -- nothing after this depends on any communicator (main() ends
    the process without further collectives when that happened).  None when it was not tried."""
    if os.environ.get("DCN_BENCH_NO_C_ABI_RCCL") or str(coll_device).startswith("cpu"):
        return None
    import threading
    t0 = time.time()
    try:
        ok = torch.tensor([1 if dcn._native.lib().dcn_comm_available() == 0 else 0], dtype=torch.int64, device=coll_device)
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if int(ok.item()) == 0:
            return {"tried": False, "why": "librccl could not be loaded by the library on some rank"}
        n_reads = min(batch.n_reads, 100_000)
        n_bases = int(batch.d_offsets[n_reads].item())
        proc = dcn.FilterProcessor(index, abs_threshold=params["abs"], rel_threshold=params["rel"], deplete=params["deplete"],
                                   max_batch_bases=n_bases, max_batch_reads=n_reads)
        keep = torch.zeros(n_reads, dtype=torch.uint8, device=device)
        proc.filter_batch_device(batch.d_bases.data_ptr(), batch.d_offsets.data_ptr(), n_reads, n_bases, keep.data_ptr(), None, None,
                                 d_unit_id=None, n_units=n_reads)
        proc.synchronize()
        local = proc.stats()
        want = dcn.distributed.allreduce_counters(local, device=coll_device)
        result = {}

        def exchange(raw):
            b_ = [raw]
            dist.broadcast_object_list(b_, src=0)
            return b_[0]

        def work():
            try:
                torch.cuda.set_device(device)
                comm = dcn.distributed.Comm(world, dist.get_rank(), device.index or 0, exchange)
                result["got"] = comm.allreduce_counters([proc])
                comm.close()
            except Exception as ex:  # noqa: BLE001
                result["error"] = repr(ex)

        th = threading.Thread(target=work, daemon=True)
        th.start()
        th.join(timeout_s)
        if th.is_alive():
            return {"tried": True, "timed_out": True, "seconds": time.time() - t0}
        proc.close()
        if "error" in result:
            return {"tried": True, "error": result["error"]}
        return {"tried": True, "matches_torch_all_reduce": result["got"] == want, "total_bp": int(result["got"]["total_bp"]),
                "seconds": time.time() - t0}
    except Exception as ex:  # never take the bench line with it
        log(f"C-ABI RCCL check failed: {ex!r}")
        return {"tried": True, "error": repr(ex)}


def committed_traffic(rf, workload, bases_per_batch, index_keys, host_genome):
    """HBM bytes per scan launch from the committed PMC passes of the same workload (profiles/collect_r3.sh ->
    profiles/make_traffic_json.py): counters cannot be read inside a timed run, so `roofline.traffic` is the per-launch
    figure measured for this workload shape on this tree, and says where it comes from."""
    names = {"short": ("r04_traffic.json", "r03_traffic.json", "r02_traffic.json"), "long": ("r04_traffic_long.json", "r03_traffic_long.json"),
             "mixed": ("r04_traffic_mixed.json", "r03_traffic_mixed.json")}
    for name in names.get(workload, ()):
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", name)))
            wlk = tr["workload"]
            bases = wlk.get("bases_per_batch", wlk["reads_per_batch"] * READ_LEN)
            if (wlk["workload"] == workload and abs(bases - bases_per_batch) <= 0.01 * bases_per_batch
                    and wlk["index_keys"] == index_keys and wlk.get("host_genome_bases") == host_genome):
                rf["traffic"] = tr["scan_kernel"]["hbm_bytes_per_launch"]
                rf["traffic_source"] = "profiles/" + name
                rf["traffic_rate_GBps"] = rf["traffic"] / (rf["avg_launch_ms"] * 1e-3) / 1e9
                rf["traffic_frac_of_peak"] = rf["traffic_rate_GBps"] / HBM_PEAK_GBS
                return rf["traffic"]
        except Exception:
            pass
    return None


def live_traffic(workload, reads, index_keys, host_genome):
    """HBM bytes per launch of the counting scan kernel, MEASURED in this run: two child runs of this same file under
    `rocprofv3 --pmc` (FETCH_SIZE, then WRITE_SIZE: separate passes, counters only -- no tracing domain beside them), same
    workload, three timed steps each.  Calibration as profiles/r02_traffic.json: for this kernel's scattered 16-byte reads
    the counter x 1024 B reproduces the sector bytes (no x2).  None when rocprofv3 is not at hand or a pass fails."""
    import csv
    import glob
    import shutil
    import subprocess
    import tempfile
    exe = shutil.which("rocprofv3") or "/opt/rocm/bin/rocprofv3"
    if not os.path.exists(exe):
        return None
    out = {}
    d = tempfile.mkdtemp(prefix="dcn_pmc_", dir="/tmp")
    try:
        for name, counters in (("fetch", ["FETCH_SIZE"]), ("write", ["WRITE_SIZE", "TCC_HIT", "TCC_MISS"])):
            od = os.path.join(d, name)
            cmd = [exe, "--pmc", *counters, "--kernel-include-regex", "scan_kernel", "--output-format", "csv", "-d", od, "-o", "pmc", "--",
                   sys.executable, os.path.abspath(__file__),
                   "--steps", "3", "--warmup", "1", "--pmc-child", "--workload", workload, "--reads", str(reads),
                   "--index-keys", str(index_keys), "--host-genome", str(host_genome)]
            env = dict(os.environ, TMPDIR="/tmp")
            p = subprocess.run(cmd, cwd="/tmp", env=env, capture_output=True, text=True, timeout=120)
            if p.returncode != 0:
                log(f"live PMC pass '{name}' failed ({p.returncode}): {p.stderr[-300:]}")
                return None
            acc = {}
            for path in glob.glob(os.path.join(od, "**", "*counter_collection.csv"), recursive=True):
                for r in csv.DictReader(open(path)):
                    if "scan_kernel<15, false, false, false" in r["Kernel_Name"]:
                        acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
            for k_, v in acc.items():
                out[k_] = sum(v) / len(v)
                out["dispatches"] = len(v)
        if "FETCH_SIZE" not in out or "WRITE_SIZE" not in out:
            return None
        out["hbm_bytes_per_launch"] = (out["FETCH_SIZE"] + out["WRITE_SIZE"]) * 1024.0
        return out
    except Exception as ex:
        log(f"live PMC passes failed: {ex!r}")
        return None
    finally:
        shutil.rmtree(d, ignore_errors=True)


def apply_live_traffic(rf, lt):
    """put a live_traffic() result into a roofline block (the committed figure, if any, stays beside it)"""
    if not lt:
        return False
    rf["traffic_committed"] = {"bytes": rf.get("traffic"), "source": rf.get("traffic_source")}
    rf["traffic"] = lt["hbm_bytes_per_launch"]
    rf["traffic_source"] = ("live: two child runs of this bench under `rocprofv3 --pmc` (FETCH_SIZE; WRITE_SIZE TCC_HIT TCC_MISS), "
                            f"mean over {lt['dispatches']} dispatches of the counting scan kernel")
    rf["traffic_counters"] = {k_: lt[k_] for k_ in ("FETCH_SIZE", "WRITE_SIZE", "TCC_HIT", "TCC_MISS") if k_ in lt}
    rf["traffic_rate_GBps"] = rf["traffic"] / (rf["avg_launch_ms"] * 1e-3) / 1e9
    rf["traffic_frac_of_peak"] = rf["traffic_rate_GBps"] / HBM_PEAK_GBS
    return True


def probe_only_rate(index, device, n=64_000_000, reps=5):
    """The library's own set-membership kernel (dcn_index_contains_device: one 16-byte group read per key, nothing else)
    on n uniformly random keys against the SAME table: the scattered-request rate this box gives this table now.
    Reported beside the documented ceiling (profiles/microbench/probe_patterns.hip), which moves 46 -> 49 G/s from
    box to box."""
    q = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=device)
    out = torch.empty(n, dtype=torch.uint8, device=device)
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
    for _ in range(2):
        index.contains_device(q.data_ptr(), n, out.data_ptr())
    torch.cuda.synchronize()
    ev[0].record()
    for _ in range(reps):
        index.contains_device(q.data_ptr(), n, out.data_ptr())
    ev[1].record()
    torch.cuda.synchronize()
    return n * reps / (ev[0].elapsed_time(ev[1]) * 1e-3)


# ---- oracle checks (the checker, never the thing measured) -----------------------------------------------------------
def sample_of(batch, max_bases):
    """first reads of a batch (whole units), at most max_bases bases -> host arrays"""
    off = batch.d_offsets.cpu().numpy().astype(np.uint64)
    n = int(np.searchsorted(off, max_bases, side="right")) - 1
    n = max(2, min(n, batch.n_reads)) // 2 * 2
    n = min(n, batch.n_reads)
    bases = batch.d_bases[:int(off[n])].cpu().numpy()
    uid = (np.arange(n, dtype=np.uint32) // 2) if batch.d_unit_id is not None else None
    n_units = n // 2 if uid is not None else n
    return bases, off[:n + 1], uid, n_units


def oracle_decisions(oidx, bases, off, uid, params, threads):
    from oracle import oracle as O
    return O.filter_batch(oidx, bases, off, uid, params["abs"], params["rel"], 0, params["deplete"], threads=threads)


def touchable_oracle_index(bases, off, host_keys_sorted, n_rand, threads):
    """CPU oracle set holding exactly the index keys the sample can touch: the sample's minimizer hashes (computed by
    the oracle) that are host-genome keys or mix64(i), 1 <= i <= n_rand.  Membership of the synthetic remainder is
    decided from its definition, so no 950 M-key CPU set has to be built for a bounded check."""
    from oracle import oracle as O
    hs = []
    for r in range(len(off) - 1):
        h, _ = O.minimizer_hashes_and_positions(bases[int(off[r]):int(off[r + 1])], K, W)
        if len(h):
            hs.append(np.asarray(h, dtype=np.uint64))
    h = np.unique(np.concatenate(hs)) if hs else np.zeros(0, np.uint64)
    pos = np.searchsorted(host_keys_sorted, h)
    in_host = (pos < len(host_keys_sorted)) & (host_keys_sorted[np.minimum(pos, len(host_keys_sorted) - 1)] == h)
    i = unmix64(h)
    in_rand = (i >= 1) & (i <= np.uint64(n_rand))
    return O.Index(h[in_host | in_rand], K, W, threads=threads)


def check_against_oracle(oidx, batch, params, max_bases, threads, keep_dev=None):
    bases, off, uid, n_units = sample_of(batch, max_bases)
    keep, hits, total = oracle_decisions(oidx, bases, off, uid, params, threads)
    got_keep = (batch.d_keep if keep_dev is None else keep_dev)[:n_units].cpu().numpy().astype(bool)
    ok = bool((got_keep == keep).all())
    if keep_dev is None:
        ok = ok and bool((batch.d_hits[:n_units].cpu().numpy() == hits).all()) and bool((batch.d_total[:n_units].cpu().numpy() == total).all())
    return ok, f"first {len(off) - 1} reads ({int(off[-1]) / 1e6:.1f} Mbp) of batch 0"


def cpu_baseline(oidx, cores, batch, params, seconds_target=10.0, tuned=False, port_result=None):
    """Time the CPU oracle (all host cores) on a bounded sample of the headline reads and check the GPU's decisions
    on that sample against it.  tuned: the "port-tuned" form of the same arithmetic (oracle/deacon_oracle.c,
    dor_filter_batch_tuned_mt: one workspace per thread, two-stack window minima, epoch-tagged seen-set, rolling
    k-mer values, units dealt in chunks) -- the leg that is not a strawman; it must equal the plain port on its sample."""
    from oracle import oracle as O
    n_total = min(batch.n_reads, 4_000_000 if tuned else 2_000_000)
    bases_np = batch.d_bases[:n_total * READ_LEN].cpu().numpy()

    def run(n, repeats=1):
        off = np.arange(n + 1, dtype=np.uint64) * np.uint64(READ_LEN)
        t = time.time()
        for _ in range(repeats):
            res = O.filter_batch(oidx, bases_np[:n * READ_LEN], off, None, params["abs"], params["rel"], 0,
                                 params["deplete"], threads=cores, tuned=tuned)
        return time.time() - t, res

    probe_n = min(10_000 * cores, n_total)
    dt, _ = run(probe_n)
    rate = probe_n / max(dt, 1e-6)
    n = int(min(n_total, max(probe_n, rate * seconds_target)))
    repeats = max(1, int(round(rate * seconds_target / n)))  # the sample is cycled until ~seconds_target of CPU work
    dt, (keep, hits, total) = run(n, repeats)
    ok = bool((batch.d_keep[:n].cpu().numpy().astype(bool) == keep).all()) and \
        bool((batch.d_hits[:n].cpu().numpy() == hits).all())
    out = {
        "value": n * repeats * READ_LEN / dt / 1e6, "unit": "Mbp/s", "cores": cores, "kind": "port-tuned" if tuned else "port",
        "sample": f"first {n} reads of batch 0 x{repeats} passes ({n * repeats * READ_LEN / 1e9:.2f} Gbp, "
                  f"{dt:.1f} s), oracle/ C restatement{' (tuned form, same arithmetic)' if tuned else ''} with a pthread pool over "
                  f"{cores} threads, same {len(oidx):,}-key index",
        "decisions_match_gpu": ok,
    }
    if tuned:
        # checked equal to the plain port on the reads both ran (keep, distinct hits, totals)
        m = min(n, len(port_result[0])) if port_result is not None else 0
        out["equals_port"] = bool(m > 0 and (keep[:m] == port_result[0][:m]).all() and (hits[:m] == port_result[1][:m]).all()
                                  and (total[:m] == port_result[2][:m]).all())
        out["per_core_Mbp_per_s"] = out["value"] / cores
        out["reference_claim"] = "the reference's README quotes > 2 Gbp/s on an Apple M1 (README.md:14): other hardware, SIMD crates"
    else:
        out["_result"] = (keep, hits, total)
    return out


# ---- the host boundary: dcn_filter_batch* from host memory, PCIe included ----------------------------------------------
def run_host_path(index, batches, params, oidx, cores, calls=6, reads_per_call=10_000_000, kinds=("pageable", "pinned", "packed"),
                  reps=5, world=1, rank=0, coll_device=None):
    """dcn_filter_batch* from host memory, PCIe included: `reads_per_call` x 150 bp per call (the headline's 10 M-read batch
    = 1.5 GB of ASCII by default), three distinct batches rotated, results copied back to the host.  Blocking calls,
    then the submit/wait form with two batches in flight.  Decisions are checked against the oracle on the first 200 k reads
    of every batch when an oracle set is at hand (N = 1), else against the device-resident run of the same batch.
    At world > 1 every rank runs this at the same time (barriers around every timed repetition): eight ranks share
    one host's memory and PCIe root complexes, which is what the per-rank and summed rates show."""
    n_reads = min(batches[0].n_reads, reads_per_call)
    n_bases = n_reads * READ_LEN
    G = (n_bases + 31) // 32

    def setup():
        """everything that can fail for lack of memory (three batches on the host, page-locked copies, the context)"""
        host = [b.d_bases[:n_bases].cpu().numpy() for b in batches]
        off = np.arange(n_reads + 1, dtype=np.uint64) * np.uint64(READ_LEN)
        n_chk = min(200_000, n_reads)
        if oidx is not None:
            want = [oracle_decisions(oidx, h[:n_chk * READ_LEN], off[:n_chk + 1], None, params, cores)[0] for h in host]
            checked = f"first {n_chk} reads of each of the {len(host)} batches vs the CPU oracle"
        else:
            n_chk = n_reads
            want = [b.d_keep[:n_reads].cpu().numpy().astype(bool) for b in batches]
            checked = f"all {n_reads} reads of each of the {len(host)} batches vs the device-resident run of the same batch"
        proc = dcn.FilterProcessor(index, abs_threshold=params["abs"], rel_threshold=params["rel"], deplete=params["deplete"],
                                   max_batch_bases=n_bases, max_batch_reads=n_reads)
        pins = []
        t0 = time.time()
        for h in host:
            pb = dcn.PinnedBuffer(n_bases, np.uint8) if "pinned" in kinds else None
            if pb is not None:
                pb.array[:] = h
            pp, pm = dcn.PinnedBuffer(2 * G, np.uint32), dcn.PinnedBuffer(G, np.uint32)
            pins.append((pb, pp, pm))
        tp = time.time()
        for h, (pb, pp, pm) in zip(host, pins):
            dcn._native.check(dcn._native.lib().dcn_pack_ascii(h.ctypes.data, n_bases, pp.array.ctypes.data, pm.array.ctypes.data, None))
        pack_s = (time.time() - tp) / len(host)
        poff = dcn.PinnedBuffer(n_reads + 1, np.uint64)
        poff.array[:] = off
        keeps = [dcn.PinnedBuffer(n_reads, np.uint8) for _ in range(2)]
        return host, off, n_chk, want, checked, proc, pins, pack_s, poff, keeps, t0

    # At N > 1 the ranks meet at barriers inside this leg: a rank that failed to set up must not leave the others waiting.
    # Every rank learns whether all of them are ready, and the leg runs on all of them or on none.
    st, err = None, None
    try:
        st = setup()
    except Exception as ex:
        err = repr(ex)
    if world > 1:
        ready = torch.tensor([0 if err else 1], dtype=torch.int64, device=coll_device)
        dist.all_reduce(ready, op=dist.ReduceOp.MIN)
        if int(ready.item()) == 0:
            return {"error": f"set-up failed on some rank ({err or 'not this one'}): the leg was skipped on every rank"}
    elif err:
        raise RuntimeError(err)
    host, off, n_chk, want, checked, proc, pins, pack_s, poff, keeps, t0 = st
    log(f"host_path: buffers ready in {time.time() - t0:.1f} s; dcn_pack_ascii {n_bases / pack_s / 1e9:.1f} Gbp/s on the host threads")
    out = {"reads_per_call": n_reads, "bases_per_call": n_bases, "calls": calls, "repetitions": reps,
           "host_pack_Gbp_per_s": n_bases / pack_s / 1e9, "checked": checked,
           "note": "PCIe-inclusive, results copied back; never the headline `value`"}
    lib, P = dcn._native.lib(), proc._params()
    import ctypes as C

    def call(kind, i, keep_arr, submit):
        pb, pp, pm = pins[i % len(pins)]
        t = C.c_uint64()
        kp = keep_arr.ctypes.data
        if kind == "pageable":
            a = (proc._h, host[i % len(host)].ctypes.data, off.ctypes.data, None, n_reads, C.byref(P), kp, None, None)
            rc = lib.dcn_filter_batch_submit(*a, C.byref(t)) if submit else lib.dcn_filter_batch(*a)
        elif kind == "pinned":
            a = (proc._h, pb.array.ctypes.data, poff.array.ctypes.data, None, n_reads, C.byref(P), kp, None, None)
            rc = lib.dcn_filter_batch_submit(*a, C.byref(t)) if submit else lib.dcn_filter_batch(*a)
        else:
            a = (proc._h, pp.array.ctypes.data, pm.array.ctypes.data, poff.array.ctypes.data, None, n_reads, C.byref(P), kp, None, None)
            rc = lib.dcn_filter_batch_packed_submit(*a, C.byref(t)) if submit else lib.dcn_filter_batch_packed(*a)
        dcn._native.check(rc)
        return t.value

    # page-locked ASCII is packed by the host threads as well where they do it with AVX-512 (csrc/api.hip, submit_impl):
    # 0.375 bytes per base on the link then, 1 when it is sent as it is
    try:
        wide_pack = (" avx512bw" in open("/proc/cpuinfo").read() and not os.environ.get("DCN_NO_AVX512")
                     and not os.environ.get("DCN_PINNED_ASCII_DMA") and not os.environ.get("DCN_NO_HOST_PACK"))
    except OSError:
        wide_pack = False
    off_b = 4 if (n_bases < 2**32 and not os.environ.get("DCN_NO_OFF32")) else 8  # offsets go as u32 when they fit (api.hip)
    # the invalid-base mask (0.125 B/bp whole) goes as its non-zero words only: nothing for these N-free reads
    pk_b = 0.25 if not os.environ.get("DCN_NO_SPARSE_MASK") else 0.375
    link_bytes = {"pageable": pk_b * n_bases + off_b * n_reads, "pinned": (pk_b if wide_pack else 1.0) * n_bases + off_b * n_reads,
                  "packed": pk_b * n_bases + off_b * n_reads}
    out["pinned_ascii_transport"] = "packed by the host threads (AVX-512)" if wide_pack else "sent as it is, packed on the device"

    def wait(tk):
        dcn._native.check(lib.dcn_filter_batch_wait(proc._h, tk))

    def fence():
        if world > 1:
            dist.barrier()

    def over_ranks(x):
        """per-rank values of one number (identity at world == 1)"""
        if world == 1:
            return [float(x)]
        t = torch.zeros(world, dtype=torch.float64, device=coll_device)
        t[rank] = float(x)
        dist.all_reduce(t, op=dist.ReduceOp.SUM)
        return [float(v) for v in t.cpu().tolist()]

    for kind in kinds:
        kb = [np.zeros(n_reads, np.uint8), np.zeros(n_reads, np.uint8)] if kind == "pageable" else [k_.array for k_ in keeps]
        # untimed: every batch once through both forms, checked (also warms the second slot, whose buffers are
        # allocated when it is first used)
        ok = True
        for i in range(len(host)):
            call(kind, i, kb[0], False)
            ok = ok and bool((kb[0][:n_chk].astype(bool) == want[i][:n_chk]).all())
        for i in range(0, len(host) + 1, 2):
            tks = [(j, call(kind, j % len(host), kb[j % 2], True)) for j in (i, i + 1)]
            for j, tk in tks:
                wait(tk)
                ok = ok and bool((kb[j % 2][:n_chk].astype(bool) == want[j % len(host)][:n_chk]).all())
        # timed: blocking calls, then submit / wait with two batches in flight; `reps` repetitions of `calls` calls each,
        # the median is reported and all are listed
        def blocking():
            fence()
            t0 = time.perf_counter()
            for i in range(calls):
                call(kind, i, kb[0], False)
            dt = time.perf_counter() - t0
            fence()
            return dt

        def pipelined():
            fence()
            t0 = time.perf_counter()
            tickets = []
            for i in range(calls):
                if len(tickets) == 2:
                    wait(tickets.pop(0))
                tickets.append(call(kind, i, kb[i % 2], True))
            for tk in tickets:
                wait(tk)
            dt = time.perf_counter() - t0
            fence()
            return dt

        reps_b = sorted(blocking() for _ in range(reps))
        reps_p = sorted(pipelined() for _ in range(reps))
        dt = reps_b[reps // 2]
        entry = {"value": calls * n_bases / dt / 1e6, "unit": "Mbp/s", "ms_per_call": dt / calls * 1e3,
                 "repetitions_Mbp_per_s": [calls * n_bases / x / 1e6 for x in reps_b],
                 "link_bytes_per_call": link_bytes[kind], "link_GBps": link_bytes[kind] * calls / dt / 1e9,
                 "link_frac_of_pcie5_x16": link_bytes[kind] * calls / dt / 1e9 / PCIE_PEAK_GBS,
                 "decisions_match_gpu": ok}
        dt = reps_p[reps // 2]
        entry["two_in_flight"] = {"value": calls * n_bases / dt / 1e6, "unit": "Mbp/s", "ms_per_call": dt / calls * 1e3,
                                  "repetitions_Mbp_per_s": [calls * n_bases / x / 1e6 for x in reps_p],
                                  "link_GBps": link_bytes[kind] * calls / dt / 1e9}
        if world > 1:  # every rank measured at the same time: per rank, and the node's sum
            for e_ in (entry, entry["two_in_flight"]):
                e_["per_rank_Mbp_per_s"] = over_ranks(e_["value"])
                e_["sum_over_ranks_Mbp_per_s"] = float(sum(e_["per_rank_Mbp_per_s"]))
            entry["decisions_match_gpu_all_ranks"] = bool(min(over_ranks(1.0 if ok else 0.0)) == 1.0)
        out[kind] = entry
        log(f"host_path.{kind}: {entry['value'] / 1e3:.1f} Gbp/s blocking, {entry['two_in_flight']['value'] / 1e3:.1f} Gbp/s with two in flight, "
            f"check ok={ok}" + (f"; node sum {entry['two_in_flight']['sum_over_ranks_Mbp_per_s'] / 1e3:.1f} Gbp/s over {world} ranks" if world > 1 else ""))
    proc.close()
    return out


# ---- what is printed -------------------------------------------------------------------------------------------------
LINE_TARGET, LINE_CAP = 4096, 8192


def _r(x, sig=6):
    """floats to `sig` significant digits (what a reader of the line needs; the detail file keeps them whole)"""
    if isinstance(x, float):
        if x != x or x in (float("inf"), float("-inf")):
            return None  # the line is dumped with allow_nan=False: a figure that could not be computed is null
        return float(f"{x:.{sig}g}")
    return x


def _pick(d, keys):
    return {k_: (_short(d[k_], 80) if isinstance(d[k_], str) else _r(d[k_])) for k_ in keys
            if isinstance(d, dict) and k_ in d and not isinstance(d[k_], (dict, list))}


def _short(text, n):
    text = str(text)
    return text if len(text) <= n else text[:n - 3] + "..."


def compact_line(out):
    """The contract's JSON line from the full result `out`: the contract's keys whole, the roofline and cpu_baseline
    objects with the figures a check recomputes, and per extra leg its rate and whether the oracle agreed.  Free text is
    cut to a bounded length, so no leg, sample description or error message can make the line outgrow LINE_CAP."""
    c = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                    "vs_baseline", "dtype", "data"))
    for k_, n_ in (("metric", 120), ("unit", 16), ("scaling", 8), ("dtype", 16), ("data", 60)):
        if isinstance(c.get(k_), str):
            c[k_] = _short(c[k_], n_)
    c["parity"] = _short(out.get("parity", ""), 160)
    cfg = out.get("config") or {}
    c["config"] = _pick(cfg, ("index_keys", "reads_per_batch_per_gpu", "bases_per_batch_per_gpu", "distinct_batches_rotated", "k", "w",
                              "host_fraction", "parallelism"))
    c["config"]["workload"] = _short(cfg.get("workload", ""), 200)
    rf = out.get("roofline") or {}
    c["roofline"] = _pick(rf, ("bound", "achieved", "peak", "unit", "frac", "traffic", "algorithmic_bytes_per_launch", "avg_launch_ms",
                               "minimizers_per_launch", "scattered_probes_per_s", "probe_ceiling_replay_per_s", "frac_of_probe_ceiling",
                               "traffic_frac_of_peak"))
    c["roofline"]["kernel"] = _short(rf.get("kernel", ""), 80)
    c["roofline"].setdefault("traffic", None)
    if rf.get("traffic_source"):
        c["roofline"]["traffic_source"] = "live rocprofv3 --pmc child runs" if str(rf["traffic_source"]).startswith("live") else _short(rf["traffic_source"], 60)
    for key in ("cpu_baseline", "cpu_baseline_tuned"):
        cb = out.get(key)
        if cb:
            c[key] = _pick(cb, ("value", "unit", "cores", "kind", "decisions_match_gpu", "equals_port"))
            if key == "cpu_baseline":
                c[key]["sample"] = _short(cb.get("sample", ""), 200)
        elif key == "cpu_baseline":
            c[key] = None
    if out.get("cpu_baseline_error"):
        c["cpu_baseline_error"] = _short(out["cpu_baseline_error"], 120)
    if "decisions_match_gpu" in out:
        c["decisions_match_gpu"] = out["decisions_match_gpu"]
    do = out.get("decisions_only")
    if do:
        c["decisions_only"] = _pick(do, ("value", "ms_per_step", "decisions_identical_to_counting_mode"))
    col = out.get("collective")
    if col:
        c["collective"] = _pick(col, ("backend", "world_size", "total_bp_all_reduced", "total_bp_expected", "total_bp_matches", "c_abi_rccl_matches"))
    oks = []

    def leg(d):
        if "error" in d:
            oks.append(False)
            return {"error": _short(d["error"], 100)}
        e = {"value": _r(d.get("value"))}
        for k_ in ("decisions_match_gpu", "decisions_match"):
            if k_ in d:
                e["decisions_match"] = bool(d[k_])
                oks.append(bool(d[k_]))
        return e

    if out.get("workloads"):
        c["legs"] = {}
        for name, d in out["workloads"].items():
            e = leg(d)
            if "error" not in e:
                if isinstance(d.get("decisions_only"), dict):
                    e["decisions_only"] = _r(d["decisions_only"].get("value"))
                r2 = d.get("roofline") or {}
                e.update(_pick(r2, ("frac", "avg_launch_ms", "traffic")))
            c["legs"][_short(name, 24)] = e
    hp = out.get("host_path")
    if hp:
        c["host_path"] = {"unit": "Mbp/s, PCIe included; never `value`"} if "error" not in hp else {"error": _short(hp["error"], 100)}
        c["host_path"].update(_pick(hp, ("reads_per_call",)))
        for kind in ("pageable", "pinned", "packed"):
            if isinstance(hp.get(kind), dict):
                e = leg(hp[kind])
                tf = hp[kind].get("two_in_flight") or {}
                e["two_in_flight"] = _r(tf.get("value"))
                e["link_GBps"] = _r(hp[kind].get("link_GBps"))
                if "sum_over_ranks_Mbp_per_s" in tf:
                    e["node_sum_two_in_flight"] = _r(tf["sum_over_ranks_Mbp_per_s"])
                c["host_path"][kind] = e
    cli = out.get("cli")
    if cli:
        c["cli"] = {"unit": "Mbp/s file to file"} if "error" not in cli else {"error": _short(cli["error"], 100)}
        for name, d in cli.items():
            if not isinstance(d, dict):
                continue
            e = {}
            for src, dst in (("Mbp_per_s_incl_index_load", "value"), ("Mbp_per_s_filter_only", "filter_only"), ("decisions_match", "decisions_match")):
                if src in d:
                    e[dst] = _r(d[src])
            if "decisions_match" in d:
                oks.append(bool(d["decisions_match"]))
            if e:
                c["cli"][_short(name, 24)] = e
    for key in ("cpu_baseline", "cpu_baseline_tuned"):
        if out.get(key) and "decisions_match_gpu" in out[key]:
            oks.append(bool(out[key]["decisions_match_gpu"]))
    c["all_checks_ok"] = bool(all(oks)) if oks else None
    c["checks"] = len(oks)
    c["bench_wall_s"] = _r(out.get("bench_wall_s"), 4)
    c["detail"] = out.get("detail_file")
    return c


def emit(out, detail_path):
    """Write the full result to `detail_path` and (one line) to stderr, then print the compact contract line as the LAST
    stdout line.  Strict JSON (allow_nan=False); should the compact line ever exceed LINE_CAP the optional blocks are
    dropped, in this order, until it fits."""
    out["detail_file"] = None
    if detail_path:
        try:
            with open(detail_path, "w") as f:
                json.dump(out, f, indent=1, default=str)
            out["detail_file"] = os.path.basename(detail_path)
        except OSError as ex:
            log(f"could not write {detail_path}: {ex!r}")
    try:
        print("[bench detail] " + json.dumps(out, default=str), file=sys.stderr, flush=True)
    except Exception:
        pass
    c = compact_line(out)
    line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    for drop in ("cli", "host_path", "legs", "cpu_baseline_tuned", "decisions_only", "collective"):
        if len(line) <= LINE_CAP:
            break
        c.pop(drop, None)
        c["dropped_for_size"] = c.get("dropped_for_size", []) + [drop]
        line = json.dumps(c, allow_nan=False, separators=(",", ":"))
    sys.stderr.flush()
    print(line, flush=True)
    return line


def stub_main(args, rank, world):
    """DCN_BENCH_STUB=1: a rehearsal of the launch, rendezvous, barriers, the counters' all-reduce, the max-over-ranks clock
    and the printed line with NO filter engine behind it (a step adds up the lengths of a synthetic batch on the CPU).
    For tests of the N > 1 plumbing on a machine without a GPU (tests/test_bench_line.py); its line says what it is."""
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend=os.environ.get("DCN_BENCH_BACKEND", "gloo"))
    cpu = torch.device("cpu")
    n_reads = min(args.reads, 100_000)
    lens = np.full(n_reads, READ_LEN, np.int64)
    local = {n: 0 for n in dcn._native.STAT_NAMES}

    def step():
        local["total_seqs"] += n_reads
        local["total_bp"] += int(lens.sum())

    for _ in range(args.warmup):
        step()
    local = {n: 0 for n in dcn._native.STAT_NAMES}
    if world > 1:
        dist.barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    counters = dcn.distributed.allreduce_counters(local, device=cpu)
    if world > 1:
        dist.barrier()
    t = torch.tensor([time.perf_counter() - t0], dtype=torch.float64)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    if rank == 0:
        expected = world * args.steps * n_reads * READ_LEN
        out = {"metric": "STUB: launcher rehearsal, nothing is filtered (DCN_BENCH_STUB=1)", "value": counters["total_bp"] / elapsed / 1e6,
               "unit": "Mbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": elapsed / args.steps * 1e3,
               "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "u64", "data": "stub", "parity": "not applicable (stub)",
               "config": {"workload": "stub: lengths of synthetic 150 bp reads added up on the CPU", "reads_per_batch_per_gpu": n_reads,
                          "bases_per_batch_per_gpu": n_reads * READ_LEN, "k": K, "w": W, "parallelism": f"reads sharded x{world}"},
               "roofline": {"bound": "hbm", "kernel": "none (stub)", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": None, "traffic": None},
               "cpu_baseline": None,
               "collective": {"backend": str(dist.get_backend()) if world > 1 else None, "world_size": world,
                              "total_bp_all_reduced": int(counters["total_bp"]), "total_bp_expected": expected,
                              "total_bp_matches": int(counters["total_bp"]) == expected},
               "bench_wall_s": time.time() - T0}
        emit(out, args.detail)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=21)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=10_000_000,
                    help="reads per batch per GPU (150 bp each); the default is SURVEY.md 8d config 2 whole: 10 M x 150 bp = 1.5 Gbp per step")
    ap.add_argument("--index-keys", type=int, default=PANHUMAN_KEYS)
    ap.add_argument("--host-genome", type=int, default=64_000_000,
                    help="bases of the synthetic host genome (SURVEY.md 8d config 2: 64 Mbp, ~8 M of the index keys)")
    ap.add_argument("--workload", choices=["short", "paired", "long", "mixed"], default="short",
                    help="the workload `value` is measured on.  short: configs[1] 150 bp single reads (the headline); "
                         "paired: configs[3] 2x150 bp --deplete; long: configs[2] ONT-style lognormal reads, mean 10 kbp; "
                         "mixed: configs[4]'s stream, half of the bases long and half short in one batch, --deplete")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extras", action="store_true",
                    help="skip the other configs and the host-path measurements that follow the headline at N = 1")
    ap.add_argument("--extras", default=DEFAULT_EXTRAS,
                    help="which of the extra measurements to run (comma separated); all of them: " + ALL_EXTRAS)
    ap.add_argument("--pmc-child", action="store_true",
                    help="(internal) the run live_traffic() starts under rocprofv3 --pmc: the timed steps only, nothing else measured or written")
    ap.add_argument("--detail", default=os.path.join(ROOT, "bench_detail.json"),
                    help="where the full result goes (the last stdout line is its compact form); '' = nowhere")
    args = ap.parse_args()
    if args.pmc_child:
        args.no_cpu_baseline, args.no_extras, args.detail = True, True, ""

    # torch sizes its CPU thread pool by the machine (256 hardware threads on the GPU box) while the job's CPU quota is a
    # 16th of it: one parallel CPU op would spend the quota of its 100 ms period at once and the host legs that follow
    # would be measured throttled (cpu.stat: nr_throttled)
    torch.set_num_threads(max(1, min(host_cores(), torch.get_num_threads())))
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:  # (a plain `bench.py --gpus N` has become a torch.distributed.run child above: self_launch_if_needed)
        log(f"--gpus {args.gpus} but WORLD_SIZE={world}: running as the launcher says")
        args.gpus = world
    if os.environ.get("DCN_BENCH_STUB"):
        return stub_main(args, rank, world)
    # rehearsal of the N > 1 path on a one-GPU box (profiles/rehearse_two_ranks.sh): every rank on GPU 0, collectives
    # on gloo (RCCL does not take two ranks on one device).  The driver's scaling run uses neither variable.
    backend = os.environ.get("DCN_BENCH_BACKEND", "nccl")
    # This rank's host threads (the library's packers, result copies, torch, the tool's child processes) go to the cores next
    # to its GPU, before anything touches the GPU or starts a thread pool: at N > 1 eight ranks share one host's memory and
    # PCIe root complexes, and at N = 1 a job whose threads roam over both sockets of the box packs and stages half of its
    # bytes across the socket link (same box, host legs from pageable memory: 76-83 Gbp/s unbound, 100-108 bound;
    # page-locked ASCII 104-116 -> 119-120).  DCN_BENCH_NO_BIND=1 leaves the affinity alone.
    binding = None
    if not os.environ.get("DCN_BENCH_NO_BIND"):
        lw = int(os.environ.get("LOCAL_WORLD_SIZE", world))
        binding = dcn.distributed.bind_rank_to_gpu_cpus(local_rank, lw, gpu_of_rank=[0] * lw if os.environ.get("DCN_BENCH_SINGLE_DEVICE") else None)
    if os.environ.get("DCN_BENCH_SINGLE_DEVICE"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    coll_device = device if backend == "nccl" else torch.device("cpu")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=device)
        else:
            dist.init_process_group(backend=backend)

    # ---- index: host-genome minimizers + mix64 keys, identical on every rank ------------------------------------
    genome_dev = make_host_genome(args.host_genome, 3, device)
    index, keys, host_keys, n_rand, index_build_s = build_index(genome_dev, args.index_keys, local_rank, world, rank)

    # ---- reads: resident in HBM before the timed region -----------------------------------------------------------
    P_SHORT = {"abs": 2, "rel": 0.01, "deplete": False}
    P_PAIRED = {"abs": 2, "rel": 0.01, "deplete": True}
    params = P_PAIRED if args.workload in ("paired", "mixed") else P_SHORT
    seeds = {"short": 5, "long": 6, "paired": 7, "mixed": 9}
    batches = make_batches(args.workload, genome_dev, args.reads, seeds[args.workload] + 1000 * rank, device)
    torch.cuda.synchronize()
    log(f"{args.workload}: {len(batches)} batches of {batches[0].n_reads:,} reads / {batches[0].n_bases / 1e6:.0f} Mbp resident in HBM")

    head, counters, elapsed, bases_done = run_device_workload(index, batches, params, args.steps, args.warmup, world, device,
                                                              reserve_long=args.workload in ("long", "mixed"), coll_device=coll_device)
    # what the job's one collective delivered, checked against an independent sum of what every rank put in
    t_exp = torch.tensor([bases_done], dtype=torch.int64, device=coll_device)
    if world > 1:
        dist.all_reduce(t_exp, op=dist.ReduceOp.SUM)
    collective = {"op": "all_reduce(SUM) of the six ProcessingStats counters (u64), once, inside the timed region",
                  "backend": (str(dist.get_backend()) + (" (RCCL)" if backend == "nccl" else "")) if world > 1 else None,
                  "world_size": dist.get_world_size() if world > 1 else 1,
                  "total_bp_all_reduced": int(counters["total_bp"]), "total_bp_expected": int(t_exp.item()),
                  "total_bp_matches": int(counters["total_bp"]) == int(t_exp.item()),
                  "cpu_binding": {k_: v for k_, v in binding.items() if k_ != "cpus"} if binding else None,
                  "c_abi_rccl": None}
    if world > 1:  # every rank's binding, for the record: number of CPUs each rank's host threads may use
        t_b = torch.zeros(world, dtype=torch.int64, device=coll_device)
        t_b[rank] = len((binding or {}).get("cpus") or os.sched_getaffinity(0))
        dist.all_reduce(t_b, op=dist.ReduceOp.SUM)
        collective["cpus_per_rank"] = [int(x) for x in t_b.cpu().tolist()]
    # N > 1: the PCIe-inclusive legs on every rank at the same time (8 ranks share one host: the part of the path that can
    # fail to scale), after the contract's timed region
    host_path_all = None
    if world > 1 and not args.no_extras and args.workload == "short":
        try:
            host_path_all = run_host_path(index, batches, params, None, host_cores(), calls=4, reads_per_call=args.reads,
                                          kinds=("pageable", "pinned", "packed"), reps=3, world=world, rank=rank, coll_device=coll_device)
        except Exception as ex:  # every rank fails or none does (same code, same sizes); never take the line with it
            log(f"host_path at N = {world} failed: {ex!r}")
            host_path_all = {"error": repr(ex)}
    if world > 1 and not args.pmc_child:
        collective["c_abi_rccl"] = "checked after the line is printed: stderr and bench_detail.json carry the result"
    if rank == 0:
        total_bp = counters["total_bp"]
        rf = head["roofline"]
        traffic = committed_traffic(rf, args.workload, batches[0].n_bases, int(index.n_keys), args.host_genome)
        try:
            if not args.pmc_child:
                rf["probe_only_kernel_live_per_s"] = probe_only_rate(index, device)
                rf["frac_of_probe_only_kernel_live"] = rf["scattered_probes_per_s"] / rf["probe_only_kernel_live_per_s"]
        except Exception as ex:
            log(f"probe-only measurement failed: {ex!r}")
        try:
            if not args.pmc_child:
                # ONE bound the kernel cannot exceed: the same table serving the same key stream -- batch 0's own
                # minimizer hashes, in the kernel's order -- to a kernel that does nothing but issue the home-group reads
                # (dcn_index_probe_ceiling: best of six launch shapes).  Beside it, the same for uniformly random groups (no
                # reuse at all: every request an L2 miss), which is what the documented 46 G/s stands for.
                t_c = time.time()
                rf["probe_ceiling_random_per_s"] = index.probe_ceiling(None, 1 << 27, reps=3)
                if args.workload == "short":
                    b0 = batches[0]
                    pc = dcn.FilterProcessor(index, max_batch_bases=b0.n_bases, max_batch_reads=b0.n_reads)
                    _, hs, _ = pc.minimizer_hashes_batch(b0.d_bases.cpu().numpy(), b0.d_offsets.cpu().numpy().astype(np.uint64))
                    pc.close()
                    d_h = torch.from_numpy(hs.view(np.int64)).to(device)
                    rf["probe_ceiling_replay_per_s"] = index.probe_ceiling(d_h.data_ptr(), d_h.numel(), reps=3)
                    rf["probe_ceiling_replay_stream"] = f"the {d_h.numel():,} valid minimizer hashes of batch 0, in read order"
                    rf["frac_of_probe_ceiling"] = rf["scattered_probes_per_s"] / rf["probe_ceiling_replay_per_s"]
                    del d_h, hs
                log(f"probe ceiling: random {rf['probe_ceiling_random_per_s'] / 1e9:.1f} G/s, replay of batch 0's hashes "
                    f"{rf.get('probe_ceiling_replay_per_s', 0) / 1e9:.1f} G/s; the scan kernel sustains {rf['scattered_probes_per_s'] / 1e9:.1f} G/s "
                    f"({time.time() - t_c:.1f} s)")
        except Exception as ex:
            log(f"probe ceiling measurement failed: {ex!r}")
        names = {"short": "configs[1]: 150 bp reads vs panhuman-1-sized index, -a 2 -r 0.01, inputs resident in HBM as ASCII",
                 "paired": "configs[3]: paired 2x150 bp --deplete vs panhuman-1-sized index, inputs resident in HBM as ASCII",
                 "long": "configs[2]: ONT-style lognormal reads (mean 10 kbp) vs panhuman-1-sized index, -a 2 -r 0.01, inputs resident in HBM as ASCII",
                 "mixed": "configs[4]'s stream shape: half of the bases in ONT-style long reads, half in 150 bp reads, interleaved, --deplete, inputs resident in HBM as ASCII"}
        out = {
            "metric": "Mbp/s filtered (k=31,w=15 vs panhuman-1-sized index), decisions bit-exact vs CPU",
            "value": head["value"], "unit": "Mbp/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": head["ms_per_step"], "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic",
            "parity": "bit-exact against the in-repo CPU oracle; parity with the reference's crates is UNPINNED at value "
                      "level (no Rust toolchain, the reference holds no value vectors: DESIGN.md section 2)",
            "config": {
                "workload": names[args.workload],
                "index_keys": int(index.n_keys), "index_table_bytes": int(index.table_bytes),
                "reads_per_batch_per_gpu": batches[0].n_reads,
                "bases_per_batch_per_gpu": batches[0].n_bases, "distinct_batches_rotated": len(batches),
                "k": K, "w": W, "host_fraction": 0.5, "host_genome_bases": args.host_genome,
                "parallelism": f"reads sharded x{world}, index replicated",
            },
            "roofline": rf,
            "stage_ms_per_launch": head["stage_ms_per_launch"],
            "kept_fraction": head["kept_fraction"],
            "decisions_only": head["decisions_only"],
            "index_build_s": index_build_s,
            "collective": collective,
        }
        if host_path_all:
            out["host_path"] = host_path_all
        extras = [] if (args.no_extras or world > 1) else [e for e in args.extras.split(",") if e]
        oidx, cores = None, host_cores()
        need_oracle = (not args.no_cpu_baseline and world == 1) or any(e in extras for e in ("long", "paired", "host_path"))
        out["cpu_baseline"] = None
        try:
            if need_oracle:
                from oracle import oracle as O
                t0 = time.time()
                oidx = O.Index(keys, K, W, threads=cores)
                log(f"CPU oracle set of {len(oidx):,} keys built with {cores} threads in {time.time() - t0:.1f} s (checker + cpu_baseline)")
            if world == 1 and not args.no_cpu_baseline and args.workload == "short":
                out["cpu_baseline"] = cpu_baseline(oidx, cores, batches[0], params)
                port_result = out["cpu_baseline"].pop("_result")
                log(f"cpu_baseline: {out['cpu_baseline']['value']:.0f} Mbp/s on {cores} cores, decisions match: {out['cpu_baseline']['decisions_match_gpu']}")
                out["cpu_baseline_tuned"] = cpu_baseline(oidx, cores, batches[0], params, seconds_target=4.0, tuned=True, port_result=port_result)
                del port_result
                log(f"cpu_baseline_tuned: {out['cpu_baseline_tuned']['value']:.0f} Mbp/s on {cores} cores, equals the port: "
                    f"{out['cpu_baseline_tuned']['equals_port']}, decisions match: {out['cpu_baseline_tuned']['decisions_match_gpu']}")
            elif world == 1 and oidx is not None:
                ok, what = check_against_oracle(oidx, batches[0], params, 300_000_000, cores)
                out["decisions_match_gpu"] = ok
        except Exception as ex:  # the checker failing (e.g. no host memory for its key set) must not take the line with it
            log(f"cpu_baseline / oracle failed: {ex!r}")
            out["cpu_baseline_error"] = repr(ex)
            oidx = None

        # ---- everything below runs after the contract's timed region and never enters `value` --------------------------
        workloads, host_path, cli = {}, None, None
        short_batches = batches if args.workload == "short" else None
        del keys
        for e in extras:
            t_e = time.time()
            try:
                if e in ("long", "paired") and e != args.workload:
                    bs = make_batches(e, genome_dev, args.reads, seeds[e], device)
                    p = P_PAIRED if e == "paired" else P_SHORT
                    r, _, _, _ = run_device_workload(index, bs, p, 12, 3, 1, device, reserve_long=e == "long")
                    ok, what = check_against_oracle(oidx, bs[0], p, 200_000_000, cores)
                    r["decisions_match_gpu"], r["oracle_sample"] = ok, what + f" vs the oracle's {len(oidx):,}-key set (keep, hits, totals)"
                    r["workload"] = names[e]
                    committed_traffic(r["roofline"], e, bs[0].n_bases, int(index.n_keys), args.host_genome)
                    if "pmc_legs" in extras:  # ... and measured in this run as well (the same batches: same generator, same seed)
                        apply_live_traffic(r["roofline"], live_traffic(e, args.reads, args.index_keys, args.host_genome))
                    workloads[e] = r
                    del bs
                    log(f"workloads.{e}: {r['value'] / 1e3:.1f} Gbp/s counting, {r['decisions_only']['value'] / 1e3:.1f} Gbp/s decisions only, "
                        f"scan {r['stage_ms_per_launch']['scan']:.2f} ms, distinct {r['stage_ms_per_launch']['distinct']:.2f} ms, oracle ok={ok} ({time.time() - t_e:.0f} s)")
                elif e == "host_path":
                    if short_batches is None:
                        short_batches = make_batches("short", genome_dev, args.reads, seeds["short"], device)
                    host_path = run_host_path(index, short_batches, P_SHORT, oidx, cores, reads_per_call=args.reads)
                elif e == "pmc":
                    # roofline.traffic measured in THIS run (it used to be read from a committed file only):
                    # child runs of this file under rocprofv3 --pmc, on the same GPU, after the timed region
                    lt = live_traffic(args.workload, args.reads, args.index_keys, args.host_genome)
                    if apply_live_traffic(out["roofline"], lt):
                        rf_ = out["roofline"]
                        log(f"live PMC: {rf_['traffic'] / 1e9:.2f} GB per scan launch ({rf_['traffic'] / rf_['algorithmic_bytes_per_launch']:.2f} x algorithmic), "
                            f"{time.time() - t_e:.0f} s")
                elif e == "cli":
                    # `deacon-hip` file to file, as a user runs it (bench_cli.py): configs[0] at its stated shape, then
                    # search / host depletion / two files of mates against this index written as an index FILE
                    import bench_cli
                    if oidx is not None:
                        del oidx  # 8.6 GB of host memory the tool's page cache can use; the legs below build their own small sets
                        oidx = None
                    cli = {"plumbing": bench_cli.plumbing(threads=cores)}
                    log(f"cli.plumbing: decisions_match={cli['plumbing']['decisions_match']} ({time.time() - t_e:.0f} s)")
                    sizes = None
                    if os.environ.get("DCN_BENCH_CLI_READS"):  # smaller files for a rehearsal
                        n_ = int(os.environ["DCN_BENCH_CLI_READS"])
                        sizes = {"search50": n_, "deplete95": n_, "paired": n_ // 2}
                    cli.update(bench_cli.file_to_file(index, genome_dev, host_keys, n_rand, make_reads, make_pairs,
                                                      touchable_oracle_index, cores, sizes=sizes, log=log))
                    cli["decisions_match"] = all(v.get("decisions_match", True) for v in cli.values() if isinstance(v, dict))
                    cli["wall_s"] = time.time() - t_e
            except Exception as ex:  # an extra that fails must not take the contract's line with it
                log(f"extra '{e}' failed: {ex!r}")
                if e == "host_path":
                    host_path = {"error": repr(ex)}
                elif e == "cli":
                    cli = {"error": repr(ex), "decisions_match": False}
                else:
                    workloads[e] = {"error": repr(ex)}
                torch.cuda.empty_cache()
        if oidx is not None:
            del oidx
        torch.cuda.empty_cache()
        for e in extras:
            t_e = time.time()
            try:
                if e == "host1g":
                    # sensitivity point: a >= 1 Gbp host genome (a real 3 Gbp host leaves far fewer of a batch's probes in cache)
                    g2 = make_host_genome(1_000_000_000, 13, device)
                    idx2, keys2, hk2, nr2, _ = build_index(g2, args.index_keys, local_rank)  # (extras run at world == 1 only)
                    del keys2
                    bs = make_batches("short", g2, args.reads, 15, device)
                    r, _, _, _ = run_device_workload(idx2, bs, P_SHORT, 12, 3, 1, device)
                    b_, o_, u_, nu_ = sample_of(bs[0], 15_000_000)
                    small = touchable_oracle_index(b_, o_, hk2, nr2, cores)
                    keep, hits, total = oracle_decisions(small, b_, o_, u_, P_SHORT, cores)
                    ok = bool((bs[0].d_keep[:nu_].cpu().numpy().astype(bool) == keep).all()) and \
                        bool((bs[0].d_hits[:nu_].cpu().numpy() == hits).all())
                    r["decisions_match_gpu"] = ok
                    r["oracle_sample"] = (f"first {len(o_) - 1} reads of batch 0 vs the oracle on the {len(small):,} index keys the sample "
                                          "can touch (membership of the mix64 remainder decided from its definition)")
                    r["workload"] = f"configs[1] with a 1 Gbp host genome ({len(hk2):,} of the {int(idx2.n_keys):,} keys are host minimizers)"
                    workloads["host1g"] = r
                    del bs
                    if "host95" in extras:
                        # the shape of a host-depletion run: 95 % of the reads come from the (1 Gbp) host, `-d` keeps the rest.
                        # Decisions only is what such a run needs; reads from the host stop being probed after abs_threshold hits
                        t_9 = time.time()
                        bs = make_batches("short", g2, args.reads, 17, device, host_frac=0.95)
                        P_DEPLETE = {"abs": 2, "rel": 0.01, "deplete": True}
                        r9, _, _, _ = run_device_workload(idx2, bs, P_DEPLETE, 12, 3, 1, device)
                        b_, o_, u_, nu_ = sample_of(bs[0], 15_000_000)
                        small = touchable_oracle_index(b_, o_, hk2, nr2, cores)
                        keep, hits, total = oracle_decisions(small, b_, o_, u_, P_DEPLETE, cores)
                        ok9 = bool((bs[0].d_keep[:nu_].cpu().numpy().astype(bool) == keep).all()) and \
                            bool((bs[0].d_hits[:nu_].cpu().numpy() == hits).all()) and \
                            bool((bs[0].d_keep2[:nu_].cpu().numpy().astype(bool) == keep).all())
                        r9["decisions_match_gpu"] = ok9
                        r9["oracle_sample"] = f"first {len(o_) - 1} reads of batch 0, counting and decisions-only mode, vs the oracle on the {len(small):,} index keys the sample can touch"
                        r9["workload"] = "a host-depletion run: 150 bp reads, 95 % from the 1 Gbp host genome, -a 2 -r 0.01 --deplete (5 % kept)"
                        workloads["host95"] = r9
                        del bs
                        log(f"workloads.host95: {r9['value'] / 1e3:.1f} Gbp/s counting, {r9['decisions_only']['value'] / 1e3:.1f} Gbp/s decisions only, oracle ok={ok9} ({time.time() - t_9:.0f} s)")
                    idx2.close()
                    del g2, hk2
                    torch.cuda.empty_cache()
                    log(f"workloads.host1g: {r['value'] / 1e3:.1f} Gbp/s counting, scan {r['stage_ms_per_launch']['scan']:.2f} ms, oracle ok={ok} ({time.time() - t_e:.0f} s)")
                elif e == "union950m":
                    index.close()
                    idx3, keys3, hk3, nr3, tb = build_index(genome_dev, UNION_KEYS, local_rank)
                    del keys3
                    for key_, kind_, seed_ in (("union950m", "mixed", 29), ("union950m_paired", "paired", 27)):
                        if key_ not in extras:
                            continue
                        bs = make_batches(kind_, genome_dev, args.reads, seed_, device)
                        r, _, _, _ = run_device_workload(idx3, bs, P_PAIRED, 12, 3, 1, device, reserve_long=kind_ == "mixed")
                        b_, o_, u_, nu_ = sample_of(bs[0], 15_000_000)
                        small = touchable_oracle_index(b_, o_, hk3, nr3, cores)
                        keep, hits, total = oracle_decisions(small, b_, o_, u_, P_PAIRED, cores)
                        ok = bool((bs[0].d_keep[:nu_].cpu().numpy().astype(bool) == keep).all()) and \
                            bool((bs[0].d_hits[:nu_].cpu().numpy() == hits).all())
                        r["decisions_match_gpu"] = ok
                        r["oracle_sample"] = (f"first {len(o_) - 1} reads of batch 0 vs the oracle on the {len(small):,} index keys the sample "
                                              "can touch (membership of the mix64 remainder decided from its definition)")
                        shape = ("configs[4]'s stream: half of the bases in ONT-style long reads (lognormal, mean 10 kbp), half in 150 bp "
                                 "reads, interleaved in one batch, --deplete" if kind_ == "mixed" else "paired 2x150 bp --deplete")
                        r["workload"] = f"configs[4]-sized table: {int(idx3.n_keys):,} keys (2^31 groups, 34 GB); {shape}"
                        if kind_ == "mixed":
                            committed_traffic(r["roofline"], "mixed", bs[0].n_bases, int(idx3.n_keys), args.host_genome)
                            if "pmc_legs" in extras:  # (the child draws its own mixed batches: same generator, another seed)
                                apply_live_traffic(r["roofline"], live_traffic("mixed", args.reads, UNION_KEYS, args.host_genome))
                        r["table_build_s"] = tb
                        workloads[key_] = r
                        del bs
                    r = workloads["union950m"]
                    idx3.close()
                    index = None
                    log(f"workloads.union950m: {r['value'] / 1e3:.1f} Gbp/s counting, scan {r['stage_ms_per_launch']['scan']:.2f} ms, oracle ok={ok} ({time.time() - t_e:.0f} s)")
            except Exception as ex:  # an extra that fails must not take the contract's line with it
                log(f"extra '{e}' failed: {ex!r}")
                if e == "host_path":
                    host_path = {"error": repr(ex)}
                else:
                    workloads[e] = {"error": repr(ex)}
                torch.cuda.empty_cache()
        if workloads:
            out["workloads"] = workloads
        if host_path:
            out["host_path"] = host_path
        if cli:
            out["cli"] = cli
        out["bench_wall_s"] = time.time() - T0
        emit(out, args.detail)
    if world > 1 and not args.pmc_child:
        # The LAST thing the ranks do together, AFTER rank 0 has printed the line: the C ABI's own RCCL communicator has never met
        # more than one device before the driver's scaling run, and nothing it does (a fault inside the library included) may
        # cost that run its line.  Result: stderr, and `collective.c_abi_rccl` of bench_detail.json.
        res = c_abi_rccl_check(index, batches[0], params, world, device, coll_device)
        collective["c_abi_rccl"] = res
        if rank == 0:
            log(f"C-ABI RCCL all-reduce of the counters over {world} ranks (dcn_comm_* / dcn_stats_allreduce_rccl): {json.dumps(res, default=str)}")
            try:
                if args.detail and os.path.exists(args.detail):
                    det = json.load(open(args.detail))
                    det.setdefault("collective", {})["c_abi_rccl"] = res
                    if res and "matches_torch_all_reduce" in res:
                        det["collective"]["c_abi_rccl_matches"] = res["matches_torch_all_reduce"]
                    with open(args.detail, "w") as f:
                        json.dump(det, f, indent=1, default=str)
            except Exception as ex:  # noqa: BLE001
                log(f"could not add the C-ABI RCCL result to {args.detail}: {ex!r}")
    if world > 1:
        if isinstance(collective.get("c_abi_rccl"), dict) and collective["c_abi_rccl"].get("timed_out"):  # a helper thread is still inside RCCL: no further collective,
            sys.stdout.flush()                                      # no orderly teardown around it
            sys.stderr.flush()
            os._exit(0)
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
