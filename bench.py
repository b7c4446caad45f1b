#!/usr/bin/env python3
"""Headline benchmark: Mbp/s filtered (k=31, w=15) against a panhuman-1-sized index on N MI355X.

  python bench.py --gpus N --steps K --warmup W
  (N > 1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

One "step" = one pass of the hot path (pack -> plan -> scan/hash/probe/distinct -> finish) over one batch of
synthetic 150 bp reads that is already resident in HBM as ASCII + offsets (BASELINE.json configs[1]).  The index is
a device table of 409,913,780 synthetic u64 keys (panhuman-1's size): the minimizers of a synthetic "host" genome
plus uniform random keys; half of the reads are drawn from the host genome (0.5 % substitutions, 0.1 % N), half
are random.  Weak scaling: every rank holds a full index replica and filters its own batch; the only collective
is the all-reduce of the six summary counters (RCCL) at the end of the timed region.

Prints ONE JSON line on rank 0 (see the task contract): value = whole-job Mbp/s, plus
  roofline      dominant kernel (scan) : algorithmic HBM bytes / HIP-event time on the kernel's own stream
  cpu_baseline  the CPU oracle (oracle/, "port") on a bounded sample of the same reads, all host cores
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch  # first: libdeacon_hip.so must bind to the HIP runtime torch already loaded
import torch.distributed as dist

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import deacon_server_amd as dcn  # noqa: E402

K, W = 31, 15
READ_LEN = 150
PANHUMAN_KEYS = 409_913_780  # README.md:52 of the reference
HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
SCATTER_CEILING = 46e9  # random 16-byte reads/s of a 2^31-slot table on MI355X, measured (profiles/r01_probe_patterns_17GB.txt)


def log(*a):
    if int(os.environ.get("RANK", "0")) == 0:
        print("[bench]", *a, file=sys.stderr, flush=True)


def make_host_genome(n, seed, device):
    g = torch.Generator(device=device)
    g.manual_seed(seed)
    alpha = torch.tensor(list(b"ACGT"), dtype=torch.uint8, device=device)
    return alpha[torch.randint(0, 4, (n,), generator=g, device=device)]


def host_minimizer_keys(genome_dev, index_k, index_w):
    """Minimizer hashes of the host genome, computed by the product path itself (dump seam): for ACGT-only
    sequence the index-side and filter-side rules coincide."""
    genome = genome_dev.cpu().numpy()
    seg, ov = 1 << 20, index_k + index_w - 2
    tmp_idx = dcn.Index.from_keys(np.arange(1, 3, dtype=np.uint64), index_k, index_w, device=genome_dev.device.index)
    proc = dcn.FilterProcessor(tmp_idx, max_batch_bases=(seg + ov) * 4, max_batch_reads=8)
    keys = []
    starts = list(range(0, len(genome), seg))
    for i in range(0, len(starts), 4):
        reads = [genome[s:min(len(genome), s + seg + ov)] for s in starts[i:i + 4]]
        offsets = np.zeros(len(reads) + 1, np.uint64)
        np.cumsum([len(r) for r in reads], out=offsets[1:])
        _, h, _ = proc.minimizer_hashes_batch(np.concatenate(reads), offsets)
        keys.append(np.unique(h))
    proc.close()
    tmp_idx.close()
    return np.unique(np.concatenate(keys))


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


def cpu_baseline(keys, bases_np, n_reads_total, params, want_keep_dev, seconds_target=15.0):
    """Time the CPU oracle (all host cores) on a bounded sample of the same reads and check the GPU's decisions
    on that sample against it."""
    from oracle import oracle as O
    cores = host_cores()
    t0 = time.time()
    oidx = O.Index(keys, K, W, threads=cores)
    log(f"cpu_baseline: built the {len(oidx):,}-key CPU set with {cores} threads in {time.time() - t0:.1f} s")

    def run(n, repeats=1):
        off = np.arange(n + 1, dtype=np.uint64) * np.uint64(READ_LEN)
        t = time.time()
        for _ in range(repeats):
            res = O.filter_batch(oidx, bases_np[:n * READ_LEN], off, None, params["abs"], params["rel"], 0,
                                 params["deplete"], threads=cores)
        return time.time() - t, res

    probe_n = min(10_000 * cores, n_reads_total)
    dt, _ = run(probe_n)
    rate = probe_n / max(dt, 1e-6)
    n = int(min(n_reads_total, max(probe_n, rate * seconds_target)))
    repeats = max(1, int(round(rate * seconds_target / n)))  # the sample is cycled until ~seconds_target of CPU work
    dt, (keep, hits, total) = run(n, repeats)
    ok = bool((want_keep_dev[:n].cpu().numpy().astype(bool) == keep).all())
    return {
        "value": n * repeats * READ_LEN / dt / 1e6, "unit": "Mbp/s", "cores": cores, "kind": "port",
        "sample": f"first {n} reads of the rank-0 batch x{repeats} passes ({n * repeats * READ_LEN / 1e9:.2f} Gbp, "
                  f"{dt:.1f} s), oracle/ C restatement with a pthread pool over {cores} threads, same "
                  f"{len(oidx):,}-key index",
        "decisions_match_gpu": ok,
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--reads", type=int, default=4_000_000, help="reads per batch per GPU (150 bp each)")
    ap.add_argument("--index-keys", type=int, default=PANHUMAN_KEYS)
    ap.add_argument("--host-genome", type=int, default=64_000_000,
                    help="bases of the synthetic host genome (SURVEY.md 8d config 2: 64 Mbp, ~8 M of the index keys)")
    ap.add_argument("--contexts", type=int, default=1,
                    help="pipeline contexts per GPU: the step's batch is split into this many sub-batches, each on its "
                         "own context/stream, so pack+plan of one overlap the scan of another")
    ap.add_argument("--workload", choices=["short", "paired", "long"], default="short",
                    help="short: configs[1] 150 bp single reads (the headline); paired: configs[3] 2x150 bp --deplete; "
                         "long: configs[2] ONT-style lognormal reads, mean 10 kbp")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            sys.exit("bench.py --gpus N with N > 1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    torch.cuda.set_device(local_rank)
    device = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group(backend="nccl", device_id=device)

    # ---- index: host-genome minimizers + uniform random keys, identical on every rank -----------------------
    t0 = time.time()
    genome_dev = make_host_genome(args.host_genome, 3, device)
    host_keys = host_minimizer_keys(genome_dev, K, W)
    rng = np.random.default_rng(4)
    n_rand = max(0, args.index_keys - len(host_keys))
    keys = np.empty(len(host_keys) + n_rand, np.uint64)
    keys[:len(host_keys)] = host_keys
    keys[len(host_keys):] = rng.integers(1, 2**63, n_rand, dtype=np.uint64) * np.uint64(2) + np.uint64(1)
    log(f"index keys: {len(host_keys):,} host + {n_rand:,} random in {time.time() - t0:.1f} s")
    t0 = time.time()
    index = dcn.Index.from_keys(keys, K, W, device=local_rank)
    index_build_s = time.time() - t0
    log(f"device table: {index.n_keys:,} distinct keys in {index_build_s:.1f} s")

    # ---- reads: resident in HBM before the timed region -------------------------------------------------------
    params = {"abs": 2, "rel": 0.01, "deplete": False}
    d_unit_id = None
    if args.workload == "short":
        n_reads = args.reads
        d_bases = make_reads(genome_dev, n_reads, 5 + rank, device)
        d_offsets = torch.arange(n_reads + 1, dtype=torch.int64, device=device) * READ_LEN
        n_units = n_reads
    elif args.workload == "paired":
        n_reads = args.reads // 2 * 2
        d_bases = make_pairs(genome_dev, n_reads // 2, 7 + rank, device)
        d_offsets = torch.arange(n_reads + 1, dtype=torch.int64, device=device) * READ_LEN
        d_unit_id = (torch.arange(n_reads, dtype=torch.int32, device=device) // 2).contiguous()
        n_units = n_reads // 2
        params["deplete"] = True
    else:
        d_bases, d_offsets = make_long_reads(genome_dev, args.reads * READ_LEN, 6 + rank, device)
        n_reads = d_offsets.numel() - 1
        n_units = n_reads
    n_bases = int(d_bases.numel())
    d_keep = torch.zeros(n_units, dtype=torch.uint8, device=device)
    d_hits = torch.zeros(n_units, dtype=torch.int32, device=device)
    d_total = torch.zeros(n_units, dtype=torch.int32, device=device)
    C = max(1, args.contexts) if args.workload == "short" else 1
    bounds = [(n_reads * c // C) // 8 * 8 for c in range(C)] + [n_reads]  # sub-batch starts stay 16-byte aligned
    if C == 1:
        procs = [dcn.FilterProcessor(index, abs_threshold=params["abs"], rel_threshold=params["rel"],
                                     deplete=params["deplete"], max_batch_bases=n_bases, max_batch_reads=n_reads)]
        if args.workload == "long":
            procs[0].reserve_records(n_bases // 6)  # ~1 hit record per 16 bp when half the reads are host-derived
    else:
        procs = [dcn.FilterProcessor(index, abs_threshold=params["abs"], rel_threshold=params["rel"],
                                     deplete=params["deplete"], max_batch_bases=(bounds[c + 1] - bounds[c]) * READ_LEN,
                                     max_batch_reads=bounds[c + 1] - bounds[c]) for c in range(C)]
    d_sub_offsets = [torch.arange(bounds[c + 1] - bounds[c] + 1, dtype=torch.int64, device=device) * READ_LEN
                     for c in range(C)] if C > 1 else None
    torch.cuda.synchronize()

    def step(counts=True, keep_ptr=None):
        # counts=True: keep + exact distinct-hit count + minimizer total per unit (the headline measurement);
        # counts=False: decisions only, as `deacon filter` consumes them (lanes stop once a decision is fixed)
        kp = d_keep.data_ptr() if keep_ptr is None else keep_ptr
        if C == 1:
            procs[0].filter_batch_device(d_bases.data_ptr(), d_offsets.data_ptr(), n_reads, n_bases, kp,
                                         d_hits.data_ptr() if counts else None, d_total.data_ptr() if counts else None,
                                         d_unit_id=d_unit_id.data_ptr() if d_unit_id is not None else None,
                                         n_units=n_units)
            return
        for c, proc in enumerate(procs):
            a, b = bounds[c], bounds[c + 1]
            proc.filter_batch_device(d_bases.data_ptr() + a * READ_LEN, d_sub_offsets[c].data_ptr(), b - a,
                                     (b - a) * READ_LEN, kp + a, d_hits.data_ptr() + 4 * a if counts else None,
                                     d_total.data_ptr() + 4 * a if counts else None)

    def sync_all():
        for proc in procs:
            proc.synchronize()

    for _ in range(args.warmup):
        step()
    sync_all()
    for proc in procs:
        proc.reset_stats()
        proc.set_profiling(True)

    # ---- timed region: exactly K steps, barrier + device sync on both sides ------------------------------------------
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t_start = time.perf_counter()
    for _ in range(args.steps):
        step()
    sync_all()
    local = {n: sum(p.stats()[n] for p in procs) for n in dcn._native.STAT_NAMES}
    # RCCL all-reduce of the six counters: the path's only collective
    counters = dcn.distributed.allreduce_counters(local, device=device)
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed = time.perf_counter() - t_start
    t = torch.tensor([elapsed], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed = float(t.item())
    counters = [counters[n] for n in dcn._native.STAT_NAMES]

    stage_ms, n_prof = {n: 0.0 for n in dcn._native.STAGE_NAMES}, 0
    for proc in procs:
        ms, nb = proc.profile()
        n_prof += nb
        for n in ms:
            stage_ms[n] += ms[n]
        proc.set_profiling(False)
    n_minimizers = int(d_total.sum(dtype=torch.int64).item())
    kept = int(d_keep.sum(dtype=torch.int64).item())

    # ---- second measurement, outside the contract's timed region: the same K steps asking for decisions only ------
    d_keep2 = torch.zeros_like(d_keep)
    for proc in procs:
        proc.set_profiling(True)
    step(False, d_keep2.data_ptr())
    sync_all()
    for proc in procs:
        proc.profile()  # drop the warm-up launch from the stage timers
        proc.set_profiling(True)
    if world > 1:
        dist.barrier()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    for _ in range(args.steps):
        step(False, d_keep2.data_ptr())
    sync_all()
    torch.cuda.synchronize()
    if world > 1:
        dist.barrier()
    elapsed2 = time.perf_counter() - t2
    t = torch.tensor([elapsed2], dtype=torch.float64, device=device)
    if world > 1:
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
    elapsed2 = float(t.item())
    scan2_ms, n2 = 0.0, 0
    for proc in procs:
        ms, nb = proc.profile()
        scan2_ms += ms["scan"]
        n2 += nb
        proc.set_profiling(False)
    same_decisions = bool(torch.equal(d_keep, d_keep2))

    if rank == 0:
        total_bp = counters[2]
        if args.workload != "long" or world == 1:  # long reads: every rank draws its own lengths
            assert total_bp == n_bases * args.steps * world, (total_bp, n_bases, args.steps, world)
        scan_ms = stage_ms["scan"] / max(n_prof, 1)
        # per scan launch (= per sub-batch): SURVEY.md 8d: 2-bit base + mask bit, 8 B per probed minimizer
        algo_bytes = (0.375 * n_bases + 8.0 * n_minimizers) / C
        achieved = algo_bytes / (scan_ms * 1e-3) / 1e9
        traffic = None  # HBM bytes per scan launch from the committed PMC passes (same workload only)
        try:
            tr = json.load(open(os.path.join(ROOT, "profiles", "r01_traffic.json")))
            wlk = tr["workload"]
            if (args.workload == wlk["workload"] and n_reads == wlk["reads_per_batch"] and C == 1
                    and int(index.n_keys) == wlk["index_keys"] and args.host_genome == wlk.get("host_genome_bases")):
                traffic = tr["scan_kernel"]["hbm_bytes_per_launch"]
        except Exception:
            pass
        out = {
            "metric": "Mbp/s filtered (k=31,w=15 vs panhuman-1-sized index), decisions bit-exact vs CPU",
            "value": total_bp / elapsed / 1e6,
            "unit": "Mbp/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {
                "workload": {"short": "configs[1]: 150 bp reads vs panhuman-1-sized index, -a 2 -r 0.01, inputs resident in HBM as ASCII",
                             "paired": "configs[3]: paired 2x150 bp --deplete vs panhuman-1-sized index, inputs resident in HBM as ASCII",
                             "long": "configs[2]: ONT-style lognormal reads (mean 10 kbp) vs panhuman-1-sized index, -a 2 -r 0.01, inputs resident in HBM as ASCII"}[args.workload],
                "index_keys": int(index.n_keys), "reads_per_batch_per_gpu": n_reads, "bases_per_batch_per_gpu": n_bases,
                "k": K, "w": W, "host_fraction": 0.5, "host_genome_bases": args.host_genome, "parallelism": f"reads sharded x{world}, index replicated",
                "contexts_per_gpu": C,
            },
            "roofline": {
                "bound": "hbm", "kernel": "scan_kernel<15> (scan+hash+probe+distinct)",
                "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                "traffic": traffic,
                # measured HBM traffic (PMC, profiles/r01_traffic.json) over the live launch time: the rate the probe
                # kernel actually pulls from HBM, against the same 8 TB/s
                "traffic_rate_GBps": (traffic / (scan_ms * 1e-3) / 1e9) if traffic else None,
                "traffic_frac_of_peak": (traffic / (scan_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic else None,
                "algorithmic_bytes_per_launch": algo_bytes, "avg_launch_ms": scan_ms,
                "minimizers_per_launch": n_minimizers // C,
                # second ceiling, reported beside the contract's: every minimizer is one scattered 16-byte
                # request, and the chip serves 46 G of those per second from a 17 GB table however they are
                # issued (profiles/r01_probe_patterns_17GB.txt; 56 G/s from a 134 MB table, _small.txt)
                "scattered_probes_per_s": n_minimizers / C / (scan_ms * 1e-3),
                "scattered_ceiling_per_s": SCATTER_CEILING,
                "frac_of_scattered_ceiling": n_minimizers / C / (scan_ms * 1e-3) / SCATTER_CEILING,
            },
            "stage_ms_per_launch": {k_: v / max(n_prof, 1) for k_, v in stage_ms.items()},
            "kept_fraction": kept / n_units,
            # not the headline: same batch, same K steps, caller passes no hits/total arrays (what the CLI does outside
            # --debug); reads whose decision is fixed after abs_threshold distinct hits are not probed further
            "decisions_only": {"value": n_bases * args.steps * world / elapsed2 / 1e6, "unit": "Mbp/s",
                               "ms_per_step": elapsed2 / args.steps * 1e3, "scan_ms_per_launch": scan2_ms / max(n2, 1),
                               "decisions_identical_to_counting_mode": same_decisions},
            "index_build_s": index_build_s,
        }
        if world == 1 and not args.no_cpu_baseline and args.workload == "short":
            bases_np = d_bases[:min(n_reads, 2_000_000) * READ_LEN].cpu().numpy()
            out["cpu_baseline"] = cpu_baseline(keys, bases_np, min(n_reads, 2_000_000), params, d_keep)
        else:
            out["cpu_baseline"] = None
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
