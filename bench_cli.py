"""File-to-file legs of bench.py (`cli` key of the driver-run line): `deacon-hip filter` / `deacon-hip index build` as a
user runs them -- FASTQ on disk in, FASTQ on disk out -- timed by the tool's own clock and checked against the CPU oracle.

  plumbing    SURVEY.md 8d config 1 (= BASELINE.json configs[0]) at its stated shape: a 4,641,652 bp uniform ACGT genome
              (seed 1, 80-column FASTA) -> `deacon-hip index build` -> 10,000 x 150 bp FASTQ (seed 2: 5,000 sampled
              from the genome on a random strand with 0.5 % substitutions, 5,000 uniform random), `-a 2 -r 0.01`, search
              and `-d`.  Checked in full: index key set == the oracle's index-side builder, kept ids == the oracle's
              decisions for all 10,000 reads in both modes.
  search50, deplete95, paired
              150 bp reads against a panhuman-1-sized index FILE (the reference's 3.7 GB bincode format, written by
              `dcn_index_write_file`): half of the reads from the host genome in search mode (half of the input is written
              back), 95 % from the host with `-d` (the shape of a host-depletion run), and two files of mates with
              `-d -O`.  Two rates each, as the reference defines them: bases / wall time of run() including the index load
              (src/local_filter.rs:726-729, the "Speed" line) and with the load subtracted (the spinner's rate,
              :312-315); plus busy core-seconds per stage (DCN_CLI_TIMING) and the ids of the kept records among the
              first reads of the file against the oracle.

The reads are generated on the GPU (the bench process already holds the genome there) and written to tmpfs; the tool runs
as a child process with its own index replica on the same device.  Nothing here enters bench.py's `value`.
"""
import json
import os
import re
import shutil
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
BIN = os.path.join(ROOT, "deacon-server_amd", "bin", "deacon-hip")
READ_LEN = 150
ID_DIGITS = 9
REC = 1 + ID_DIGITS + 1 + READ_LEN + 1 + 2 + READ_LEN + 1  # "@%09d\n" seq "\n+\n" qual "\n" = 315 bytes

ECOLI_BASES = 4_641_652  # E. coli K-12 MG1655


def tmp_root():
    return tempfile.mkdtemp(prefix="dcn_cli_", dir="/dev/shm" if os.path.isdir("/dev/shm") else None)


def _comp(a):
    t = np.arange(256, dtype=np.uint8)
    for x, y in zip(b"ACGT", b"TGCA"):
        t[x] = y
    return t[a]


def fastq_records(seqs, first_id=0):
    """(n, 150) uint8 -> (n, 315) uint8 FASTQ records '@000000123\\nSEQ\\n+\\nIII...\\n'"""
    n = seqs.shape[0]
    rec = np.empty((n, REC), np.uint8)
    ids = np.arange(first_id, first_id + n, dtype=np.int64)
    rec[:, 0] = ord("@")
    for d in range(ID_DIGITS):
        rec[:, ID_DIGITS - d] = ord("0") + (ids // 10 ** d) % 10
    rec[:, 1 + ID_DIGITS] = 10
    rec[:, 2 + ID_DIGITS:2 + ID_DIGITS + READ_LEN] = seqs
    o = 2 + ID_DIGITS + READ_LEN
    rec[:, o] = 10
    rec[:, o + 1] = ord("+")
    rec[:, o + 2] = 10
    rec[:, o + 3:o + 3 + READ_LEN] = ord("I")
    rec[:, o + 3 + READ_LEN] = 10
    return rec


def ids_of_output(path, limit_id=None):
    """ids of a FASTQ file whose records all have the 315-byte shape above (the tool copies such records unchanged)"""
    size = os.path.getsize(path)
    assert size % REC == 0, f"{path}: {size} bytes is not a whole number of {REC}-byte records"
    n = size // REC
    if n == 0:
        return np.zeros(0, np.int64)
    mm = np.memmap(path, dtype=np.uint8, mode="r", shape=(n, REC))
    take = n
    if limit_id is not None:  # output keeps the input order: the ids below limit_id are a prefix
        take = min(n, limit_id)
    digits = np.asarray(mm[:take, 1:1 + ID_DIGITS]).astype(np.int64) - ord("0")
    ids = (digits * (10 ** np.arange(ID_DIGITS - 1, -1, -1, dtype=np.int64))[None, :]).sum(1)
    assert (np.asarray(mm[:take, 0]) == ord("@")).all()
    if limit_id is not None:
        ids = ids[ids < limit_id]
    return ids


def run_tool(args, timing=True):
    env = dict(os.environ)
    if timing:
        env["DCN_CLI_TIMING"] = "1"
    t = time.perf_counter()
    p = subprocess.run([BIN, *map(str, args)], capture_output=True, text=True, env=env)
    wall = time.perf_counter() - t
    if p.returncode != 0:
        raise RuntimeError(f"deacon-hip {' '.join(map(str, args))} failed ({p.returncode}): {p.stderr[-2000:]}")
    return p, wall


def parse_timing(stderr):
    out = {}
    m = re.search(r"timing: wall ([0-9.]+) s; busy seconds: parse ([0-9.]+) \(all workers\), GPU stage ([0-9.]+) "
                  r"\(main thread waited ([0-9.]+) for it\), format ([0-9.]+) \(all workers\), write ([0-9.]+)", stderr)
    if m:
        out["run_wall_s"] = float(m.group(1))
        out["busy_core_s"] = {"parse": float(m.group(2)), "gpu_stage": float(m.group(3)), "format": float(m.group(5)),
                              "write": float(m.group(6))}
    m = re.search(r"timing: process CPU ([0-9.]+) s user \+ ([0-9.]+) s system", stderr)
    if m:
        out["process_cpu_s"] = {"user": float(m.group(1)), "system": float(m.group(2))}
    m = re.search(r"index loaded ([0-9.]+), contexts ready ([0-9.]+), all input parsed\+queued ([0-9.]+), GPU stage drained ([0-9.]+), "
                  r"all written ([0-9.]+)", stderr)
    if m:
        out["milestones_s"] = {"index_loaded": float(m.group(1)), "contexts_ready": float(m.group(2)),
                               "input_parsed_and_queued": float(m.group(3)), "gpu_stage_drained": float(m.group(4)),
                               "all_written": float(m.group(5))}
    return out


def filter_run(idx_path, inputs, out_paths, extra, summary_path):
    """one `deacon-hip filter` run -> rates as the reference defines them + stage accounting"""
    args = ["filter", idx_path, *inputs, "-o", out_paths[0]]
    if len(out_paths) > 1:
        args += ["-O", out_paths[1]]
    args += ["-s", summary_path, "-q", *extra]
    p, wall = run_tool(args)
    s = json.load(open(summary_path))
    t = parse_timing(p.stderr)
    run_wall = t.get("run_wall_s", s["time"])
    load = t.get("milestones_s", {}).get("index_loaded", 0.0)
    res = {
        "args": " ".join(["-q", *extra]) + (" -O" if len(out_paths) > 1 else ""),
        "seqs_in": s["seqs_in"], "seqs_out": s["seqs_out"], "bp_in": s["bp_in"], "bp_out": s["bp_out"],
        "bytes_in": sum(os.path.getsize(x) for x in inputs), "bytes_out": sum(os.path.getsize(x) for x in out_paths),
        "process_wall_s": wall, "run_wall_s": run_wall, "index_load_s": load,
        # src/local_filter.rs:726-729: bases / wall time of run(), index load included
        "Mbp_per_s_incl_index_load": s["bp_in"] / run_wall / 1e6,
        # src/local_filter.rs:312-315: the spinner's clock starts after the index is loaded
        "Mbp_per_s_filter_only": s["bp_in"] / max(run_wall - load, 1e-9) / 1e6,
    }
    res.update({k_: v for k_, v in t.items() if k_ != "run_wall_s"})
    return res


# ---- (a) plumbing: SURVEY.md 8d config 1 at its stated shape -------------------------------------------------------------
def gzip_one_member(src, dst, nbytes, threads, level=1, piece=16 << 20):
    """the first nbytes of src as ONE gzip member (one deflate stream, what `gzip` / `pigz` write -- not BGZF), compressed
    piece by piece on `threads` threads the way pigz does it: every piece but the last ends in a full flush (an empty stored
    block at a byte boundary), the raw pieces follow each other, the trailer carries the CRC-32 of the whole"""
    import mmap
    import struct
    import zlib
    from concurrent.futures import ThreadPoolExecutor
    with open(src, "rb") as f, mmap.mmap(f.fileno(), 0, access=mmap.ACCESS_READ) as mm:
        view = memoryview(mm)[:nbytes]
        spans = [(a, min(a + piece, nbytes)) for a in range(0, nbytes, piece)]

        def deflate(i):
            a, b = spans[i]
            c = zlib.compressobj(level, zlib.DEFLATED, -15)
            return c.compress(view[a:b]) + c.flush(zlib.Z_FINISH if i == len(spans) - 1 else zlib.Z_FULL_FLUSH)

        def crc_all():
            crc = 0
            for a, b in spans:
                crc = zlib.crc32(view[a:b], crc)
            return crc
        with ThreadPoolExecutor(max(2, threads)) as ex:  # (zlib releases the GIL)
            crc_f = ex.submit(crc_all)
            parts = list(ex.map(deflate, range(len(spans))))
            crc = crc_f.result()
        with open(dst, "wb") as o:
            o.write(b"\x1f\x8b\x08\x00\x00\x00\x00\x00\x00\xff")
            for p_ in parts:
                o.write(p_)
            o.write(struct.pack("<II", crc & 0xFFFFFFFF, nbytes & 0xFFFFFFFF))
        del view


def ecoli_shaped_inputs(d, n_reads=10_000):
    """genome (seed 1) as 80-column FASTA, reads (seed 2) as FASTQ; returns (genome bytes, (n, 150) reads, paths)"""
    alpha = np.frombuffer(b"ACGT", dtype=np.uint8)
    genome = alpha[np.random.default_rng(1).integers(0, 4, ECOLI_BASES)]
    fa = os.path.join(d, "ecoli_shaped.fa")
    with open(fa, "wb") as f:
        f.write(b">synthetic_K12 4641652 bp uniform ACGT seed 1\n")
        body = genome[:ECOLI_BASES // 80 * 80].reshape(-1, 80)
        lines = np.empty((body.shape[0], 81), np.uint8)
        lines[:, :80] = body
        lines[:, 80] = 10
        f.write(lines.tobytes())
        f.write(genome[ECOLI_BASES // 80 * 80:].tobytes() + b"\n")
    rng = np.random.default_rng(2)
    half = n_reads // 2
    starts = rng.integers(0, ECOLI_BASES - READ_LEN, half)
    host = genome[starts[:, None] + np.arange(READ_LEN)[None, :]]
    strand = rng.random(half) < 0.5
    host = np.where(strand[:, None], _comp(host)[:, ::-1], host)
    sub = rng.random(host.shape) < 0.005
    host = np.where(sub, alpha[rng.integers(0, 4, host.shape)], host)
    rnd = alpha[rng.integers(0, 4, (n_reads - half, READ_LEN))]
    reads = np.concatenate([host, rnd]).astype(np.uint8)
    reads = reads[rng.permutation(n_reads)]
    fq = os.path.join(d, "reads_10k.fq")
    fastq_records(reads).tofile(fq)
    return genome.tobytes(), reads, fa, fq


def plumbing(d=None, threads=1):
    """configs[0]: index build + search + deplete through the tool, every result compared with the oracle"""
    from oracle import oracle as O
    own = d is None
    d = d or tmp_root()
    try:
        genome, reads, fa, fq = ecoli_shaped_inputs(d)
        idx_path = os.path.join(d, "ecoli_shaped.idx")
        t0 = time.perf_counter()
        p, wall_build = run_tool(["index", "build", fa, "-o", idx_path, "-q"], timing=False)
        t_or = time.perf_counter()
        oidx = O.Index.build([genome], 31, 15)
        oracle_build_s = time.perf_counter() - t_or
        got = O.Index.read(idx_path)
        keys_equal = (got.k, got.w) == (31, 15) and len(got) == len(oidx) and \
            bool((np.sort(got.keys()) == np.sort(oidx.keys())).all())
        bases = reads.reshape(-1)
        off = np.arange(len(reads) + 1, dtype=np.uint64) * np.uint64(READ_LEN)
        out = {"workload": f"SURVEY 8d config 1: {ECOLI_BASES:,} bp genome (seed 1) -> index build -> {len(reads):,} x "
                           f"{READ_LEN} bp FASTQ (seed 2), -a 2 -r 0.01, search and -d",
               "index_build": {"process_wall_s": wall_build, "Mbp_per_s": ECOLI_BASES / wall_build / 1e6,
                               "keys": len(got), "oracle_keys": len(oidx), "oracle_build_s": oracle_build_s,
                               "key_set_equals_oracle": keys_equal}}
        ok = keys_equal
        for mode, extra, deplete in (("search", [], False), ("deplete", ["-d"], True)):
            o = os.path.join(d, f"plumb_{mode}.fq")
            r = filter_run(idx_path, [fq], [o], extra, os.path.join(d, "s.json"))
            keep, _, _ = O.filter_batch(oidx, bases, off, None, 2, 0.01, 0, deplete, threads=threads)
            want = np.nonzero(keep)[0]
            ids = ids_of_output(o)
            r["kept_ids_equal_oracle"] = bool(len(ids) == len(want) and (ids == want).all())
            r["checked"] = f"all {len(reads)} reads"
            ok = ok and r["kept_ids_equal_oracle"] and r["seqs_out"] == len(want)
            out[mode] = r
        out["decisions_match"] = bool(ok)
        return out
    finally:
        if own:
            shutil.rmtree(d, ignore_errors=True)


# ---- (b) file to file against a panhuman-sized index file --------------------------------------------------------------------
def write_fastq_from_device(path, make_chunk, n_reads, chunk=4_000_000, first_id=0):
    """make_chunk(a, m) -> (m, 150) uint8 device tensor; records are formatted on the device and appended to `path`"""
    import torch
    with open(path, "wb") as f:
        for a in range(0, n_reads, chunk):
            m = min(chunk, n_reads - a)
            seqs = make_chunk(a, m)
            dev = seqs.device
            rec = torch.empty((m, REC), dtype=torch.uint8, device=dev)
            ids = torch.arange(first_id + a, first_id + a + m, dtype=torch.int64, device=dev)
            rec[:, 0] = ord("@")
            for dg in range(ID_DIGITS):
                rec[:, ID_DIGITS - dg] = (ord("0") + (ids // 10 ** dg) % 10).to(torch.uint8)
            rec[:, 1 + ID_DIGITS] = 10
            rec[:, 2 + ID_DIGITS:2 + ID_DIGITS + READ_LEN] = seqs
            o = 2 + ID_DIGITS + READ_LEN
            rec[:, o] = 10
            rec[:, o + 1] = ord("+")
            rec[:, o + 2] = 10
            rec[:, o + 3:o + 3 + READ_LEN] = ord("I")
            rec[:, o + 3 + READ_LEN] = 10
            rec.cpu().numpy().tofile(f)
            del rec, seqs


def file_to_file(index, genome_dev, host_keys_sorted, n_rand, make_reads, make_pairs, touchable_oracle_index, cores,
                 sizes=None, log=lambda *a: None, check_reads=200_000):
    """search50 / deplete95 / paired against the bench's panhuman-sized index written as an index FILE"""
    from oracle import oracle as O
    sizes = sizes or {"search50": 16_000_000, "deplete95": 32_000_000, "paired": 8_000_000}
    d = tmp_root()
    out = {}
    try:
        idx_path = os.path.join(d, "panhuman_sized.idx")
        t = time.perf_counter()
        index.write(idx_path)
        out["index_file"] = {"keys": int(index.n_keys), "bytes": os.path.getsize(idx_path), "write_s": time.perf_counter() - t,
                             "format": "bincode-2 varint, src/index.rs:130-164"}
        log(f"cli: index file of {out['index_file']['bytes'] / 1e9:.2f} GB written in {out['index_file']['write_s']:.1f} s")
        device = genome_dev.device
        # the first process to load the file is not one of the timed legs: the time a fresh process takes to get the 34 GB
        # table up was seen to vary from 0.28 to 1.26 s on the same files (box state, not the tool); it is reported beside them
        warm_fq = os.path.join(d, "warm.fq")
        fastq_records(np.full((1000, READ_LEN), ord("A"), np.uint8)).tofile(warm_fq)
        w = filter_run(idx_path, [warm_fq], [os.path.join(d, "warm.out.fq")], [], os.path.join(d, "s.json"))
        out["index_file"]["first_load_s"] = w["index_load_s"]
        out["index_file"]["first_run_wall_s"] = w["run_wall_s"]

        def check(inputs_seqs, uid, deplete, out_path, n_units):
            """kept ids among the first reads of the input vs the oracle on the index keys that sample can touch"""
            bases = inputs_seqs.reshape(-1)
            off = np.arange(inputs_seqs.shape[0] + 1, dtype=np.uint64) * np.uint64(READ_LEN)
            small = touchable_oracle_index(bases, off, host_keys_sorted, n_rand, cores)
            keep, _, _ = O.filter_batch(small, bases, off, uid, 2, 0.01, 0, deplete, threads=cores)
            want = np.nonzero(keep)[0]
            ids = ids_of_output(out_path, limit_id=n_units)
            return bool(len(ids) == len(want) and (ids == want).all())

        for name, host_frac, extra, deplete in (("search50", 0.5, [], False), ("deplete95", 0.95, ["-d"], True)):
            n = sizes[name]
            if not n:
                continue
            fq = os.path.join(d, f"{name}.fq")
            t = time.perf_counter()
            first = {}

            def chunk(a, m, _seed=31 if name == "search50" else 37):
                x = make_reads(genome_dev, m, _seed + a, device, host_frac=host_frac).reshape(m, READ_LEN)
                if a == 0:
                    first["seqs"] = x[:check_reads].cpu().numpy()
                return x
            write_fastq_from_device(fq, chunk, n)
            gen_s = time.perf_counter() - t
            o = os.path.join(d, f"{name}.out.fq")
            r = filter_run(idx_path, [fq], [o], extra, os.path.join(d, "s.json"))
            r["workload"] = (f"{n:,} x {READ_LEN} bp FASTQ on tmpfs, {int(host_frac * 100)} % of the reads from the host genome, "
                             f"-a 2 -r 0.01{' -d' if deplete else ''}, panhuman-sized index file")
            nchk = min(check_reads, n)
            r["decisions_match"] = check(first["seqs"][:nchk], None, deplete, o, nchk)
            r["checked"] = f"ids of the kept records among the first {nchk} reads == the oracle's decisions"
            r["input_generated_s"] = gen_s
            if cores > 8 and os.environ.get("DCN_BENCH_CLI_THREADS_AB"):  # the same files with three quarters of the CPU share (the tool's default is the reference's: -t 8; the
                # formatters, the library's copy threads and the runtime want the rest: -t 12 beat -t 8 and -t 16 on 16 CPUs)
                o2 = os.path.join(d, f"{name}.out_t.fq")
                nt = max(9, cores * 3 // 4)
                rt = filter_run(idx_path, [fq], [o2], extra + ["-t", str(nt)], os.path.join(d, "s.json"))
                rt["decisions_match"] = check(first["seqs"][:nchk], None, deplete, o2, nchk)
                r[f"threads_{nt}"] = {k_: rt[k_] for k_ in ("args", "run_wall_s", "index_load_s", "Mbp_per_s_incl_index_load",
                                                               "Mbp_per_s_filter_only", "busy_core_s", "decisions_match") if k_ in rt}
                r["decisions_match"] = r["decisions_match"] and rt["decisions_match"]
                os.unlink(o2)
            n_gz = min(n, (sizes.get("deplete95_gz", 8_000_000) if name == "deplete95" else 0))
            if n_gz:
                # the same reads from ONE gzip stream (what a .fastq.gz from a sequencing run is; not BGZF): the tool's own inflate
                # on several threads (cli/parallel_gzip.hpp) in front of the same pipeline
                gz, o2 = fq + ".gz", os.path.join(d, f"{name}.out_gz.fq")
                t = time.perf_counter()
                gzip_one_member(fq, gz, n_gz * REC, cores)
                gz_s = time.perf_counter() - t
                rg = filter_run(idx_path, [gz], [o2], extra, os.path.join(d, "s.json"))
                rg["workload"] = (f"the first {n_gz:,} reads of deplete95 as one gzip member ({os.path.getsize(gz) / 1e9:.2f} GB, level 1, "
                                  f"compressed in {gz_s:.1f} s), same arguments")
                nchk_g = min(check_reads, n_gz)
                rg["decisions_match"] = check(first["seqs"][:nchk_g], None, deplete, o2, nchk_g)
                rg["checked"] = r["checked"]
                out[name + "_gz"] = rg
                log(f"cli.{name}_gz: run() {rg['run_wall_s']:.2f} s, {rg['Mbp_per_s_incl_index_load'] / 1e3:.2f} Gbp/s incl. index load, "
                    f"{rg['Mbp_per_s_filter_only'] / 1e3:.2f} Gbp/s filter only, oracle ok={rg['decisions_match']}")
                os.unlink(gz)
                os.unlink(o2)
            out[name] = r
            log(f"cli.{name}: run() {r['run_wall_s']:.2f} s, {r['Mbp_per_s_incl_index_load'] / 1e3:.2f} Gbp/s incl. index load "
                f"({r['index_load_s']:.2f} s), {r['Mbp_per_s_filter_only'] / 1e3:.2f} Gbp/s filter only, kept {r['seqs_out']}/{r['seqs_in']}, "
                f"oracle ok={r['decisions_match']}")
            os.unlink(fq)
            os.unlink(o)
        n = sizes.get("paired", 0)
        if n:
            f1, f2 = os.path.join(d, "R1.fq"), os.path.join(d, "R2.fq")
            t = time.perf_counter()
            first = {}
            cache = {}

            def mates(a, m, which):
                if cache.get("a") != a:
                    cache["a"], cache["x"] = a, make_pairs(genome_dev, m, 41 + a, device).reshape(m, 2, READ_LEN)
                    if a == 0:
                        first["seqs"] = cache["x"][:check_reads // 2].cpu().numpy()
                return cache["x"][:, which].contiguous()
            # the two files are written chunk by chunk in step, so that a chunk of pairs is generated once
            with open(f1, "wb"), open(f2, "wb"):
                pass
            chunk_n = 4_000_000
            for a in range(0, n, chunk_n):
                m = min(chunk_n, n - a)
                for which, path in ((0, f1), (1, f2)):
                    tmp = path + ".part"
                    write_fastq_from_device(tmp, lambda a_, m_, w_=which, a0=a: mates(a0, m_, w_), m, chunk=m, first_id=a)
                    with open(path, "ab") as dst, open(tmp, "rb") as src:
                        shutil.copyfileobj(src, dst, 1 << 24)
                    os.unlink(tmp)
            cache.clear()
            gen_s = time.perf_counter() - t
            o1, o2 = os.path.join(d, "out1.fq"), os.path.join(d, "out2.fq")
            r = filter_run(idx_path, [f1, f2], [o1, o2], ["-d"], os.path.join(d, "s.json"))
            r["workload"] = (f"{n:,} pairs of 2 x {READ_LEN} bp in two FASTQ files on tmpfs (configs[3]'s shape: half of the pairs "
                             f"from the host genome), -a 2 -r 0.01 -d -O, panhuman-sized index file")
            npairs = min(check_reads // 2, n)
            seqs = first["seqs"][:npairs].reshape(2 * npairs, READ_LEN)
            uid = (np.arange(2 * npairs, dtype=np.uint32) // 2)
            r["decisions_match"] = check(seqs, uid, True, o1, npairs) and \
                bool((ids_of_output(o1, npairs) == ids_of_output(o2, npairs)).all())
            r["checked"] = f"ids of the kept pairs among the first {npairs} pairs (both output files) == the oracle's decisions"
            r["input_generated_s"] = gen_s
            out["paired"] = r
            log(f"cli.paired: run() {r['run_wall_s']:.2f} s, {r['Mbp_per_s_incl_index_load'] / 1e3:.2f} Gbp/s incl. index load, "
                f"{r['Mbp_per_s_filter_only'] / 1e3:.2f} Gbp/s filter only, oracle ok={r['decisions_match']}")
        return out
    finally:
        shutil.rmtree(d, ignore_errors=True)


if __name__ == "__main__":  # python bench_cli.py: the plumbing leg alone (needs a GPU)
    sys.path.insert(0, ROOT)
    print(json.dumps(plumbing(threads=os.cpu_count() or 1), indent=1))
