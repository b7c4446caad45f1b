#!/usr/bin/env python3
"""f2 measurement: `deacon-hip filter` end to end on files (tmpfs): FASTQ parse -> GPU -> FASTQ write.
usage: python profiles/cli_bench.py [n_reads] [genome_bases] [index_keys] [host_fraction]
  host_fraction (default 0.5): share of the reads drawn from the indexed genome; a host-depletion run of a clinical
  sample is 0.9-0.99 (`-d` then writes the few per cent that are left), a 0.5 mix writes half of the input back
  index_keys > 0: the index file also holds mix64 keys up to that many (409913780 = panhuman-1's size, a 3.7 GB file)
Variants run on the same files: DCN_CLI_VARIANTS = ';'-separated entries of space-separated tokens, each token
either ENV=VALUE or a command-line argument, e.g. "-t 16;DCN_CLI_NO_MMAP_OUT=1 -t 16" (default: one run, no extras).
The summary's bp_per_second is the reference's rate definition (src/local_filter.rs:726-729: bases / wall time of
run(), index load included); "filter only" subtracts the index-load milestone (DCN_CLI_TIMING)."""
import json
import os
import re
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "deacon-server_amd", "bin", "deacon-hip")
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
n_genome = int(sys.argv[2]) if len(sys.argv) > 2 else 64_000_000
n_keys = int(sys.argv[3]) if len(sys.argv) > 3 else 0
host_frac = float(sys.argv[4]) if len(sys.argv) > 4 else 0.5
rng = np.random.default_rng(1)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n_genome)]
    with open(os.path.join(d, "g.fa"), "wb") as f:
        f.write(b">host\n")
        f.write(genome.tobytes())
        f.write(b"\n")
    t = time.perf_counter()
    idx_path = os.path.join(d, "g.idx")
    subprocess.run([BIN, "index", "build", os.path.join(d, "g.fa"), "-o", idx_path, "-q"], check=True, stderr=subprocess.DEVNULL)
    print(f"index build CLI: {n_genome/1e6:.0f} Mbp FASTA -> index file in {time.perf_counter()-t:.2f} s "
          f"({os.path.getsize(idx_path)/1e6:.0f} MB)", flush=True)
    if n_keys:
        import torch  # noqa: F401
        sys.path.insert(0, ROOT)
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import deacon_server_amd as dcn
        from conftest import mix64
        t = time.perf_counter()
        small = dcn.Index.from_file(idx_path)
        hk = small.keys()
        small.close()
        keys = np.empty(n_keys, np.uint64)
        keys[:len(hk)] = hk
        step = 1 << 26
        for a in range(len(hk), n_keys, step):
            m = min(step, n_keys - a)
            keys[a:a + m] = mix64(np.arange(1 + a - len(hk), 1 + a - len(hk) + m, dtype=np.uint64))
        big = dcn.Index.from_keys(keys, 31, 15)
        idx_path = os.path.join(d, "big.idx")
        big.write(idx_path)
        big.close()
        del keys
        print(f"panhuman-sized index: {n_keys:,} keys -> {os.path.getsize(idx_path)/1e9:.2f} GB file in {time.perf_counter()-t:.1f} s", flush=True)
    # FASTQ with 150 bp reads, host_frac of them from the genome
    L = 150
    starts = rng.integers(0, n_genome - L, n_reads)
    mat = genome[starts[:, None] + np.arange(L)[None, :]]
    rnd = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n_reads, L))]
    host = rng.random(n_reads) < host_frac
    mat = np.where(host[:, None], mat, rnd)
    rec = np.empty((n_reads, 2 * L + 16), np.uint8)
    ids = np.char.zfill(np.arange(n_reads).astype(str), 9).astype("S9")
    rec[:, 0] = ord("@")
    rec[:, 1:10] = np.frombuffer(ids.tobytes(), np.uint8).reshape(n_reads, 9)
    rec[:, 10] = 10
    rec[:, 11:11 + L] = mat
    rec[:, 11 + L] = 10
    rec[:, 12 + L] = ord("+")
    rec[:, 13 + L] = 10
    rec[:, 14 + L:14 + 2 * L] = ord("I")
    rec[:, 14 + 2 * L] = 10
    rec = rec[:, :15 + 2 * L]
    fq = os.path.join(d, "r.fq")
    rec.tofile(fq)
    del rec, mat, rnd
    size = os.path.getsize(fq)
    if os.environ.get("DCN_CLI_IN_CODEC"):  # compressed input: "gzip -1" / "zstd -1" style command, e.g. DCN_CLI_IN_CODEC="gzip -1 -k"
        t = time.perf_counter()
        subprocess.run(os.environ["DCN_CLI_IN_CODEC"].split() + [fq], check=True)
        ext = ".gz" if "gzip" in os.environ["DCN_CLI_IN_CODEC"] else ".zst"
        fq = fq + ext
        print(f"input compressed to {os.path.getsize(fq)/1e9:.2f} GB in {time.perf_counter()-t:.1f} s", flush=True)
    variants = [v.split() for v in os.environ.get("DCN_CLI_VARIANTS", "").split(";")] or [[]]
    # a variant may name another driver binary (DCN_CLI_BIN=path, relative to the repo): old vs new on the same files
    bins = sorted({t_.split("=", 1)[1] for v in variants for t_ in v if t_.startswith("DCN_CLI_BIN=")} | {BIN})
    for b_ in bins:
        for th in (8, 16):
            p = subprocess.run([os.path.join(ROOT, b_), "bench-parse", fq, "-t", str(th)], capture_output=True, text=True)
            print(f"[{os.path.basename(b_)}] {p.stdout.strip()}", flush=True)
    modes = (("search", []), ("deplete", ["-d"]))
    if os.environ.get("DCN_CLI_SEARCH_ONLY"):
        modes = modes[:1]
    if os.environ.get("DCN_CLI_DEPLETE_ONLY"):
        modes = modes[1:]
    for vi, toks in enumerate(variants):
        envs = dict(t_.split("=", 1) for t_ in toks if re.match(r"^[A-Z_]+=", t_))
        args = [t_ for t_ in toks if not re.match(r"^[A-Z_]+=", t_)]
        for mode, extra in modes:
            env = dict(os.environ, DCN_CLI_TIMING="1", **envs)
            out = os.path.join(d, "out.fq" + os.environ.get("DCN_CLI_OUT_EXT", ""))  # ".gz" / ".zst" / ".xz": compressed output
            if envs.get("SLEEP_BEFORE"):  # a variant token SLEEP_BEFORE=seconds: pause between this run and the one before it
                time.sleep(float(envs["SLEEP_BEFORE"]))
            t = time.perf_counter()
            inputs = [fq, fq] if os.environ.get("DCN_CLI_PAIRED") else [fq]  # paired: the same file as both mates
            p = subprocess.run([os.path.join(ROOT, envs.get("DCN_CLI_BIN", BIN)), "filter", idx_path, *inputs, "-o", out, "-s", os.path.join(d, "s.json"), "-q", *extra, *args],
                               check=True, env=env, capture_output=True, text=True)
            dt = time.perf_counter() - t
            s = json.load(open(os.path.join(d, "s.json")))
            out_size = os.path.getsize(out)
            os.unlink(out)
            m = re.search(r"index loaded ([0-9.]+)", p.stderr)
            w = re.search(r"timing: wall ([0-9.]+)", p.stderr)
            t_idx, t_wall = (float(m.group(1)) if m else 0.0), (float(w.group(1)) if w else s["time"])
            print(f"[{' '.join(toks) or 'default'}] filter ({mode}): {n_reads} x {L} bp FASTQ ({size/1e9:.2f} GB in, {out_size/1e9:.2f} GB out) "
                  f"in {dt:.2f} s process wall; run() {t_wall:.3f} s = {s['bp_in']/t_wall/1e6:.0f} Mbp/s incl. index load "
                  f"({t_idx:.3f} s), {s['bp_in']/max(t_wall - t_idx, 1e-9)/1e6:.0f} Mbp/s filter only; kept {s['seqs_out']}/{s['seqs_in']}", flush=True)
            for line in p.stderr.splitlines():
                if line.startswith("timing:") or line.startswith("load timing:"):
                    print("    " + line)
finally:
    subprocess.run(["rm", "-rf", d])
