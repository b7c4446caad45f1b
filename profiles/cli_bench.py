#!/usr/bin/env python3
"""f2 measurement: `deacon-hip filter` end to end on files (tmpfs): FASTQ parse -> GPU -> FASTQ write.
usage: python profiles/cli_bench.py [n_reads] [genome_bases]"""
import json, os, subprocess, sys, tempfile, time
import numpy as np

BIN = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "deacon-server_amd", "bin", "deacon-hip")
n_reads = int(sys.argv[1]) if len(sys.argv) > 1 else 16_000_000
n_genome = int(sys.argv[2]) if len(sys.argv) > 2 else 64_000_000
rng = np.random.default_rng(1)
d = tempfile.mkdtemp(dir="/dev/shm" if os.path.isdir("/dev/shm") else None)
try:
    genome = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, n_genome)]
    with open(os.path.join(d, "g.fa"), "wb") as f:
        f.write(b">host\n"); f.write(genome.tobytes()); f.write(b"\n")
    t = time.perf_counter()
    subprocess.run([BIN, "index", "build", os.path.join(d, "g.fa"), "-o", os.path.join(d, "g.idx"), "-q"], check=True,
                   stderr=subprocess.DEVNULL)
    print(f"index build CLI: {n_genome/1e6:.0f} Mbp FASTA -> index file in {time.perf_counter()-t:.2f} s "
          f"({os.path.getsize(os.path.join(d,'g.idx'))/1e6:.0f} MB)")
    # FASTQ with 150 bp reads, half from the genome
    L = 150
    starts = rng.integers(0, n_genome - L, n_reads)
    mat = genome[starts[:, None] + np.arange(L)[None, :]]
    rnd = np.frombuffer(b"ACGT", dtype=np.uint8)[rng.integers(0, 4, (n_reads, L))]
    host = rng.random(n_reads) < 0.5
    mat = np.where(host[:, None], mat, rnd)
    rec = np.empty((n_reads, 2 * L + 16), np.uint8)
    ids = np.char.zfill(np.arange(n_reads).astype(str), 9).astype("S9")
    rec[:, 0] = ord("@"); rec[:, 1:10] = np.frombuffer(ids.tobytes(), np.uint8).reshape(n_reads, 9); rec[:, 10] = 10
    rec[:, 11:11 + L] = mat; rec[:, 11 + L] = 10; rec[:, 12 + L] = ord("+"); rec[:, 13 + L] = 10
    rec[:, 14 + L:14 + 2 * L] = ord("I"); rec[:, 14 + 2 * L] = 10
    rec = rec[:, :15 + 2 * L]
    fq = os.path.join(d, "r.fq")
    rec.tofile(fq)
    size = os.path.getsize(fq)
    # A/B runs on the same files: DCN_CLI_BINS = comma-separated "binary[:ENV=VAL[:ENV=VAL...]]" entries
    bins = [b for b in os.environ.get("DCN_CLI_BINS", BIN).split(",") if b]
    for run_no, (spec, mode, extra) in enumerate([(b, m, e) for b in bins for m, e in (("search", []), ("deplete", ["-d"]))]):
        BIN, *envs = spec.split(":")
        env = dict(os.environ, **dict(kv.split("=", 1) for kv in envs))
        if len(bins) > 1:
            print(os.path.basename(BIN), " ".join(envs), end=": ", flush=True)
        t = time.perf_counter()
        subprocess.run([BIN, "filter", os.path.join(d, "g.idx"), fq, "-o", os.path.join(d, f"out_{run_no}.fq"), "-s",
                        os.path.join(d, "s.json"), "-q", *extra, *os.environ.get("DCN_CLI_ARGS", "").split()], check=True, env=env)
        dt = time.perf_counter() - t
        s = json.load(open(os.path.join(d, "s.json")))
        os.unlink(os.path.join(d, f"out_{run_no}.fq"))
        print(f"filter CLI ({mode}): {n_reads} x {L} bp FASTQ ({size/1e9:.2f} GB) in {dt:.2f} s wall; summary: "
              f"{s['bp_per_second']/1e6:.0f} Mbp/s incl. index load, kept {s['seqs_out']}/{s['seqs_in']}; "
              f"{size/dt/1e9:.2f} GB/s of FASTQ")
finally:
    subprocess.run(["rm", "-rf", d])
