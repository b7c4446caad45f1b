"""how long do compressed outputs take: 4 M reads, half kept (630 MB of FASTQ out), plain / .gz / .zst outputs, constant and random quality strings"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_cli
BIN = bench_cli.BIN
d = "/dev/shm/gzout"; os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(1)
alpha = np.frombuffer(b"ACGT", np.uint8)
genome = alpha[rng.integers(0, 4, 4_000_000)]
open(f"{d}/g.fa", "wb").write(b">g\n" + genome.tobytes() + b"\n")
subprocess.check_call([BIN, "index", "build", f"{d}/g.fa", "-o", f"{d}/g.idx", "-q"])
n = 4_000_000
starts = rng.integers(0, len(genome) - 150, n)
seqs = genome[starts[:, None] + np.arange(150)[None, :]]
rnd = alpha[rng.integers(0, 4, (n, 150))]
seqs = np.where((np.arange(n) % 2 == 0)[:, None], seqs, rnd).astype(np.uint8)
rec = bench_cli.fastq_records(seqs)
for qual in ("const", "random"):
    r = rec.copy()
    if qual == "random":
        o = 2 + bench_cli.ID_DIGITS + bench_cli.READ_LEN
        r[:, o + 3:o + 3 + bench_cli.READ_LEN] = rng.integers(33, 74, (n, bench_cli.READ_LEN), dtype=np.uint8)
    r.tofile(f"{d}/{qual}.fq")
    del r
for rep in range(2):
    for qual in ("const", "random"):
        for ext, extra, env in (("", [], {}), (".gz", [], {}), (".gz", [], {"DCN_CLI_ZLIB_DEFLATE": "1"}), (".gz", ["--compression-level", "4"], {}), (".zst", [], {})):
            out = f"{d}/out.fq{ext}"
            t = time.perf_counter()
            p = subprocess.run([BIN, "filter", f"{d}/g.idx", f"{d}/{qual}.fq", "-o", out, "-q", *extra] + sys.argv[1:], capture_output=True, env=dict(os.environ, DCN_CLI_TIMING="1", **env), text=True)
            dt = time.perf_counter() - t
            w = [l for l in p.stderr.splitlines() if l.startswith("timing: wall")]
            print(f"{qual:6s} out{ext or '.fq':5s} {' '.join(extra) + ' ' + ' '.join(env):28s} wall {dt:.2f} s rc {p.returncode} size {os.path.getsize(out) / 1e6:.0f} MB | {w[0][:190] if w else ''}", flush=True)
import shutil; shutil.rmtree(d)
