"""The tool's gzip readers on the GPU box: zlib, fast_inflate.hpp on one thread, parallel_gzip.hpp on 2..16 threads --
`deacon-hip cat --count` (input side alone) and `filter -d` end to end, on a gzip stream and a BGZF file of the same
4 M x 150 bp FASTQ, constant and random quality strings."""
import gzip, os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(sys.path[0], "tests"))
import bench_cli
from test_cli import bgzf_compress
BIN = bench_cli.BIN
d = "/dev/shm/fastgz_ab"; os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(1)
alpha = np.frombuffer(b"ACGT", np.uint8)
genome = alpha[rng.integers(0, 4, 4_000_000)]
open(f"{d}/g.fa", "wb").write(b">g\n" + genome.tobytes() + b"\n")
subprocess.check_call([BIN, "index", "build", f"{d}/g.fa", "-o", f"{d}/g.idx", "-q"])
n = 4_000_000
starts = rng.integers(0, len(genome) - 150, n)
seqs = genome[starts[:, None] + np.arange(150)[None, :]]
rnd = alpha[rng.integers(0, 4, (n, 150))]
seqs = np.where((rng.random(n) < 0.5)[:, None], seqs, rnd).astype(np.uint8)
rec = bench_cli.fastq_records(seqs)
files = {}
for qual in ("const", "random"):
    r = rec.copy()
    if qual == "random":
        o = 2 + bench_cli.ID_DIGITS + bench_cli.READ_LEN
        r[:, o + 3:o + 3 + bench_cli.READ_LEN] = rng.integers(33, 74, (n, bench_cli.READ_LEN), dtype=np.uint8)
    fq = r.tobytes()
    assert fq.count(b"\n") == 4 * n
    open(f"{d}/{qual}.plain.fq.gz", "wb").write(gzip.compress(fq, 4))
    open(f"{d}/{qual}.bgzf.fq.gz", "wb").write(bgzf_compress(fq, level=4))
    files[qual] = len(fq)
    print(qual, "FASTQ %.2f GB -> gzip %.2f GB, bgzf %.2f GB" % (len(fq) / 1e9, os.path.getsize(f"{d}/{qual}.plain.fq.gz") / 1e9, os.path.getsize(f"{d}/{qual}.bgzf.fq.gz") / 1e9), flush=True)
    del fq, r
WAYS = [("zlib", {"DCN_CLI_ZLIB_INFLATE": "1"}), ("one", {"DCN_CLI_NO_PARALLEL_GZ": "1"})] + [("par%d" % t, {"DCN_CLI_GZ_THREADS": str(t)}) for t in (2, 4, 8, 12, 16)] + [("default", {})]
for rep in range(2):
    for qual in ("const", "random"):
        for name in ("plain", "bgzf"):
            for way, env in (WAYS if name == "plain" else [WAYS[0], WAYS[-1]]):
                f = f"{d}/{qual}.{name}.fq.gz"
                e = dict(os.environ, **env)
                t = time.perf_counter()
                p = subprocess.run([BIN, "cat", f, "--count"], capture_output=True, env=dict(e, DCN_CLI_GZ_STATS="1"))
                stats = [l for l in p.stderr.decode().splitlines() if l.startswith("gzip reader:")]
                dt_cat = time.perf_counter() - t
                t = time.perf_counter()
                q = subprocess.run([BIN, "filter", "-d", f"{d}/g.idx", f, "-o", f"{d}/out.fq", "-q"], capture_output=True, env=e)
                dt = time.perf_counter() - t
                print("%-6s %-5s %-7s cat %.2f s = %.2f GB/s of text | filter wall %.2f s = %.2f Gbp/s  rc %d %d out %d" % (
                    qual, name, way, dt_cat, files[qual] / dt_cat / 1e9, dt, n * 150 / dt / 1e9, p.returncode, q.returncode, os.path.getsize(f"{d}/out.fq")), flush=True)
                if stats and rep == 0:
                    print("        " + stats[0], flush=True)
import shutil; shutil.rmtree(d)
