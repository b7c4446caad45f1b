"""paired inputs: two plain files vs two .gz files (one gzip member each) vs interleaved stdin, 4 M pairs; where does the time go"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_cli
BIN = bench_cli.BIN
d = "/dev/shm/paired_gz"; os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(1)
alpha = np.frombuffer(b"ACGT", np.uint8)
genome = alpha[rng.integers(0, 4, 4_000_000)]
open(f"{d}/g.fa", "wb").write(b">g\n" + genome.tobytes() + b"\n")
subprocess.check_call([BIN, "index", "build", f"{d}/g.fa", "-o", f"{d}/g.idx", "-q"])
n = 4_000_000
for which in (1, 2):
    starts = rng.integers(0, len(genome) - 150, n)
    seqs = genome[starts[:, None] + np.arange(150)[None, :]]
    rnd = alpha[rng.integers(0, 4, (n, 150))]
    seqs = np.where((np.arange(n) % 2 == 0)[:, None], seqs, rnd).astype(np.uint8)
    bench_cli.fastq_records(seqs).tofile(f"{d}/R{which}.fq")
    bench_cli.gzip_one_member(f"{d}/R{which}.fq", f"{d}/R{which}.fq.gz", n * bench_cli.REC, 16)
print("sizes", os.path.getsize(f"{d}/R1.fq") / 1e9, os.path.getsize(f"{d}/R1.fq.gz") / 1e9, flush=True)
for rep in range(2):
    for name, inputs in (("plain", [f"{d}/R1.fq", f"{d}/R2.fq"]), ("gz", [f"{d}/R1.fq.gz", f"{d}/R2.fq.gz"]), ("single-gz", [f"{d}/R1.fq.gz"])):
        t = time.perf_counter()
        outs = ["-o", f"{d}/o1.fq"] + (["-O", f"{d}/o2.fq"] if len(inputs) == 2 else [])
        p = subprocess.run([BIN, "filter", "-d", f"{d}/g.idx", *inputs, *outs, "-q"], capture_output=True, env=dict(os.environ, DCN_CLI_TIMING="1"), text=True)
        dt = time.perf_counter() - t
        bases = n * 150 * len(inputs)
        print(f"{name}: wall {dt:.2f} s = {bases / dt / 1e9:.2f} Gbp/s rc {p.returncode} out {os.path.getsize(d + '/o1.fq')}", flush=True)
        if rep == 0:
            for line in p.stderr.splitlines():
                if "timing" in line:
                    print("    " + line[:400])
import shutil; shutil.rmtree(d)
