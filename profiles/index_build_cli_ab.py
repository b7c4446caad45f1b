"""`deacon-hip index build` from a plain and from a gzip-compressed FASTA (400 Mbp, 8 records, 80-column lines)"""
import os, subprocess, sys, time
import numpy as np
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench_cli
BIN = bench_cli.BIN
d = "/dev/shm/index_gz"; os.makedirs(d, exist_ok=True)
rng = np.random.default_rng(1)
alpha = np.frombuffer(b"ACGT", np.uint8)
with open(f"{d}/g.fa", "wb") as f:
    for c in range(8):
        seq = alpha[rng.integers(0, 4, 50_000_000)].reshape(-1, 80)
        lines = np.concatenate([seq, np.full((seq.shape[0], 1), 10, np.uint8)], axis=1)
        f.write(b">chr%d\n" % c)
        f.write(lines.tobytes())
size = os.path.getsize(f"{d}/g.fa")
bench_cli.gzip_one_member(f"{d}/g.fa", f"{d}/g.fa.gz", size, 16)
print("fasta %.2f GB, gz %.2f GB" % (size / 1e9, os.path.getsize(f"{d}/g.fa.gz") / 1e9), flush=True)
for rep in range(2):
    for name, env in (("g.fa", {}), ("g.fa.gz", {}), ("g.fa.gz", {"DCN_CLI_NO_PARALLEL_GZ": "1"}), ("g.fa.gz", {"DCN_CLI_ZLIB_INFLATE": "1"})):
        t = time.perf_counter()
        p = subprocess.run([BIN, "index", "build", f"{d}/{name}", "-o", f"{d}/g.idx"], capture_output=True, text=True, env=dict(os.environ, DCN_CLI_TIMING="1", DCN_INDEX_TIMING="1", **env))
        dt = time.perf_counter() - t
        tail = " | ".join(p.stderr.strip().splitlines()[-3:])
        print(f"{name} {env}: {dt:.2f} s = {400 / dt:.0f} Mbp/s rc {p.returncode} idx {os.path.getsize(d + '/g.idx') / 1e6:.0f} MB | {tail}", flush=True)
import shutil; shutil.rmtree(d)
