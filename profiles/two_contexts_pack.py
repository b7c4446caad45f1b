#!/usr/bin/env python3
"""do the contexts of ONE process pack their pageable input side by side?

  python profiles/two_contexts_pack.py            (on the GPU box; prints one JSON line)

Each leg is a child process (the pool's width is fixed when the library first uses it: DCN_HOST_THREADS):
  one  @ T threads   one context, blocking dcn_filter_batch on pageable ASCII, R reads per call
  two  @ T threads   two contexts on GPU 0, one caller thread each, the same calls at the same time (--devices 0,0's shape)
for T = half of the CPUs the job may use and T = all of them (at most 12 / 24).  The criterion: two contexts with twice the
threads >= 1.5 x one context, and two contexts with the SAME threads no slower than one.  Decisions of every call are compared
with the first call's (same batch, same answer)."""
import json
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child(n_ctx, reads, calls):
    import threading

    import numpy as np
    import torch  # noqa: F401
    sys.path.insert(0, ROOT)
    import bench
    import deacon_server_amd as dcn
    dev = torch.device("cuda", 0)
    g = bench.make_host_genome(16_000_000, 3, dev)
    idx = dcn.Index.build([g.cpu().numpy()], 31, 15, device=0)
    host = bench.make_reads(g, reads, 5, dev).cpu().numpy()
    off = np.arange(reads + 1, dtype=np.uint64) * np.uint64(150)
    procs = [dcn.FilterProcessor(idx, max_batch_bases=reads * 150, max_batch_reads=reads) for _ in range(n_ctx)]
    first = procs[0].filter_batch(host, off)[0].copy()
    ok = [True] * n_ctx
    for p in procs[1:]:
        p.filter_batch(host, off)
    barrier = threading.Barrier(n_ctx)
    span = [None] * n_ctx

    def work(t):
        barrier.wait()
        t0 = time.perf_counter()
        for _ in range(calls):
            k = procs[t].filter_batch(host, off)[0]
            ok[t] = ok[t] and bool((k == first).all())
        span[t] = (t0, time.perf_counter())

    ths = [threading.Thread(target=work, args=(t,)) for t in range(n_ctx)]
    [t.start() for t in ths]
    [t.join() for t in ths]
    wall = max(s[1] for s in span) - min(s[0] for s in span)
    print(json.dumps({"contexts": n_ctx, "Mbp_per_s": n_ctx * calls * reads * 150 / wall / 1e6, "same_decisions": all(ok)}))


def main():
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        return child(int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]))
    sys.path.insert(0, ROOT)
    import bench
    cores = bench.host_cores()
    reads, calls = int(os.environ.get("DCN_TCP_READS", "4000000")), int(os.environ.get("DCN_TCP_CALLS", "8"))
    out = {"cpus": cores, "reads_per_call": reads, "calls_per_context": calls, "legs": []}
    for threads in sorted({max(2, min(12, cores // 2)), max(2, min(24, cores))}):
        for n_ctx in (1, 2):
            env = dict(os.environ, DCN_HOST_THREADS=str(threads))
            best = None
            for _ in range(2):
                p = subprocess.run([sys.executable, os.path.abspath(__file__), "child", str(n_ctx), str(reads), str(calls)],
                                   env=env, capture_output=True, text=True, timeout=600)
                if p.returncode != 0:
                    raise RuntimeError(p.stderr[-2000:])
                r = json.loads(p.stdout.strip().splitlines()[-1])
                if best is None or r["Mbp_per_s"] > best["Mbp_per_s"]:
                    best = r
            best["host_threads"] = threads
            out["legs"].append(best)
            print(f"[two_contexts_pack] {n_ctx} context(s), {threads} host threads: {best['Mbp_per_s'] / 1e3:.1f} Gbp/s, same decisions {best['same_decisions']}",
                  file=sys.stderr, flush=True)
    by = {(leg["contexts"], leg["host_threads"]): leg["Mbp_per_s"] for leg in out["legs"]}
    ts = sorted({t for _, t in by})
    if len(ts) == 2:
        out["two_contexts_twice_the_threads_vs_one"] = by[(2, ts[1])] / by[(1, ts[0])]
        out["two_contexts_same_threads_vs_one"] = by[(2, ts[0])] / by[(1, ts[0])]
    print(json.dumps(out))


if __name__ == "__main__":
    main()
