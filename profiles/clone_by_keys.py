#!/usr/bin/env python3
"""How long does a replica of the panhuman-sized index take, made from the table and made from the keys?

  python profiles/clone_by_keys.py [n_keys]        (on the GPU box; prints one JSON line)

The one-GPU box can only clone onto the same device, so the link is not in these numbers: both forms are timed end to end on
GPU 0 (DCN_CLONE_BY_KEYS=1 sends the same-device clone down the cross-device code), with the bytes each form would put on an
xGMI link beside them (the table vs 8 bytes per key) and the time those bytes take at 50 GB/s of one link's payload rate
(an assumption, stated as such: no second device to measure)."""
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import deacon_server_amd as dcn  # noqa: E402


def main():
    n_keys = int(sys.argv[1]) if len(sys.argv) > 1 else 409_913_780
    dev = torch.device("cuda", 0)
    keys = np.empty(n_keys, np.uint64)
    step = 1 << 27
    for a in range(0, n_keys, step):
        m = min(step, n_keys - a)
        keys[a:a + m] = bench.mix64_device(1 + a, m, dev).cpu().numpy().view(np.uint64)
    src = dcn.Index.from_keys(keys, 31, 15, device=0)
    table_bytes = src.table_bytes
    probe = keys[:: max(1, n_keys // 1_000_000)].copy()
    out = {"n_keys": src.n_keys, "table_bytes": table_bytes, "key_bytes": 8 * src.n_keys}
    for form in ("copy", "keys", "copy", "keys"):
        if form == "keys":
            os.environ["DCN_CLONE_BY_KEYS"] = "1"
        else:
            os.environ.pop("DCN_CLONE_BY_KEYS", None)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        rep = src.clone(0)
        dt = time.perf_counter() - t0
        ok = rep.n_keys == src.n_keys and bool(rep.contains(probe).all()) and not bool(rep.contains(probe ^ np.uint64(1)).any())
        out.setdefault(f"clone_by_{form}_s", []).append(round(dt, 4))
        out[f"clone_by_{form}_ok"] = out.get(f"clone_by_{form}_ok", True) and ok
        rep.close()
    link = 50e9
    if table_bytes:
        out["link_s_at_50GBps"] = {"table": round(table_bytes / link, 3), "keys": round(8 * src.n_keys / link, 3)}
    print(json.dumps(out))


if __name__ == "__main__":
    main()
