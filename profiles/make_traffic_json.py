#!/usr/bin/env python3
"""gpurun_out/r3_pmc_<workload>_{fetch,write}/ (profiles/collect_r3.sh) -> profiles/r03_traffic[_<workload>].json: mean FETCH_SIZE /
WRITE_SIZE per dispatch of the counting scan kernel, in the calibration of profiles/r02_traffic.json (scattered 16-byte reads:
counter x 1024 B = sector bytes, no x2), plus the per-kernel time summary of the kernel-trace run.
usage: python profiles/make_traffic_json.py short|long|mixed [round, default 3]
(round >= 4: profiles/collect_r4.sh; the bench's full result is read from gpurun_out/r4_*_detail.json, its stdout line being the compact one;
the per-dispatch durations of the counting scan kernel from the kernel trace are summarised per rotated batch as well)"""
import csv
import glob
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
wl = sys.argv[1]
RND = int(sys.argv[2]) if len(sys.argv) > 2 else 3


def counters(kind):
    acc = {}
    for path in glob.glob(os.path.join(ROOT, "gpurun_out", f"r{RND}_pmc_{wl}_{kind}", "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(path)):
            if "scan_kernel" in r["Kernel_Name"] and r["Kernel_Name"].rstrip().endswith("false>(dcn_scan_args)") is False:
                pass
            # the counting kernel: FAST == false (4th template argument), VAR == false
            if "scan_kernel<15, false, false, false" not in r["Kernel_Name"]:
                continue
            acc.setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}, {k: len(v) for k, v in acc.items()}


f, nf = counters("fetch")
w, nw = counters("write")
line = json.load(open(os.path.join(ROOT, "gpurun_out", f"r{RND}_pmc_{wl}_fetch.json" if RND < 4 else f"r{RND}_pmc_{wl}_fetch_detail.json")))
cal = json.load(open(os.path.join(ROOT, "profiles", "r02_traffic.json")))["calibration"]
out = {
    "source": f"rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE (separate passes, profiles/collect_r{RND}.sh) on `python3 bench.py --steps 3 --warmup 1 "
              f"--no-cpu-baseline --no-extras --workload {wl}`, MI355X, round {RND}; mean over the {nf.get('FETCH_SIZE', 0)} dispatches of "
              "scan_kernel<15,false,false,false> (the counting kernel)",
    "workload": {"workload": wl, "reads_per_batch": line["config"]["reads_per_batch_per_gpu"], "bases_per_batch": line["config"]["bases_per_batch_per_gpu"],
                 "index_keys": line["config"]["index_keys"], "host_genome_bases": line["config"]["host_genome_bases"],
                 "index_table_bytes": line["config"]["index_table_bytes"]},
    "scan_kernel": {"FETCH_SIZE_KiB": f["FETCH_SIZE"], "WRITE_SIZE_KiB": w["WRITE_SIZE"],
                    "hbm_bytes_per_launch": (f["FETCH_SIZE"] + w["WRITE_SIZE"]) * 1024.0,
                    "TCC_HIT": w.get("TCC_HIT"), "TCC_MISS": w.get("TCC_MISS"),
                    "algorithmic_bytes_per_launch": line["roofline"]["algorithmic_bytes_per_launch"],
                    "minimizers_per_launch": line["roofline"]["minimizers_per_launch"]},
    "calibration": cal,
}
name = f"r{RND:02d}_traffic.json" if wl == "short" else f"r{RND:02d}_traffic_{wl}.json"
json.dump(out, open(os.path.join(ROOT, "profiles", name), "w"), indent=1)
print(name, json.dumps(out["scan_kernel"]))
# kernel-trace summary of the same workload
for path in glob.glob(os.path.join(ROOT, "gpurun_out", f"r{RND}_stats_{wl}", "**", "*kernel_stats.csv"), recursive=True):
    dst = os.path.join(ROOT, "profiles", f"r{RND:02d}_kernel_stats.csv" if wl == "short" else f"r{RND:02d}_kernel_stats_{wl}.csv")
    rows = list(csv.reader(open(path)))
    keep = [rows[0]] + [r for r in rows[1:] if any(s in r[0] for s in ("scan_kernel", "pack_kernel", "plan_kernel", "distinct", "big_insert", "finish_kernel", "table_", "probe_"))]
    csv.writer(open(dst, "w")).writerows(keep)
    print("wrote", dst, len(keep) - 1, "kernels")

# per-dispatch durations of the counting scan kernel, in launch order (kernel trace): is the spread a property of the batch
# (three are rotated: dispatch i runs batch i % 3), of time, or neither?
for path in glob.glob(os.path.join(ROOT, "gpurun_out", f"r{RND}_stats_{wl}", "**", "*kernel_trace.csv"), recursive=True):
    rows = [r for r in csv.DictReader(open(path)) if "scan_kernel<15, false, false, false" in r["Kernel_Name"]]
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    ms = [(int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e6 for r in rows]
    if ms:
        dst = os.path.join(ROOT, "profiles", f"r{RND:02d}_scan_dispatches{'' if wl == 'short' else '_' + wl}.txt")
        with open(dst, "w") as fo:
            fo.write(f"# duration (ms) of every dispatch of scan_kernel<15,false,false,false> in launch order, {wl} workload, rocprofv3 --kernel-trace (profiles/collect_r{RND}.sh)\n")
            fo.write(" ".join(f"{x:.3f}" for x in ms) + "\n")
            n = len(ms)
            fo.write(f"# n={n} mean={sum(ms) / n:.3f} min={min(ms):.3f} max={max(ms):.3f}\n")
            for b in range(3):
                sel = ms[b::3]
                if sel:
                    fo.write(f"# dispatches {b} mod 3: n={len(sel)} mean={sum(sel) / len(sel):.3f} min={min(sel):.3f} max={max(sel):.3f}\n")
        print("wrote", dst)
