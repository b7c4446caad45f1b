#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc CSVs: per kernel name, mean counter value per dispatch.
usage: python profiles/summarize_pmc.py gpurun_out/<prefix>_{sq1,sq2,fetch,write}/pmc_counter_collection.csv"""
import csv
import sys
from collections import defaultdict

acc = defaultdict(lambda: defaultdict(list))
for path in sys.argv[1:]:
    for r in csv.DictReader(open(path)):
        name = r["Kernel_Name"]
        short = name.split("(")[0].replace("(anonymous namespace)::", "").replace("void ", "")[:40]
        if "scan_kernel" in name:
            short = name.split("(dcn")[0].replace("void (anonymous namespace)::", "")
        acc[short][r["Counter_Name"]].append(float(r["Counter_Value"]))
keep = [k for k in acc if any(s in k for s in ("scan_kernel", "pack_kernel", "finish", "plan_", "distinct", "scan_"))]
for k in sorted(keep):
    print(k)
    for c, v in sorted(acc[k].items()):
        print(f"    {c:28s} n={len(v):3d} mean={sum(v)/len(v):.6g}")
