#!/usr/bin/env python3
"""Per-kernel averages of the counters tools/pmc_kernel.sh collected:  python tools/pmc_table.py <out-dir> [kernel substring] [json out]"""
import csv
import glob
import json
import os
import sys

src = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""
acc = {}


def key(name: str) -> str:
    """'void (anonymous namespace)::dsp_x_kernel<float, 2>(Args, ...)' -> 'dsp_x_kernel<float, 2>'"""
    n = name.replace("(anonymous namespace)::", "")
    n = n[5:] if n.startswith("void ") else n
    return n.split("(")[0].strip()[:80]


for f in glob.glob(f"{src}/*/*/*_counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        if want in r["Kernel_Name"]:
            acc.setdefault(key(r["Kernel_Name"]), {}).setdefault(r["Counter_Name"], []).append(float(r["Counter_Value"]))
out = {k: {c: sum(v) / len(v) for c, v in d.items()} | {"dispatches": max(len(v) for v in d.values())} for k, d in acc.items()}
for f in glob.glob(f"{src}/trace/*/*_kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        for k in out:
            if key(r["Name"]) == k:
                out[k]["trace_avg_ms"] = float(r["AverageNs"]) / 1e6
                out[k]["trace_calls"] = int(r["Calls"])
print(json.dumps(out, indent=1))
if len(sys.argv) > 3:
    json.dump(out, open(sys.argv[3], "w"), indent=1)
