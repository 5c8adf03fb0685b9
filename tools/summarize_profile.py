#!/usr/bin/env python3
"""Summarise the rocprofv3 passes of tools/profile_kernel.sh for one tag: per kernel the dispatch count, mean duration
and mean counter values per dispatch.  usage: summarize_profile.py <tag> [kernel-substring]  -> JSON on stdout"""
import csv, glob, json, os, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1]
want = sys.argv[2] if len(sys.argv) > 2 else ""

def files(kind, pattern):
    return glob.glob(os.path.join(ROOT, "gpurun_out", f"prof_{tag}_{kind}", "**", pattern), recursive=True)

out = defaultdict(dict)
for f in files("stats", "*kernel_trace.csv"):
    dur = defaultdict(list)
    for row in csv.DictReader(open(f)):
        dur[row["Kernel_Name"]].append(int(row["End_Timestamp"]) - int(row["Start_Timestamp"]))
    for k, v in dur.items():
        out[k]["dispatches"] = len(v)
        out[k]["mean_us"] = sum(v) / len(v) / 1e3
        out[k]["min_us"] = min(v) / 1e3
        out[k]["max_us"] = max(v) / 1e3
for kind in ("sq", "fetch", "write"):
    for f in files(kind, "*counter_collection.csv"):
        acc = defaultdict(lambda: defaultdict(list))
        for row in csv.DictReader(open(f)):
            acc[row["Kernel_Name"]][row["Counter_Name"]].append(float(row["Counter_Value"]))
        for k, cs in acc.items():
            for c, v in cs.items():
                out[k][c] = sum(v) / len(v)
res = {k: v for k, v in out.items() if want in k}
print(json.dumps(res, indent=1))
