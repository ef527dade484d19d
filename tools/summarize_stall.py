#!/usr/bin/env python3
"""Per-launch averages of the counters tools/gpu_stall_counters.sh collected (k_render only).  usage: summarize_stall.py [dir]"""
import collections, csv, glob, json, os, sys
d = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/stall"
out = {}
for g in sorted(glob.glob(os.path.join(d, "pmc*"))):
    # gpurun MERGES into gpurun_out/, so a pass directory can hold files of earlier calls: the newest one is this call's
    f = max(glob.glob(os.path.join(g, "*", "*_counter_collection.csv")), key=os.path.getmtime)
    rows = [r for r in csv.DictReader(open(f)) if "k_render" in r["Kernel_Name"]]
    n = len({r["Dispatch_Id"] for r in rows}) or 1
    agg = collections.defaultdict(float)
    for r in rows:
        agg[r["Counter_Name"]] += float(r["Counter_Value"])
    for k, v in agg.items():
        out[k] = v / n
print(json.dumps(out, indent=1))
