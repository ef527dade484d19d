#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc counter_collection CSVs for one kernel: per-dispatch sums of every counter.
usage: pmc_summary.py <kernel-prefix> <dir> [<dir> ...]"""
import csv, glob, collections, sys
pref = sys.argv[1]
for d in sys.argv[2:]:
    for f in glob.glob(f"{d}/**/*_counter_collection.csv", recursive=True):
        by = collections.defaultdict(lambda: collections.defaultdict(float))
        for r in csv.DictReader(open(f)):
            if r["Kernel_Name"].startswith(pref):
                by[int(r["Dispatch_Id"])][r["Counter_Name"]] += float(r["Counter_Value"])
        for i in sorted(by):
            print(d, i, " ".join(f"{k}={v:.4g}" for k, v in sorted(by[i].items())))
