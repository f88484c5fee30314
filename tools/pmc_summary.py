#!/usr/bin/env python3
"""Sum the counters of rocprofv3 counter_collection CSVs per kernel (substring match on the name).
usage: tools/pmc_summary.py <dir> <kernel-substring> [...]"""
import csv, glob, sys, collections
d = sys.argv[1]
subs = sys.argv[2:]
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    acc = collections.defaultdict(lambda: collections.defaultdict(float))
    disp = collections.defaultdict(set)
    for row in csv.DictReader(open(f)):
        k = row["Kernel_Name"]
        for s in subs:
            if s in k:
                acc[s][row["Counter_Name"]] += float(row["Counter_Value"])
                disp[s].add(row["Dispatch_Id"])
    for s in subs:
        n = max(1, len(disp[s]))
        print(f"{s}: {n} dispatches")
        for c, v in sorted(acc[s].items()):
            print(f"   {c:28s} {v / n:16.1f} per dispatch")
