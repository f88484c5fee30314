#!/usr/bin/env python3
"""Counters of the LARGEST dispatches of a kernel (by grid size) in rocprofv3 counter_collection CSVs, or of the
dispatches of a given grid size (the level-0 launch of a kernel is not always its largest grid: at 4K x 32 pairs
the iteration kernel runs 35 x 2 x 32 workgroups at level 0 and 18 x 4 x 32 at level 1).
usage: tools/pmc_top.py <dir> <kernel-substring> [grid-size]"""
import csv, glob, sys, collections
d, sub = sys.argv[1], sys.argv[2]
want = int(sys.argv[3]) if len(sys.argv) > 3 else None
for f in sorted(glob.glob(d + "/**/*counter_collection.csv", recursive=True)):
    rows = [r for r in csv.DictReader(open(f)) if sub in r["Kernel_Name"]]
    if not rows:
        continue
    big = want if want is not None else max(int(r["Grid_Size"]) for r in rows)
    acc, disp = collections.defaultdict(float), set()
    for r in rows:
        if int(r["Grid_Size"]) == big:
            acc[r["Counter_Name"]] += float(r["Counter_Value"])
            disp.add(r["Dispatch_Id"])
    print(f"{f.split('/')[-3] if f.count('/') > 2 else f}: {sub} grid {big}: {len(disp)} dispatches")
    for c, v in sorted(acc.items()):
        print(f"   {c:30s} {v / len(disp):18.1f}")
