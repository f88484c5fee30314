#!/bin/bash
# usage (GPU box, repo root): tools/micro/remap_traffic.sh: HBM bytes per pixel of the one-kernel remap step (separate FETCH_SIZE / WRITE_SIZE passes
# over tools/kprof.py 4k 32; traffic = 2 * FETCH_SIZE + WRITE_SIZE KiB, profiles/README.md)
timeout -k 10 200 tools/pmc_pass.sh rm_f 4k 32 FETCH_SIZE
timeout -k 10 200 tools/pmc_pass.sh rm_w 4k 32 WRITE_SIZE
python3 - <<'P'
import csv, glob, collections
def tot(d, c):
    per = collections.defaultdict(float)
    for f in glob.glob(f"gpurun_out/pmc_{d}/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == c and "k_remap_step" in r["Kernel_Name"]:
                per[r["Dispatch_Id"]] += float(r["Counter_Value"])
    return sum(per.values()) / max(1, len(per)), len(per)
f, n = tot("rm_f", "FETCH_SIZE"); w, m = tot("rm_w", "WRITE_SIZE")
px = 3840 * 2160
print(f"k_remap_step: {n} dispatches; fetched {2 * f * 1024 / px:.2f} B/px, written {w * 1024 / px:.2f} B/px, total {(2 * f + w) * 1024 / px:.2f} B/px")
P
