#!/usr/bin/env python3
"""HBM-side bytes per level pixel of the iteration kernel at the bench's own batch (consecutive pairs share a frame and
the XCD-aware order lets them share it in L2), from two tools/pmc_pass.sh passes over tools/kprof.py 4k 32
(FETCH_SIZE, WRITE_SIZE), merged into a traffic table as "fb_flow_iter_batched".
usage: tools/traffic_batched.py <fetch_dir> <write_dir> <traffic.json> [pairs=32]"""
import collections
import csv
import glob
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transflow_amd.roofline import level_is_fused, level_sizes  # noqa: E402

pairs = int(sys.argv[4]) if len(sys.argv) > 4 else 32


def total(d, counter):
    per_dispatch = collections.defaultdict(float)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter and "k_flow_iter_pc" in row["Kernel_Name"]:
                per_dispatch[row["Dispatch_Id"]] += float(row["Counter_Value"])
    return sum(per_dispatch.values()), len(per_dispatch)


f_kib, nf = total(sys.argv[1], "FETCH_SIZE")
w_kib, nw = total(sys.argv[2], "WRITE_SIZE")
assert nf == nw and nf > 0, (nf, nw)
px_per_set = sum(w * h for (w, h) in level_sizes(3840, 2160, 0.5, 5) if level_is_fused(w * h, pairs)) * pairs * 3  # 3 iterations
sets = nf / (3 * sum(1 for (w, h) in level_sizes(3840, 2160, 0.5, 5) if level_is_fused(w * h, pairs)))
bpp = (2 * f_kib + w_kib) * 1024 / (px_per_set * sets)
t = json.load(open(sys.argv[3]))
t["fb_flow_iter_batched"] = {"fetch_kib": f_kib / sets, "write_kib": w_kib / sets, "launches_seen": nf, "pairs": pairs,
                             "bytes_per_px": bpp,
                             "_per": "all launches of the iteration kernel in a step of 3840x2160 x %d pairs (levels 0-3), per level pixel" % pairs}
json.dump(t, open(sys.argv[3], "w"), indent=1)
print(f"fb_flow_iter_batched {bpp:.2f} B/px over {nf} launches ({sets:g} steps)")
