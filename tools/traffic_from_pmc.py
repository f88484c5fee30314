#!/usr/bin/env python3
"""profiles/r01_traffic.json from two rocprofv3 passes over tools/calibrate_fetch.py
(--pmc FETCH_SIZE and --pmc WRITE_SIZE, each with --kernel-trace --output-format csv).
traffic = 2 * FETCH_SIZE + WRITE_SIZE (KiB * 1024), per dispatch, divided by the level pixels; the
factor 2 is MI355X_MICROARCH.md's gfx950 correction, checked here on k_pp_clip's known byte count.
usage: tools/traffic_from_pmc.py <fetch_dir> <write_dir> <out.json>"""
import collections
import csv
import glob
import json
import sys

PIXELS = 3840 * 2160
KERNELS = {"fb_flow_iter": ("k_flow_iter_pc", 1), "fb_update_matrices": ("k_update_matrices", 1),
           "fb_blur_solve": ("k_blur_solve_wave", 1),
           "fb_level_polyexp": ("k_level0_polyexp_t", 2), "pp_clip": ("k_pp_clip", 1),
           # (a name may ask for several substrings: "a&b")
           "fb_flow_carry": ("k_flow_carry_pc&false>(", 1), "fb_flow_vsum": ("k_flow_carry_pc&true>(", 1),
           "fb_exact_vsum": ("k_exact_vsum", 1), "fb_exact_hsolve": ("k_exact_hsolve", 1)}


def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        per_dispatch = collections.defaultdict(float)
        names = {}
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                per_dispatch[row["Dispatch_Id"]] += float(row["Counter_Value"])
                names[row["Dispatch_Id"]] = row["Kernel_Name"]
        for did, v in per_dispatch.items():
            acc[names[did]].append(v)
    return acc


fetch, write = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"_note": "HBM-side bytes per level pixel and launch, from separate rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE "
                "passes over tools/calibrate_fetch.py (3840x2160, one pair, one scale); traffic = 2*FETCH_SIZE + "
                "WRITE_SIZE (KiB*1024): FETCH_SIZE reports half of the bytes on gfx950 for 4-, 8- and 16-byte loads "
                "(check: pp_clip below reads 8 B/px and writes 8 B/px). tools/traffic_from_pmc.py.",
       "pixels": PIXELS}
for label, (sub, images) in KERNELS.items():
    fk = [v for k, vs in fetch.items() if all(p in k for p in sub.split("&")) for v in vs]
    wk = [v for k, vs in write.items() if all(p in k for p in sub.split("&")) for v in vs]
    if not fk or not wk:
        continue
    f_kib, w_kib = sum(fk) / len(fk), sum(wk) / len(wk)
    out[label] = {"fetch_kib": f_kib, "write_kib": w_kib, "launches_seen": len(fk),
                  "bytes_per_px": (2 * f_kib + w_kib) * 1024 / (PIXELS * images)}
    if images > 1:
        out[label]["_per"] = "per image (the launch covers 2 images)"
json.dump(out, open(sys.argv[3], "w"), indent=1)
for k, v in out.items():
    if isinstance(v, dict):
        print(f"{k:22s} {v['bytes_per_px']:7.2f} B/px  ({v['launches_seen']} launches)")
