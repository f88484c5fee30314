#!/usr/bin/env python3
"""HBM-side bytes of a WHOLE STEP of the bench's workload, kernel by kernel, from two rocprofv3 counter passes over
tools/kprof.py <workload> <pairs> (tools/pmc_pass.sh: --pmc FETCH_SIZE and --pmc WRITE_SIZE, each in a pass of its own
with --kernel-trace only, the program directly after `--`).  Every dispatch of the process is summed -- kprof.py runs
`steps` identical steps and nothing else on the GPU -- and divided by the steps:

    bytes = 2 * FETCH_SIZE + WRITE_SIZE   (KiB * 1024; the factor 2 is MI355X_MICROARCH.md's gfx950 correction for
                                           FETCH_SIZE, calibrated in profiles/r05_traffic.json on k_pp_clip's known 16 B/px)

-> profiles/r06_traffic_step.json, which bench.py reads for roofline.whole_step.counter_GBs / counter_frac (a profile
constant scaled by nothing: it is used only when the run's workload and pairs per pass are the table's).
usage: tools/traffic_step.py <fetch_dir> <write_dir> <out.json> <workload> <pairs> <steps>"""
import collections
import csv
import glob
import json
import os
import re
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def short(name):
    """`void (anonymous namespace)::k_flow_iter_pc<7, 2>(...)` -> `k_flow_iter_pc`"""
    m = re.search(r"(k_[A-Za-z0-9_]+)", name)
    return m.group(1) if m else name.split("(")[0].strip()[:60]


def per_kernel(d, counter):
    per_dispatch, names = collections.defaultdict(float), {}
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for row in csv.DictReader(open(f)):
            if row["Counter_Name"] == counter:
                key = (f, row["Dispatch_Id"])
                per_dispatch[key] += float(row["Counter_Value"])
                names[key] = short(row["Kernel_Name"])
    tot, cnt = collections.defaultdict(float), collections.Counter()
    for key, v in per_dispatch.items():
        tot[names[key]] += v
        cnt[names[key]] += 1
    return tot, cnt


def main():
    fetch_dir, write_dir, out_path, workload, pairs, steps = sys.argv[1:7]
    pairs, steps = int(pairs), int(steps)
    import bench
    from transflow_amd import roofline as rf
    wl = bench.WORKLOADS[workload]
    fetch, nf = per_kernel(fetch_dir, "FETCH_SIZE")
    write, nw = per_kernel(write_dir, "WRITE_SIZE")
    assert fetch and write, "no counter rows found"
    kernels, setup = {}, {}
    for k in sorted(set(fetch) | set(write)):
        assert nf[k] == nw[k], f"{k}: {nf[k]} dispatches in the FETCH_SIZE pass, {nw[k]} in the WRITE_SIZE pass"
        if nf[k] % steps != 0 or not k.startswith("k_"):
            # not part of the steps: the runtime's own copy / fill kernels behind the frames' uploads and the buffers' set-up
            setup[k] = {"dispatches": nf[k], "bytes": (2 * fetch[k] + write[k]) * 1024}
            continue
        b = (2 * fetch[k] + write[k]) * 1024 / steps
        kernels[k] = {"launches_per_step": nf[k] // steps, "fetch_kib_per_step": fetch[k] / steps,
                      "write_kib_per_step": write[k] / steps, "bytes_per_step": b}
    total = sum(v["bytes_per_step"] for v in kernels.values())
    n = [w * h for w, h in rf.level_sizes(wl["w"], wl["h"], 0.5, wl["levels"])]
    built = rf.built_step_bytes(wl["w"], wl["h"], wl["levels"], pairs, reset_mask=wl["reset"], forward=wl["direction"] == 0)
    out = {"_note": "HBM-side bytes per STEP (one pass of `pairs` frame pairs: Farneback + the remap steps) of tools/kprof.py "
                    f"{workload} {pairs}, every dispatch of the process summed over {steps} identical steps: 2*FETCH_SIZE + "
                    "WRITE_SIZE (KiB*1024) from separate rocprofv3 --pmc passes; tools/traffic_step.py",
           "workload": workload, "width": wl["w"], "height": wl["h"], "levels": wl["levels"], "pairs": pairs, "steps": steps,
           "bytes_per_step": total, "built_bytes_per_step": built, "counter_over_built": total / built,
           "bytes_per_pair_and_full_resolution_pixel": total / (pairs * n[0]),
           "kernels": dict(sorted(kernels.items(), key=lambda kv: -kv[1]["bytes_per_step"])),
           "outside_the_steps": setup}
    json.dump(out, open(out_path, "w"), indent=1)
    print(f"{workload} x {pairs}: {total / 1e9:.2f} GB per step by the counters, {built / 1e9:.2f} GB built ({total / built:.3f})")
    for k, v in out["kernels"].items():
        print(f"  {k:28s} {v['launches_per_step']:4d} launches/step  {v['bytes_per_step'] / 1e9:8.3f} GB/step")


if __name__ == "__main__":
    main()
