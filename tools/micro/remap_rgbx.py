#!/usr/bin/env python3
"""Does the remap step's RGB gather (three byte loads at a scattered address) cost more than one dword load from an
RGBX copy of the pixmap would?  The timed step with its RGB pixmap, then with the same pixels as RGBA (alpha 1) through
the kernel's four-channel form; images compared.  usage (GPU box, repo root): python3 tools/micro/remap_rgbx.py [batch]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from transflow_amd import _lib  # noqa: E402

batch = int(sys.argv[1]) if len(sys.argv) > 1 else 16
wl = bench.WORKLOADS["4k"]
res = {}
for channels in (3, 4):
    job = bench.Job(wl, batch, bench.make_plan(batch + 1, batch, 0, 1), batch + 1, seed=2000, device=0, lanes=1)
    if channels == 4:
        rgba = np.concatenate([job.pixmap, np.ones(job.pixmap.shape[:2] + (1,), np.uint8)], axis=2)
        p = C.c_void_p()
        job.check(job.lib.tf_dev_alloc(C.byref(p), rgba.nbytes))
        job.check(job.lib.tf_dev_upload(p, C.c_void_p(rgba.ctypes.data), rgba.nbytes))
        job.pixmap_dev = p.value
        job.remap_pair = lambda layer, comp, i, job=job: layer.step_dev(comp, job.flow_ptrs[i], job.pixmap_dev, 4, clip_flow=True, seed=bench.SEED_U)
    for _ in range(2):
        job.step()
    job.sync()
    # the remap steps alone, nothing beside them: one pass of flows, then the frames' steps timed by events
    job.prof(True)
    job.prof_reset()
    for _ in range(3):
        for i in range(job.batch):
            job.remap_pair(job.layer, job.comps[i], i)
        job.sync()
    rep = job.prof_report()
    job.prof(False)
    for k, (cnt, ms) in rep.items():
        if k.startswith("remap_step"):
            print(f"channels={channels}: {k} {cnt} launches, {ms / cnt * 1e3:.1f} us per launch")
    res[channels] = [c.download().copy() for c in job.comps[:3]]
print("images equal:", all(np.array_equal(a, b) for a, b in zip(res[3], res[4])))
