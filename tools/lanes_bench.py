#!/usr/bin/env python3
"""Batches alternating between a Farneback handle and its lane (tf_fb_create_lane: two call streams) against one handle
doing every batch: frames/s of the Farneback + remap step.  usage (GPU box): python3 tools/lanes_bench.py [workload] [batch] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from transflow_amd import _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "4k"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 32
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 12
for arg in sys.argv[4:]:
    k, v = arg.split("=")
    _lib.set_option(k, int(v))


def make(lanes):
    return bench.Job(bench.WORKLOADS[name], batch, bench.make_plan(batch + 1, batch, 0, 1), batch + 1, seed=2000, device=0, lanes=lanes)


def run(job, n):
    for _ in range(2):
        job.step()
    job.sync()
    t0 = time.perf_counter()
    for _ in range(n):
        job.step()
    job.sync()
    dt = time.perf_counter() - t0
    return n * batch / dt, dt / n * 1e3


one, two = make(1), make(2)
for _ in range(2):
    print("one lane : %.0f frames/s, %.2f ms/step" % run(one, steps))
    print("two lanes: %.0f frames/s, %.2f ms/step" % run(two, steps))
