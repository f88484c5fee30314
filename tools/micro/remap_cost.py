#!/usr/bin/env python3
"""What does the remap -- a serial chain of per-frame launches beside the next pass's Farneback kernels -- cost the step?
The bench's step with and without its remap launches.  usage (GPU box): python3 tools/micro/remap_cost.py [workload] [batch] [steps]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "4k"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 128
steps = int(sys.argv[3]) if len(sys.argv) > 3 else 6
job = bench.Job(bench.WORKLOADS[name], batch, bench.make_plan(batch + 1, batch, 0, 1), batch + 1, seed=2000, device=0, lanes=1)


def run(with_remap):
    def one():
        job.calc_pass(job.n_steps)
        job.n_steps += 1
        if with_remap:
            for i in range(job.batch):
                job.remap_pair(job.layer, job.comps[i], i)
    for _ in range(2):
        one()
    job.sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    job.sync()
    return (time.perf_counter() - t0) / steps * 1e3


for _ in range(2):
    a, b = run(True), run(False)
    print(f"{name} x {batch}: {a:.2f} ms per pass with the remap, {b:.2f} ms without: the remap costs {a - b:.2f} ms "
          f"({batch / a * 1e3:.0f} against {batch / b * 1e3:.0f} frames/s)")
