#!/usr/bin/env python3
"""Per-kernel, per-level timing of one batched Farneback+remap step (HIP events through
tf_prof_*).  Usage on the GPU box:  python tools/kprof.py [workload] [batch] [option=value ...]"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from transflow_amd import _lib  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "4k"
batch = int(sys.argv[2]) if len(sys.argv) > 2 else 8
_lib.set_option("prof_levels", 1)
reps = 5
for arg in sys.argv[3:]:
    k, v = arg.split("=")
    if k == "reps":
        reps = int(v)          # a longer run (tools/clock_watch.sh samples clocks and power beside it)
    else:
        _lib.set_option(k, int(v))
job = bench.Job(bench.WORKLOADS[name], batch, bench.make_plan(batch + 1, batch, 0, 1), batch + 1, seed=2000, device=0, lanes=1)
for _ in range(2):
    job.step()
job.sync()
job.prof(True)
job.prof_reset()
for _ in range(reps):
    job.step()
job.sync()
rep = job.prof_report()
job.prof(False)
tot = sum(v[1] for v in rep.values())
print(f"{name} batch={batch}: {tot / reps:.3f} ms/step in kernels")
for k, (cnt, ms) in sorted(rep.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:32s} {cnt // reps:4d} launches/step  {ms / reps:8.4f} ms/step  {ms / cnt * 1e3:9.2f} us/launch")
