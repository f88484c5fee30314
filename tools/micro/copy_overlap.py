#!/usr/bin/env python3
"""How well host<->device copies of two threads (each on its own library stream, tf_thread_stream) overlap: the ceiling of
the drop-in path, which moves ~116 MB up and ~91 MB down per 4K frame.  usage (GPU box): python3 tools/micro/copy_overlap.py"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from transflow_amd import _lib  # noqa: E402
from transflow_amd.device import DevBuffer, pinned_empty  # noqa: E402

lib = _lib.load()
_lib.check(lib.tf_init(0))
MB = 66
n = MB << 20


def worker(kind, pinned, stream, reps, out):
    _lib.check(lib.tf_thread_stream(stream))
    host = pinned_empty((n,), np.uint8) if pinned else np.empty(n, np.uint8)
    host[:] = 1
    dev = DevBuffer(n)
    f = lib.tf_dev_upload if kind == "up" else lib.tf_dev_download
    args = (C.c_void_p(dev.ptr), C.c_void_p(host.ctypes.data), n) if kind == "up" else (C.c_void_p(host.ctypes.data), C.c_void_p(dev.ptr), n)
    f(*args)
    t0 = time.perf_counter()
    for _ in range(reps):
        _lib.check(f(*args))
    out.append((kind, pinned, MB * reps / (time.perf_counter() - t0) / 1e3))


def run(specs, reps=20):
    out, ts = [], []
    for i, (kind, pinned) in enumerate(specs):
        ts.append(threading.Thread(target=worker, args=(kind, pinned, i + 1, reps, out)))
    t0 = time.perf_counter()
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    print(" + ".join(f"{k} {'pinned' if p else 'pageable'}" for k, p in specs), "->",
          ", ".join(f"{k}: {r:.1f} GB/s" for k, p, r in out), f"(wall {dt * 1e3:.0f} ms)")


for spec in ([("up", True)], [("down", True)], [("up", False)], [("down", False)],
             [("up", True), ("down", True)], [("up", True), ("up", True)], [("down", True), ("down", True)],
             [("up", False), ("down", True)], [("up", True), ("down", True), ("up", False)]):
    run(spec)
