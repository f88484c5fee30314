#!/usr/bin/env python3
"""How fast does a consumer process read a producer's device buffer through a HIP IPC mapping on the same GPU?
(The two-process pipeline at 4K ran 70 frames/s where the in-process one runs 750: tools/stress_ipc_pipeline.py.)
A forked child allocates and exports a 66 MB buffer; the parent opens it and times, per 66 MB: hipMemcpyAsync device to
device out of the mapping (tf_dev_copy), the 16-byte-per-lane copy kernel out of the mapping (tf_dev_stream_copy), and
the same two between buffers of its own.   usage (GPU box): python3 tools/micro/ipc_copy_rate.py"""
import ctypes as C
import multiprocessing as mp
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
N = 3840 * 2160 * 8


def child(q, done):
    from transflow_amd import _lib
    from transflow_amd.device import DevBuffer
    lib = _lib.load()
    _lib.check(lib.tf_init(0))
    buf = DevBuffer(N)
    h = (C.c_char * 64)()
    _lib.check(lib.tf_ipc_export(C.c_void_p(buf.ptr), h))
    _lib.check(lib.tf_sync())
    q.put(bytes(h.raw))
    done.wait(300)


def main():
    ctx = mp.get_context("fork")
    q, done = ctx.Queue(), ctx.Event()
    p = ctx.Process(target=child, args=(q, done))
    p.start()
    handle = q.get(timeout=120)
    from transflow_amd import _lib
    from transflow_amd.device import DevBuffer
    lib = _lib.load()
    _lib.check(lib.tf_init(0))
    src = C.c_void_p()
    _lib.check(lib.tf_ipc_open(handle, C.byref(src)))
    own_a, own_b = DevBuffer(N), DevBuffer(N)

    def rate(fn, reps=10):
        fn()
        _lib.check(lib.tf_sync())
        t0 = time.perf_counter()
        for _ in range(reps):
            fn()
        _lib.check(lib.tf_sync())
        dt = (time.perf_counter() - t0) / reps
        return f"{dt * 1e3:8.3f} ms per 66 MB = {N / dt / 1e9:8.1f} GB/s"

    print("hipMemcpyAsync D2D out of the IPC mapping :", rate(lambda: _lib.check(lib.tf_dev_copy(C.c_void_p(own_a.ptr), src, N))))
    print("copy kernel out of the IPC mapping         :", rate(lambda: _lib.check(lib.tf_dev_stream_copy(C.c_void_p(own_a.ptr), src, N))))
    print("hipMemcpyAsync D2D between own buffers     :", rate(lambda: _lib.check(lib.tf_dev_copy(C.c_void_p(own_a.ptr), C.c_void_p(own_b.ptr), N))))
    print("copy kernel between own buffers            :", rate(lambda: _lib.check(lib.tf_dev_stream_copy(C.c_void_p(own_a.ptr), C.c_void_p(own_b.ptr), N))))
    done.set()
    p.join(60)


if __name__ == "__main__":
    main()
