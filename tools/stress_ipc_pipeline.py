#!/usr/bin/env python3
"""The reference's process layout under sustained load: a FORKED flow process (pipeline.py:56-101) whose flows cross a
multiprocessing.Queue(maxsize=1) as HIP IPC tokens (FlowConfig.hip_device_flows = "ipc", look-ahead and prefetch on), the
compositor in the parent with lazy frames.  Every frame's 64-bit digest is compared with the same clip run in ONE process through
plain host arrays; the producer reports its ring (buffers, exports, what was left unacknowledged when it let go).
usage (GPU box):  python3 tools/stress_ipc_pipeline.py [frames=300] [size=960x540] > gpurun_out/r06_stress_ipc.txt"""
import multiprocessing as mp
import os
import sys
import time
import zlib

try:
    import xxhash

    def digest(a):
        return xxhash.xxh3_64_intdigest(memoryview(np.ascontiguousarray(a)).cast("B"))
except ImportError:      # (25 MB through zlib.crc32 is 13 ms: slower than the frame it checks)
    def digest(a):
        return zlib.crc32(np.ascontiguousarray(a).tobytes())

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402


def make_frames(h, w, n):
    clip = bench.ClipSynth(h, w, n, 2000)
    return [np.stack([f // 2 + 20, f, 255 - (255 - f) // 2], axis=2).astype(np.uint8) for f in (clip.frame(t) for t in range(n))]


def producer(frames, queue, meta):
    from transflow_amd.config import FlowConfig
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    try:
        # the child's own copies, as a video capture inside the child would hand them over: the forked pages are shared
        # copy-on-write with the parent, and pinning such a page for a transfer first copies it
        frames = [f.copy() for f in frames]
        cfg = FlowConfig(hip_device_flows="ipc", hip_batch=4, hip_prefetch=4)
        report = None
        with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward", cv_config=cfg) as source:
            meta.put((source.width, source.height, source.length))
            t_next = t_put = 0.0
            it = iter(source)
            while True:
                t0 = time.perf_counter()
                try:
                    flow = next(it)
                except StopIteration:
                    break
                t1 = time.perf_counter()
                queue.put(flow)
                t2 = time.perf_counter()
                t_next += t1 - t0
                t_put += t2 - t1
            del flow
            print(f"producer: {t_next:.2f} s in next(source), {t_put:.2f} s in queue.put", flush=True)
            ring = source._flow_ring
            source.prev_flow = None
            report = (bool(ring.drain(timeout=120.0)), len(ring._all), ring.exports, ring.unacknowledged())
        meta.put(report)
        queue.put(None)
    except Exception as err:      # noqa: BLE001 -- surfaces in the parent
        queue.put(err)


class Pixmaps:
    def __init__(self, h, w):
        self.introduction_mask = np.ones((h, w), bool)
        self.pix = [np.random.default_rng(50 + i).integers(0, 256, (h, w, 3), dtype=np.uint8) for i in range(3)]
        self.n = 0

    def next(self, timeout=1):
        self.n += 1
        return self.pix[self.n % 3]


def main():
    n = next((int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("frames=")), 300)
    w, h = (int(v) for v in next((a.split("=")[1] for a in sys.argv[1:] if a.startswith("size=")), "960x540").split("x"))
    frames = make_frames(h, w, n)
    ctx = mp.get_context("fork")
    queue, meta = ctx.Queue(maxsize=1), ctx.Queue()
    child = ctx.Process(target=producer, args=(frames, queue, meta))
    child.start()                           # forked before this process touches the GPU
    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import LayerConfig
    from transflow_amd.deviceflow import DeviceFlow
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    assert meta.get(timeout=300) == (w, h, n - 1)
    comp = HipCompositor.from_args(h, w, [LayerConfig(0, reset_mode="random", reset_random_factor=0.02)], rng="device", lazy_frames=True)
    comp.set_sources({0: [Pixmaps(h, w)]})
    crcs, held, tokens = [], None, 0
    t0 = t_steady = time.perf_counter()
    t_get = t_upd = t_ren = t_crc = 0.0
    while True:
        if len(crcs) == 20:
            t_steady = time.perf_counter()      # (the first flows carry the handle's creation and the pools' first arrays)
        a = time.perf_counter()
        item = queue.get(timeout=300)
        b = time.perf_counter()
        if item is None:
            break
        if isinstance(item, Exception):
            raise item
        tokens += isinstance(item, DeviceFlow) and item._host is None
        comp.update(item)
        c = time.perf_counter()
        frame = comp.render()
        d = time.perf_counter()
        if held is not None:
            crcs.append(digest(np.asarray(held)))
        held = frame
        e = time.perf_counter()
        t_get += b - a
        t_upd += c - b
        t_ren += d - c
        t_crc += e - d
    print(f"consumer: {t_get:.2f} s in queue.get, {t_upd:.2f} s in update, {t_ren:.2f} s in render, {t_crc:.2f} s reading the frame before (digest)")
    crcs.append(digest(np.asarray(held)))
    dt = time.perf_counter() - t0
    steady = (len(crcs) - 20) / (time.perf_counter() - t_steady) if len(crcs) > 40 else float("nan")
    drained, buffers, exports, pending = meta.get(timeout=300)
    child.join(timeout=120)
    print(f"two processes, {w}x{h}: {len(crcs)} frames in {dt:.2f} s ({len(crcs) / dt:.0f} frames/s; {steady:.0f} from frame 20 on), {tokens} flows crossed as IPC tokens; "
          f"producer: {exports} exports, ring of {buffers} buffers, drained {drained}, unacknowledged at the end {pending}, exit code {child.exitcode}")
    # the same clip in one process through plain host arrays
    comp2 = HipCompositor.from_args(h, w, [LayerConfig(0, reset_mode="random", reset_random_factor=0.02)], rng="device")
    comp2.set_sources({0: [Pixmaps(h, w)]})
    ref = []
    with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward") as source:
        for flow in source:
            comp2.update(flow)
            ref.append(digest(comp2.render()))
    same = sum(a == b for a, b in zip(crcs, ref))
    ok = len(crcs) == len(ref) == n - 1 and same == len(ref) and drained and not pending and child.exitcode == 0 and tokens == n - 1
    print(f"one process, host arrays: {len(ref)} frames; {same} of {len(ref)} frames have the digest of the two-process run")
    print("# OK" if ok else "# FAILED")
    sys.exit(0 if ok else 1)


if __name__ == "__main__":
    main()
