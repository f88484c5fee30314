#!/usr/bin/env python3
"""The drop-in (host-array) path end to end: what a transflow pipeline sees per frame when flows and
frames cross PCIe as numpy arrays -- HipFlowSource.next + post_process, HipCompositor.update + render.
This is the PCIe-inclusive rate DESIGN.md section 5 quotes; it is never bench.py's `value`.  The frames the
source reads are decoded BGR frames, as a video capture hands them over (cv.py:461-466): the upload of the colour
frame and its conversion to grey on the device are inside the measured time (`grey` as a third argument feeds
ready-made grey frames instead, the form the round-2 figures were taken with).
`prefetch` lets the flow source run two flows ahead in a worker thread with a stream of its own (FlowConfig.hip_prefetch --
what the reference's child process + queue give it, pipeline.py:56-64): flow t + 1 is computed while the compositor
works on flow t.
`device` (FlowConfig.hip_device_flows): the flows stay in HBM between the source and the compositor (DeviceFlow) -- the
frame still goes up and the rendered frame still comes down, the 66 MB per 4K flow no longer travel at all.
`frame` (HipCompositor lazy_frames): render() returns a DeviceFrame whose download is under way; the loop reads frame
t - 1 (as the reference's output process would, behind its queue) after it has issued frame t, so a frame's 25 MB come
down beside the next frame's uploads.
Usage on the GPU box:  python tools/bench_host_path.py [1080p|4k] [frames] [bgr|grey] [exact] [prefetch] [device] [frame] [batch=n] [reps=r]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from transflow_amd.compositor import HipCompositor  # noqa: E402
from transflow_amd.config import LayerConfig  # noqa: E402
from transflow_amd.flow import ArrayFrameProvider, HipFlowSource  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "1080p"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 12
w, h = {"1080p": (1920, 1080), "4k": (3840, 2160)}[name]
clip = bench.ClipSynth(h, w, n + 6, 2000)
frames = [clip.frame(t) for t in range(n + 6)]
kind = sys.argv[3] if len(sys.argv) > 3 else "bgr"
exact = "exact" in sys.argv[4:]          # flows bit-identical to the CPU path's (option fb_exact_sums)
prefetch = "prefetch" in sys.argv[4:]
device = "device" in sys.argv[4:]
lazy = "frame" in sys.argv[4:]
if "nobeside" in sys.argv[4:]:        # A/B: the pixmap's upload on the caller's stream even behind a device-flow update
    from transflow_amd import remap as _remap
    _g = _remap.RemapLayer.gather
    _remap.RemapLayer.gather = lambda self, i, pm, beside=False: _g(self, i, pm, False)
batch = next((int(a.split("=")[1]) for a in sys.argv[4:] if a.startswith("batch=")), 1)   # FlowConfig.hip_batch
if kind == "bgr":   # three channels around the texture, so that the grey value still carries it
    frames = [np.stack([f // 2 + 20, f, 255 - (255 - f) // 2], axis=2).astype(np.uint8) for f in frames]
reps = next((int(a.split("=")[1]) for a in sys.argv[4:] if a.startswith("reps=")), 1)
if reps > 1:            # a longer run over the same frames, there and back again (no jump between the last and the first)
    frames = (frames + frames[-2:0:-1]) * reps + frames[:1]
    n = len(frames) - 6
pix = np.random.default_rng(1).integers(0, 256, (h, w, 3), dtype=np.uint8)


class Src:
    introduction_mask = np.ones((h, w), bool)

    def next(self, timeout=1):
        return pix


comp = HipCompositor.from_args(h, w, [LayerConfig(0)], lazy_frames=lazy)
comp.set_sources({0: [Src()]})
t_flow = t_comp = 0.0
cfg = None
if exact or prefetch or device or batch > 1:
    from transflow_amd.config import FlowConfig  # noqa: E402
    cfg = FlowConfig(hip_exact_sums=exact, hip_device_flows=device, hip_batch=batch,
                     **({"hip_prefetch": max(2, batch)} if prefetch else {}))
with HipFlowSource.from_args(ArrayFrameProvider(frames, 30.0), direction="backward", cv_config=cfg) as source:
    it = iter(source)
    for _ in range(6):                   # warm-up: handle creation, first launches, the pools of page-locked arrays
        flow = next(it)
        comp.update(flow)
        comp.render()
    k, img = 0, None
    while True:
        t0 = time.perf_counter()
        try:
            flow = next(it)
        except StopIteration:
            break
        t1 = time.perf_counter()
        comp.update(flow)
        img, before = comp.render(), (img if k else None)
        if lazy and before is not None:
            checksum = int(np.asarray(before)[0, 0, 0])      # the frame before this one: in host memory now (waits if not)
        t2 = time.perf_counter()
        t_flow += t1 - t0
        t_comp += t2 - t1
        k += 1
print(f"{name} ({kind} frames in{', exact sums' if exact else ''}{', flow source prefetching' if prefetch else ''}{', flows stay on the device' if device else ''}{', frames read one late (lazy)' if lazy else ''}{f', {batch} pairs per call' if batch > 1 else ''}): {k} frames; flow source {t_flow / k * 1e3:.1f} ms/frame, compositor {t_comp / k * 1e3:.1f} ms/frame, "
      f"{k / (t_flow + t_comp):.1f} frames/s end to end {'(frames in, frames out)' if device else 'through host arrays'} (one process)")
