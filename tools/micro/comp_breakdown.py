#!/usr/bin/env python3
"""Where a frame's time goes in the consumer's thread of the in-process path (HipCompositor.update + render on device
flows, the flow source prefetching beside it): per call, wall time.  usage (GPU box): python3 tools/micro/comp_breakdown.py [4k|1080p] [frames] [idle]
`idle`: the source is exhausted first (flows kept), so nothing runs beside the compositor."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import bench  # noqa: E402
from transflow_amd.compositor import HipCompositor  # noqa: E402
from transflow_amd.config import FlowConfig, LayerConfig  # noqa: E402
from transflow_amd.flow import ArrayFrameProvider, HipFlowSource  # noqa: E402

name = sys.argv[1] if len(sys.argv) > 1 else "4k"
n = int(sys.argv[2]) if len(sys.argv) > 2 else 40
idle = "idle" in sys.argv[3:]
w, h = {"1080p": (1920, 1080), "4k": (3840, 2160)}[name]
clip = bench.ClipSynth(h, w, n + 6, 2000)
frames = [np.stack([f // 2 + 20, f, 255 - (255 - f) // 2], axis=2).astype(np.uint8) for f in (clip.frame(t) for t in range(n + 6))]
pix = np.random.default_rng(1).integers(0, 256, (h, w, 3), dtype=np.uint8)


class Src:
    introduction_mask = np.ones((h, w), bool)

    def next(self, timeout=1):
        return pix


comp = HipCompositor.from_args(h, w, [LayerConfig(0)])
comp.set_sources({0: [Src()]})
layer = comp.layers[0]
from transflow_amd.device import ArrayPool  # noqa: E402
pool = ArrayPool((h, w, 3), np.uint8, pinned=True)
T = dict(next=0.0, update_dev=0.0, gather=0.0, render_k=0.0, download=0.0, check=0.0)
with HipFlowSource.from_args(ArrayFrameProvider(frames, 30.0), direction="backward",
                             cv_config=FlowConfig(hip_prefetch=4, hip_device_flows=True, hip_batch=4)) as source:
    it = iter(source)
    flows = list(it) if idle else None
    k = 0
    for i in range(n + 5):
        t0 = time.perf_counter()
        flow = flows[i] if idle else next(it)
        t1 = time.perf_counter()
        dev = layer._layer()
        dev.update(flow, None, 0)
        t2 = time.perf_counter()
        dev.gather(0, pix, beside=True)
        t3 = time.perf_counter()
        c = comp._image()
        c.begin()
        dev.render(c)
        t4 = time.perf_counter()
        out = c.download(pool.take())
        t5 = time.perf_counter()
        dev.out_of_frame()
        t6 = time.perf_counter()
        if i >= 5:
            k += 1
            for key, d in zip(T, (t1 - t0, t2 - t1, t3 - t2, t4 - t3, t5 - t4, t6 - t5)):
                T[key] += d
print(name, "idle" if idle else "beside the flow source", {k_: round(v / k * 1e3, 3) for k_, v in T.items()}, "ms per frame")
