#!/usr/bin/env python3
"""Per-kernel timing of the rows either side of the hot path with inputs resident
in HBM: HIP events through tf_prof_*, algorithmic bytes per pixel stated per row, achieved GB/s
against the 8 TB/s peak.  Usage on the GPU box:  python tools/bench_next_rows.py [4k|1080p]"""
import ctypes as C
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transflow_amd import _lib  # noqa: E402
from transflow_amd._lib import check  # noqa: E402
from transflow_amd.device import DevBuffer, sync  # noqa: E402
from transflow_amd.remap import CompImage, RemapLayer  # noqa: E402

W, H = {"4k": (3840, 2160), "1080p": (1920, 1080)}[sys.argv[1] if len(sys.argv) > 1 else "4k"]
N = W * H
lib = _lib.load()
check(lib.tf_init(0))
rng = np.random.default_rng(1)
flow = (rng.normal(0, 3, (H, W, 2))).astype(np.float32)
jj, ii = np.meshgrid(np.arange(W), np.arange(H))
flow[:, :, 0] = np.clip(flow[:, :, 0], -jj, W - 1 - jj)
flow[:, :, 1] = np.clip(flow[:, :, 1], -ii, H - 1 - ii)
d_flow = [DevBuffer.from_array(flow) for _ in range(3)]
d_out = DevBuffer(N * 16 * 4)
d_scr = DevBuffer(N * 4)
pix = rng.integers(0, 256, (H, W, 3), dtype=np.uint8)
d_pix = DevBuffer.from_array(pix)
REPS = 10
rows = []


def timed(label, bytes_per_px, fn, px=N):
    fn()
    sync()
    check(lib.tf_prof_reset())
    check(lib.tf_prof_set_filter(None))
    check(lib.tf_prof_enable(1))
    for _ in range(REPS):
        fn()
    sync()
    check(lib.tf_prof_enable(0))
    buf = C.create_string_buffer(1 << 16)
    check(lib.tf_prof_report(buf, len(buf)))
    ms = sum(float(line.split()[2]) for line in buf.value.decode().splitlines()) / REPS
    gbs = bytes_per_px * px / (ms * 1e-3) / 1e9
    rows.append((label, bytes_per_px, ms * 1e3, gbs))


def ptrs(n):
    return (C.c_void_p * n)(*[d_flow[i].ptr for i in range(n)])


P = C.c_void_p
for kind, name in ((1, "merge sum, 3 flows"), (5, "merge maskbin, 3 flows"), (7, "merge absmax, 2 flows")):
    n = 2 if kind == 7 else 3
    timed(name, 8 * (n + 1), lambda k=kind, n=n: check(lib.tf_flow_merge_dev(k, n, ptrs(n), P(d_out.ptr), N * 2)))
timed("upscale x2,x2 (per OUTPUT pixel)", 8 + 2, lambda: check(lib.tf_flow_upscale_dev(P(d_flow[0].ptr), P(d_out.ptr), W // 2, H // 2, 2, 2)))
for (kh, kw, wide) in ((3, 3, 1), (5, 5, 1), (3, 3, 0)):
    k = np.ascontiguousarray(rng.normal(0, 0.3, (kh, kw)), dtype=np.float64 if wide else np.float32)
    dk = DevBuffer.from_array(k)
    timed(f"convolve {kh}x{kw} {'f64' if wide else 'f32'}", 8 + (16 if wide else 8),
          lambda dk=dk, kh=kh, kw=kw, wide=wide: check(lib.tf_flow_convolve_dev(P(d_flow[0].ptr), P(dk.ptr), kh, kw, wide, P(d_out.ptr), W, H)))
timed("post_process f64 BACKWARD", 32, lambda: check(lib.tf_flow_post_process_dev(P(d_out.ptr), 1, W, H, 1, P(d_scr.ptr))))
timed("post_process f64 FORWARD", 32 + 16 + 8, lambda: check(lib.tf_flow_post_process_dev(P(d_out.ptr), 1, W, H, 0, P(d_scr.ptr))))
col2 = (C.c_float * 6)(0, 0, 0, 255, 255, 255)
col4 = (C.c_float * 12)(255, 255, 0, 0, 0, 255, 255, 0, 255, 0, 255, 0)
timed("render1d", 4 + 3, lambda: check(lib.tf_flow_render1d_dev(P(d_flow[0].ptr), P(d_out.ptr), N, 0.3, col2, 0)))
timed("render2d", 8 + 3, lambda: check(lib.tf_flow_render2d_dev(P(d_flow[0].ptr), P(d_out.ptr), N, 0.1, col4)))
timed("bgr -> grey (same size)", 3 + 1, lambda: check(lib.tf_frame_grey_dev(P(d_pix.ptr), W, H, P(d_out.ptr), W, H)))

# layer classes: one frame = update + per-source step + render (one source, RGB pixmap)
comp = CompImage(H, W)
d_f = d_flow[0].ptr
for cls, bpp in (("sum", 8 + 32 + 16 + 3 + 8 + 3), ("static", 1 + 3 + 8 + 4 + 3), ("introduction", 8 + 64 + 64 + 1 + 64 + 3 + 32 + 3)):
    layer = RemapLayer(H, W, layer_class=cls)
    layer.set_sources([np.ones((H, W), np.uint8)])

    def frame(layer=layer, cls=cls):
        layer.update_dev(d_f)
        if cls == "introduction":
            check(lib.tf_remap_introduce_dev(layer._h, 0, P(d_pix.ptr), 3, 7))
        else:
            layer.gather_dev(0, d_pix.ptr, 3)
        comp.begin()
        layer.render(comp)
    timed(f"{cls} layer, one frame (all its kernels)", bpp, frame)
    layer.close()

print(f"{W}x{H}, {REPS} repetitions, HIP events per kernel; GB/s on the algorithmic bytes per pixel given")
print(f"{'row':44s} {'B/px':>6s} {'us':>9s} {'GB/s':>8s} {'of 8 TB/s':>9s}")
for label, bpp, us, gbs in rows:
    print(f"{label:44s} {bpp:6d} {us:9.1f} {gbs:8.0f} {gbs / 8000:9.1%}")
