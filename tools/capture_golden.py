#!/usr/bin/env python3
"""Generate tests/golden/remap_*.npz by RUNNING the reference's own code.

Run in the build container only (``/root/reference`` does not exist on the
GPU box):

    PYTHONDONTWRITEBYTECODE=1 python tools/capture_golden.py

It imports the importable half of the reference (the
compositor, ``FlowSource.post_process`` and their helpers), feeds seeded
inputs, and stores inputs + the reference's outputs as small ``.npz`` files.
Only DATA is written to the repo -- no reference source.  The Farnebäck half
cannot be captured (it lives in cv2, absent here): see oracle/farneback_ref.c.
"""
import os
import sys

import numpy as np

REF = "/root/reference"
sys.path.insert(0, REF)
sys.dont_write_bytecode = True

from transflow.compositor import Compositor  # noqa: E402
from transflow.compositor.layers.move_reference import MoveReferenceLayer  # noqa: E402
from transflow.config import LayerConfig  # noqa: E402
from transflow.flow.sources.source import FlowSource  # noqa: E402

OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")
os.makedirs(OUT, exist_ok=True)


class FakeSource:
    """Stands in for PixmapSourceInterface (pixmap_source_interface.py:12-37)."""

    def __init__(self, frames, introduction_mask):
        self.frames = list(frames)
        self.introduction_mask = introduction_mask
        self.counter = -1

    def next(self, timeout=1):
        self.counter += 1
        return self.frames[self.counter % len(self.frames)]

    @property
    def frame_number(self):
        return self.counter


def make_fs(direction, h, w):
    return FlowSource(direction, w, h, 30.0, None, 0, 0, 0)


def capture_post_process():
    out = {}
    rng = np.random.default_rng(101)
    idx = 0
    for (h, w) in [(1, 6), (16, 24), (37, 53)]:
        for sigma in [0.4, 3.0, 40.0]:
            for direction in (FlowSource.Direction.FORWARD, FlowSource.Direction.BACKWARD):
                flow = rng.normal(0, sigma, (h, w, 2)).astype(np.float32)
                fs = make_fs(direction, h, w)
                res = fs.post_process(flow.copy())
                out[f"in_{idx}"] = flow
                out[f"out_{idx}"] = np.asarray(res, dtype=np.float32)
                out[f"dir_{idx}"] = np.int32(direction.value)
                idx += 1
    # ties: exact halves exercise round-half-even; collisions exercise last-write-wins
    h, w = 5, 9
    flow = np.zeros((h, w, 2), np.float32)
    vals = [0.5, 1.5, 2.5, -0.5, -1.5, -2.5, 3.5, -3.5, 0.49999997]
    for j, v in enumerate(vals):
        flow[1, j, 0] = v
        flow[2, j, 1] = v
        flow[3, j, :] = (v, -v)
    flow[0, :, 0] = np.arange(w)[::-1] - np.arange(w)  # every pixel of row 0 lands mirrored
    flow[4, :, 0] = 4 - np.arange(w)                   # whole row collides on column 4
    for direction in (FlowSource.Direction.FORWARD, FlowSource.Direction.BACKWARD):
        fs = make_fs(direction, h, w)
        res = fs.post_process(flow.copy())
        out[f"in_{idx}"] = flow
        out[f"out_{idx}"] = np.asarray(res, dtype=np.float32)
        out[f"dir_{idx}"] = np.int32(direction.value)
        idx += 1
    out["count"] = np.int32(idx)
    np.savez_compressed(os.path.join(OUT, "remap_post_process.npz"), **out)
    print("post_process cases:", idx)


def smooth_mask(rng, h, w):
    yy, xx = np.mgrid[0:h, 0:w]
    m = 0.5 + 0.5 * np.sin(xx / 3.1 + rng.uniform(0, 6)) * np.cos(yy / 2.3 + rng.uniform(0, 6))
    return m.astype(np.float32)


def run_layer_case(name, h, w, cfg_kwargs, nframes, rng, sigma=2.5, rgba_pixmap=False,
                   two_sources=False, masks=None, direction=FlowSource.Direction.BACKWARD,
                   background="#ffffff", classname="moveref"):
    """Runs one layer recurrence (MoveReferenceLayer unless `classname` says otherwise) through
    the reference and returns a dict of inputs and per-frame outputs."""
    from transflow.compositor.layers.layer import Layer
    masks = masks or {}
    cfg = LayerConfig(0, classname=classname, **cfg_kwargs)
    layer = Layer.from_args(cfg, h, w, [])
    # arbitrary mask arrays are installed on the layer object directly; the
    # reference's loaders (utils.py:51-144) are host-side parsing, not the path
    for k, v in masks.items():
        setattr(layer, k, v)
    c = 4 if rgba_pixmap else 3
    intro = [np.ones((h, w), dtype=bool)]
    if two_sources:
        m1 = np.zeros((h, w), dtype=bool)
        m1[:, w // 2:] = True
        m1[h // 3: h // 2, : w // 4] = True
        intro.append(m1)
    pixmaps = [rng.integers(0, 256, (nframes, h, w, c), dtype=np.uint8) for _ in intro]
    sources = [FakeSource(list(pm), im) for pm, im in zip(pixmaps, intro)]
    layer.set_sources(sources)
    comp = Compositor(h, w, [layer], background_color=background)
    fs = make_fs(direction, h, w)
    d = {
        "h": np.int32(h), "w": np.int32(w), "nframes": np.int32(nframes),
        "nsources": np.int32(len(intro)),
        "classname": np.array(classname),
        "mask_alpha": layer.mask_alpha.copy(),
        "background": np.array(comp.background_color, dtype=np.uint8),
        "cfg_keys": np.array(sorted(cfg_kwargs.keys())),
        "cfg_vals": np.array([str(cfg_kwargs[k]) for k in sorted(cfg_kwargs.keys())]),
    }
    for key in ("mask_src", "mask_dst", "reset_mask"):
        if hasattr(layer, key):
            d[key] = getattr(layer, key).copy()
    if hasattr(layer, "data"):
        d["data_init"] = layer.data.copy()
    d["rgba_init"] = np.array(layer.rgba, copy=True)
    for s, (pm, im) in enumerate(zip(pixmaps, intro)):
        d[f"pixmap_{s}"] = pm
        d[f"intro_{s}"] = im
    orig_random = np.random.random
    for t in range(nframes):
        raw = rng.normal(0, sigma, (h, w, 2)).astype(np.float32)
        flow = np.asarray(fs.post_process(raw.copy()), dtype=np.float32)
        u = rng.random((h, w))
        np.random.random = lambda size=None, _u=u: _u.copy()
        try:
            comp.update(flow)
        finally:
            np.random.random = orig_random
        d[f"flow_{t}"] = flow
        d[f"u_{t}"] = u
        if hasattr(layer, "data"):
            d[f"data_{t}"] = layer.data.copy()
        d[f"rgba_{t}"] = np.array(layer.rgba, copy=True)
        d[f"frame_{t}"] = comp.render().copy()
        d[f"rgba_after_render_{t}"] = np.array(layer.rgba, copy=True)
        if hasattr(layer, "data"):
            d[f"data_after_render_{t}"] = layer.data.copy()
    return d


def capture_layers():
    rng = np.random.default_rng(202)
    cases = {}
    # 1. the 2^4 move flags, default masks, no reset (16x24, 3 frames, RGB).
    #    Transparent pixels only exist once leave_empty_spot made some.
    for bits in range(16):
        kw = dict(
            transparent_pixels_can_move=bool(bits & 1),
            pixels_can_move_to_empty_spot=bool(bits & 2),
            pixels_can_move_to_filled_spot=bool(bits & 4),
            moving_pixels_leave_empty_spot=bool(bits & 8),
        )
        cases[f"flags{bits:02d}"] = run_layer_case(f"flags{bits}", 16, 24, kw, 3, rng)
    # 2. the same flags with an alpha pattern that already has holes (via reset off and
    #    a first frame run with leave_empty) -- longer recurrence, odd size
    for bits in (8, 9, 10, 12, 13, 15):
        kw = dict(
            transparent_pixels_can_move=bool(bits & 1),
            pixels_can_move_to_empty_spot=bool(bits & 2),
            pixels_can_move_to_filled_spot=bool(bits & 4),
            moving_pixels_leave_empty_spot=bool(bits & 8),
        )
        cases[f"holes{bits:02d}"] = run_layer_case(f"holes{bits}", 37, 53, kw, 5, rng, sigma=4.0)
    # 3. masks
    h, w = 37, 53
    msrc = rng.random((h, w)) < 0.7
    mdst = rng.random((h, w)) < 0.6
    malpha = smooth_mask(rng, h, w)
    cases["masks_bool"] = run_layer_case("masks_bool", h, w, {}, 4, rng,
                                         masks={"mask_src": msrc, "mask_dst": mdst})
    cases["masks_alpha"] = run_layer_case("masks_alpha", h, w, {}, 3, rng,
                                          masks={"mask_alpha": malpha}, background="#204060")
    cases["masks_spec"] = run_layer_case(
        "masks_spec", h, w,
        dict(mask_src="border:2", mask_dst="circle:40%", mask_alpha="rect:60%:inv"), 3, rng)
    # 4. random reset
    for p in (0.1, 0.5, 1.0):
        rm = smooth_mask(rng, h, w)
        cases[f"reset_random_p{int(p * 10):02d}"] = run_layer_case(
            "reset", h, w, dict(reset_mode="random", reset_random_factor=p), 4, rng,
            masks={"reset_mask": rm})
    cases["reset_random_leave"] = run_layer_case(
        "reset_leave", h, w,
        dict(reset_mode="random", reset_random_factor=0.3, moving_pixels_leave_empty_spot=True,
             transparent_pixels_can_move=True), 5, rng,
        masks={"reset_mask": rng.random((h, w)).astype(np.float32)})
    cases["reset_random_source"] = run_layer_case(
        "reset_source", h, w,
        dict(reset_mode="random", reset_random_factor=0.4, reset_source=True), 4, rng,
        two_sources=True, masks={"reset_mask": smooth_mask(rng, h, w)})
    # 5. pixmaps: RGBA, two sources (RGB quirk: only the last source stays opaque)
    cases["rgba_pixmap"] = run_layer_case("rgba", 16, 24, {}, 3, rng, rgba_pixmap=True)
    cases["two_sources_rgb"] = run_layer_case("two_rgb", 16, 24, {}, 3, rng, two_sources=True)
    cases["two_sources_rgba"] = run_layer_case("two_rgba", 16, 24, {}, 3, rng, two_sources=True,
                                               rgba_pixmap=True)
    # 6. FORWARD-direction flows feeding the layer
    cases["forward_dir"] = run_layer_case("fwd", h, w, {}, 4, rng,
                                          direction=FlowSource.Direction.FORWARD, sigma=3.0)
    # 7. other reset modes
    cases["reset_constant"] = run_layer_case(
        "reset_constant", h, w, dict(reset_mode="constant", reset_constant_step=1.7), 5, rng,
        masks={"reset_mask": smooth_mask(rng, h, w)})
    cases["reset_linear"] = run_layer_case(
        "reset_linear", h, w, dict(reset_mode="linear", reset_linear_factor=0.35), 5, rng,
        masks={"reset_mask": smooth_mask(rng, h, w)})
    # 8. bigger frame, default transflow configuration
    cases["default_72x96"] = run_layer_case("default", 72, 96, {}, 2, rng, sigma=5.0)
    for name, d in cases.items():
        np.savez_compressed(os.path.join(OUT, f"remap_layer_{name}.npz"), **d)
    print("layer cases:", len(cases))


def capture_other_layers():
    """the sum, static and introduction layer classes."""
    rng = np.random.default_rng(303)
    cases = {}
    h, w = 37, 53
    # SumLayer (sum.py:7-14): accumulates floor(flow) into (i, j), then the reference layer's reset + gather
    cases["sum_default"] = run_layer_case("sum", h, w, {}, 4, rng, classname="sum")
    cases["sum_reset_random"] = run_layer_case(
        "sum_rr", h, w, dict(reset_mode="random", reset_random_factor=0.3), 4, rng, classname="sum",
        masks={"reset_mask": smooth_mask(rng, h, w)})
    cases["sum_reset_linear_two"] = run_layer_case(
        "sum_rl", h, w, dict(reset_mode="linear", reset_linear_factor=0.4), 4, rng, classname="sum",
        two_sources=True, rgba_pixmap=True, masks={"mask_alpha": smooth_mask(rng, h, w)})
    cases["sum_reset_constant"] = run_layer_case(
        "sum_rc", 16, 24, dict(reset_mode="constant", reset_constant_step=2.2), 5, rng, classname="sum", sigma=4.0)
    # StaticLayer (static.py:7-17)
    cases["static_rgb"] = run_layer_case("static", 16, 24, {}, 3, rng, classname="static")
    cases["static_rgba_two"] = run_layer_case("static2", h, w, {}, 3, rng, classname="static", two_sources=True,
                                              rgba_pixmap=True, masks={"mask_alpha": smooth_mask(rng, h, w)},
                                              background="#102030")
    cases["static_rgb_two_alpha"] = run_layer_case("static3", h, w, {}, 3, rng, classname="static", two_sources=True,
                                                   masks={"mask_alpha": smooth_mask(rng, h, w)})
    # IntroductionLayer (introduction.py:8-67): every introduce_* flag alone, then combinations
    flags = ["introduce_pixels_on_empty_spots", "introduce_pixels_on_filled_spots", "introduce_moving_pixels",
             "introduce_unmoving_pixels", "introduce_once", "introduce_on_all_filled_spots",
             "introduce_on_all_empty_spots"]
    defaults = dict(zip(flags, [True, True, True, True, False, False, False]))
    cases["intro_default"] = run_layer_case("intro", h, w, {}, 4, rng, classname="introduction")
    for f in flags:
        kw = {f: not defaults[f]}
        cases[f"intro_{f}"] = run_layer_case("intro_" + f, 16, 24, kw, 4, rng, classname="introduction", sigma=2.0)
    cases["intro_once_leave_two_rgba"] = run_layer_case(
        "intro_combo", h, w, dict(introduce_once=True, moving_pixels_leave_empty_spot=True,
                                  transparent_pixels_can_move=True), 5, rng,
        classname="introduction", two_sources=True, rgba_pixmap=True, sigma=3.0)
    cases["intro_partial_masks"] = run_layer_case(
        "intro_masks", h, w, dict(introduce_moving_pixels=False, pixels_can_move_to_filled_spot=False), 5, rng,
        classname="introduction", two_sources=True,
        masks={"mask_alpha": smooth_mask(rng, h, w), "mask_src": rng.random((h, w)) < 0.8})
    cases["intro_filled_only_leave"] = run_layer_case(
        "intro_fl", h, w, dict(introduce_pixels_on_filled_spots=False, moving_pixels_leave_empty_spot=True), 5, rng,
        classname="introduction", sigma=3.0)
    for name, d in cases.items():
        np.savez_compressed(os.path.join(OUT, f"layer2_{name}.npz"), **d)
    print("other-layer cases:", len(cases))


def capture_flow_ops():
    """flow merging (pipeline.py:149-158 + utils.py:359-381), upscale_array
    (utils.py:417-418), the convolution-kernel pre-step of post_process (source.py:344-348) and the
    flow visualisation (output/render.py:9-48).  transflow.pipeline needs typing.Self (Python 3.11):
    taken from typing_extensions for the import."""
    import typing
    import typing_extensions
    if not hasattr(typing, "Self"):
        typing.Self = typing_extensions.Self
    from transflow.pipeline import Pipeline
    from transflow.utils import upscale_array
    from transflow.output.render import render1d, render2d
    rng = np.random.default_rng(404)
    d = {}
    # ---- merging
    n_case = 0
    for (h, w) in ((12, 16), (23, 29)):
        for n in (1, 2, 3, 5):
            flows = [rng.normal(0, 1.5, (h, w, 2)).astype(np.float32) for _ in range(n)]
            for f in flows[1:]:
                f[rng.random((h, w, 2)) < 0.3] = 0          # exact zeros and small values for the masks
                f[rng.random((h, w, 2)) < 0.1] = np.float32(0.2)
            for kind, fn in Pipeline.FLOW_MERGING_FUNCTIONS.items():
                if kind == "absmax" and n != 2:
                    continue
                ins = [f.copy() for f in flows]
                out = np.asarray(fn(ins))
                d[f"merge_{n_case}_kind"] = np.array(kind)
                d[f"merge_{n_case}_n"] = np.int32(n)
                for i, f in enumerate(flows):
                    d[f"merge_{n_case}_in{i}"] = f
                d[f"merge_{n_case}_out"] = out
                n_case += 1
    d["merge_cases"] = np.int32(n_case)
    # ---- upscale
    ups = [(1, 1), (2, 2), (3, 2), (1, 4)]
    for i, (wf, hf) in enumerate(ups):
        a = rng.normal(0, 3, (11, 17, 2)).astype(np.float32)
        d[f"up_{i}_in"], d[f"up_{i}_f"] = a, np.array([wf, hf], np.int32)
        d[f"up_{i}_out"] = upscale_array(a, wf, hf)
    d["up_cases"] = np.int32(len(ups))
    # ---- convolution kernel inside post_process
    kernels = [
        np.full((3, 3), 1 / 9.0),                                        # float64 box
        rng.normal(0, 0.3, (5, 3)),                                      # float64, odd x odd, not symmetric
        rng.normal(0, 0.3, (4, 4)),                                      # even sizes: 'same' centring
        rng.normal(0, 0.3, (3, 3)).astype(np.float32),                   # float32 kernel: float32 result
        np.array([[2.0]]),                                               # 1x1
        np.array([[0, 1, 0], [1, -4, 1], [0, 1, 0]], dtype=np.int64),    # integer kernel: float64 result
        rng.normal(0, 0.2, (1, 7)),                                      # row kernel
    ]
    n_case = 0
    for k in kernels:
        for direction in (FlowSource.Direction.FORWARD, FlowSource.Direction.BACKWARD):
            h, w = (24, 31) if n_case % 2 else (37, 53)
            fs = make_fs(direction, h, w)
            fs.kernel = k
            raw = rng.normal(0, 3.0, (h, w, 2)).astype(np.float32)
            out = np.asarray(fs.post_process(raw.copy()))
            d[f"conv_{n_case}_kernel"] = k
            d[f"conv_{n_case}_dir"] = np.int32(0 if direction == FlowSource.Direction.FORWARD else 1)
            d[f"conv_{n_case}_in"] = raw
            d[f"conv_{n_case}_out"] = out
            n_case += 1
    d["conv_cases"] = np.int32(n_case)
    # ---- render1d / render2d
    n_case = 0
    for scale, colors, binary in ((1, None, False), (0.25, ("#102030", "#f0e0d0"), False), (0.5, None, True),
                                  (2.0, ("#ff0000", "#00ff80"), True)):
        arr = np.abs(rng.normal(0, 2, (19, 23))).astype(np.float32)
        d[f"r1_{n_case}_in"] = arr
        d[f"r1_{n_case}_scale"] = np.float64(scale)
        d[f"r1_{n_case}_colors"] = np.array(colors if colors else ("#000000", "#ffffff"))
        d[f"r1_{n_case}_binary"] = np.bool_(binary)
        d[f"r1_{n_case}_out"] = render1d(arr, scale=scale, colors=colors, binary=binary)
        n_case += 1
    d["r1_cases"] = np.int32(n_case)
    n_case = 0
    for scale, colors in ((1, None), (0.2, None), (0.05, ("#ff8000", "#0080ff", "#80ff00", "#8000ff"))):
        arr = rng.normal(0, 4, (19, 23, 2)).astype(np.float32)
        d[f"r2_{n_case}_in"] = arr
        d[f"r2_{n_case}_scale"] = np.float64(scale)
        d[f"r2_{n_case}_colors"] = np.array(colors if colors else ("#ffff00", "#0000ff", "#ff00ff", "#00ff00"))
        d[f"r2_{n_case}_out"] = render2d(arr, scale=scale, colors=colors)
        n_case += 1
    d["r2_cases"] = np.int32(n_case)
    np.savez_compressed(os.path.join(OUT, "flow_ops.npz"), **d)
    print("flow ops:", int(d["merge_cases"]), "merges,", int(d["conv_cases"]), "convolutions")


POLAR_CASES = [
    ("r", "a"), ("r*2", "a+t"), ("numpy.sqrt(r)+1", "-a"), ("numpy.where(r>2, r, 0)", "a"), ("r", "t"), ("3", "a"),
    ("r**2/10", "numpy.float64(0.5)*a"), ("numpy.minimum(r, 3)", "a + math.pi/4"),
    ("numpy.clip(r,1,4)*numpy.cos(a)", "numpy.arctan2(numpy.sin(a), numpy.cos(a)*2)"), ("r % 1.5", "a // 0.5 * 0.5"),
    ("abs(r-2)**0.5", "numpy.maximum(a, -1) * (t + 1)"), ("2*t", "math.pi*t"),
]


def capture_polar():
    """The polar flow filter (filters.py:75-87) applied by the reference; t = 0.7."""
    from transflow.flow.filters import FlowFilter as RefFilter
    rng = np.random.default_rng(505)
    d = {"cases": np.int32(len(POLAR_CASES)), "t": np.float64(0.7)}
    for i, (er, ea) in enumerate(POLAR_CASES):
        flow = rng.normal(0, 2.5, (21, 34, 2)).astype(np.float32)
        flow[0, 0] = 0
        d[f"in_{i}"] = flow.copy()
        RefFilter.from_args("polar", (er, ea)).apply(flow, 0.7)
        d[f"out_{i}"] = flow
        d[f"er_{i}"], d[f"ea_{i}"] = np.array(er), np.array(ea)
    # through post_process, between two other filters
    fs = make_fs(FlowSource.Direction.BACKWARD, 21, 34)
    fs.flow_filters = [RefFilter.from_args("scale", ("1.5",)), RefFilter.from_args("polar", ("r+1", "a*2")),
                       RefFilter.from_args("clip", ("4",))]
    fs.output_frame_index = 21          # t = 21 / 30
    raw = rng.normal(0, 2.5, (21, 34, 2)).astype(np.float32)
    d["chain_in"] = raw.copy()
    d["chain_out"] = np.asarray(fs.post_process(raw))
    np.savez_compressed(os.path.join(OUT, "flow_polar.npz"), **d)
    print("polar cases:", len(POLAR_CASES))


def capture_known_answers():
    """tests/test_compositor.py:20-54 re-run, outputs stored."""
    d = {}
    d["basic_render"] = Compositor(1, 1, [], background_color="#ff8000").render()
    flow = np.array([[[0, 1], [0, 1], [0, 0]], [[0, 0], [0, 0], [0, 0]]]).astype(np.float32)
    d["flow"] = flow
    layer = MoveReferenceLayer(LayerConfig(0), 2, 3, [])
    layer.update(flow)
    d["moveref_data"] = layer.data.copy()
    layer = MoveReferenceLayer(LayerConfig(0, reset_mode="random", reset_random_factor=1), 2, 3, [])
    layer.update(flow)
    d["moveref_reset_data"] = layer.data.copy()
    layer = MoveReferenceLayer(LayerConfig(0, reset_mode="random", reset_random_factor=1,
                                           reset_mask="border-left:1"), 2, 3, [])
    layer.update(flow)
    d["moveref_reset_mask_data"] = layer.data.copy()
    d["moveref_reset_mask"] = layer.reset_mask.copy()
    np.savez_compressed(os.path.join(OUT, "remap_known_answers.npz"), **d)


def capture_multilayer():
    """Two moveref layers over one background (compositor.py:31-40 order)."""
    rng = np.random.default_rng(303)
    h, w = 24, 32
    l0 = MoveReferenceLayer(LayerConfig(0), h, w, [])
    l1 = MoveReferenceLayer(LayerConfig(1, moving_pixels_leave_empty_spot=True), h, w, [])
    l1.mask_alpha = (rng.random((h, w)) < 0.5).astype(np.float32)
    pm0 = rng.integers(0, 256, (3, h, w, 3), dtype=np.uint8)
    pm1 = rng.integers(0, 256, (3, h, w, 3), dtype=np.uint8)
    l0.set_sources([FakeSource(list(pm0), np.ones((h, w), bool))])
    l1.set_sources([FakeSource(list(pm1), np.ones((h, w), bool))])
    comp = Compositor(h, w, [l0, l1], background_color="#123456")
    fs = make_fs(FlowSource.Direction.BACKWARD, h, w)
    d = {"h": np.int32(h), "w": np.int32(w), "pixmap_l0": pm0, "pixmap_l1": pm1,
         "mask_alpha_l1": l1.mask_alpha.copy(),
         "background": np.array(comp.background_color, dtype=np.uint8)}
    for t in range(3):
        flow = np.asarray(fs.post_process(rng.normal(0, 3, (h, w, 2)).astype(np.float32)), np.float32)
        comp.update(flow)
        d[f"flow_{t}"] = flow
        d[f"frame_{t}"] = comp.render().copy()
        d[f"data_l0_{t}"] = l0.data.copy()
        d[f"data_l1_{t}"] = l1.data.copy()
    np.savez_compressed(os.path.join(OUT, "remap_multilayer.npz"), **d)


MASK_SPECS = ["zeros", "ones", "border:2", "border:3:5", "border:1:2:3:4", "border:10%", "border-left:1",
              "border-top:25%", "border-right:4", "border-bottom:2", "hline:5", "vline:30%", "circle:40%", "circle:7",
              "rect:60%", "rect:20:10", "rect:50%:inv", "grid:2:3:4", "border:2:inv", "circle:40%:inv"]
COLORS = ["#ff8000", "white", "black", "cornflowerblue", "Tomato", "rgb(1, 2, 3)", "(10,20,30)", "0x123456", "abcdef",
          "rebeccapurple", "grey", "gray", "lime"]


def capture_masks():
    """utils.load_float_mask / load_bool_mask / parse_color outputs for the argument grammar."""
    from transflow.utils import load_bool_mask, load_float_mask, parse_color
    d = {"specs": np.array(MASK_SPECS), "colors": np.array(COLORS)}
    for shape in [(37, 53), (60, 80)]:
        for i, spec in enumerate(MASK_SPECS):
            d[f"f_{shape[0]}_{i}"] = np.asarray(load_float_mask(spec, shape, 1))
            d[f"b_{shape[0]}_{i}"] = np.asarray(load_bool_mask(spec, shape, True))
    d["none_float"] = load_float_mask(None, (4, 5), 1)
    d["none_bool"] = load_bool_mask(None, (4, 5), True)
    d["color_values"] = np.array([parse_color(c) for c in COLORS], dtype=np.int32)
    np.savez_compressed(os.path.join(OUT, "masks.npz"), **d)



def capture_flow_presteps():
    """FlowSource.post_process with flow filters and a flow mask (source.py:339-343,
    filters.py:36-72) -- the values the filters' lambdas return keep their Python/numpy type."""
    from transflow.flow.filters import FlowFilter
    rng = np.random.default_rng(404)
    h, w = 37, 53
    cases = [
        ("scale=2.5*t", 0.4), ("scale=numpy.float64(0.3)+t", 0.2), ("threshold=1.5+t", 0.5),
        ("threshold=numpy.float64(2.0)", 0.0), ("clip=2.0", 0.0), ("clip=numpy.float64(1.0)+t", 1.0),
        ("scale=-1;threshold=0.8;clip=3", 0.0), ("clip=0", 0.0),
    ]
    d = {"count": np.int32(len(cases)), "specs": np.array([c[0] for c in cases]),
         "ts": np.array([c[1] for c in cases])}
    mask = rng.random((h, w, 1)).astype(np.float32)
    d["mask"] = mask
    for i, (spec, t) in enumerate(cases):
        for direction in (FlowSource.Direction.FORWARD, FlowSource.Direction.BACKWARD):
            for use_mask in (False, True):
                fs = make_fs(direction, h, w)
                fs.flow_filters = [FlowFilter.from_args(p[:p.index("=")].strip(), tuple(p[p.index("=") + 1:].strip().split(":")))
                                   for p in spec.split(";")]
                fs.mask = mask if use_mask else None
                fs.output_frame_index = int(round(t * fs.framerate))
                raw = rng.normal(0, 2.0, (h, w, 2)).astype(np.float32)
                raw[0, 0] = 0   # a zero vector: norm 0 (0/0 in clip=0)
                work = raw.copy()
                with np.errstate(all="ignore"):
                    out = fs.post_process(work)
                key = f"{i}_{direction.value}_{int(use_mask)}"
                d[f"in_{key}"] = raw
                d[f"out_{key}"] = np.asarray(out)
                d[f"raw_after_{key}"] = work      # filters act in place on the raw flow
                d[f"t_{key}"] = np.float64(fs.t)
    np.savez_compressed(os.path.join(OUT, "flow_presteps.npz"), **d)
    print("flow pre-step cases:", len(cases) * 4)


if __name__ == "__main__":
    if "--polar-only" in sys.argv:
        capture_polar()
        sys.exit(0)
    if "--flowops-only" in sys.argv:
        capture_flow_ops()
        sys.exit(0)
    if "--layers2-only" in sys.argv:
        capture_other_layers()
        sys.exit(0)
    if "--presteps-only" in sys.argv:
        capture_flow_presteps()
        sys.exit(0)
    capture_flow_presteps()
    capture_masks()
    if "--masks-only" in sys.argv:
        sys.exit(0)
    capture_post_process()
    capture_layers()
    capture_known_answers()
    capture_multilayer()
    capture_other_layers()
    capture_flow_ops()
    capture_polar()
    total = sum(os.path.getsize(os.path.join(OUT, f)) for f in os.listdir(OUT))
    print("golden bytes:", total)
