"""The numpy oracle against the reference ITSELF, run live on random configurations (build container
only: skipped where /root/reference does not exist).  The captured vectors under tests/golden pin
18 + 18 hand-picked cases; this walks the configuration space -- every move flag, reset mode and
introduce_* flag, masks, one or two sources of 3 or 4 channels, shapes down to one pixel -- so the
oracle the GPU tests compare with is the reference's behaviour, not a reading of it.  CPU only."""
import os
import sys

import numpy as np
import pytest

from oracle import remap_ref as R
from tests.helpers import INTRO_KEYS, PRM_KEYS, capture_frame_numbers

REF = "/root/reference"
pytestmark = pytest.mark.skipif(not os.path.isdir(os.path.join(REF, "transflow")),
                                reason="needs the reference tree (build container only)")


class FakeSource:
    """PixmapSourceInterface stand-in (pixmap_source_interface.py:12-37), as tools/capture_golden.py's."""

    def __init__(self, frames, introduction_mask):
        self.frames, self.introduction_mask, self.counter = list(frames), introduction_mask, -1

    def next(self, timeout=1):
        self.counter += 1
        return self.frames[self.counter % len(self.frames)]

    @property
    def frame_number(self):
        return self.counter


@pytest.fixture(scope="module")
def ref():
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from transflow.compositor import Compositor
    from transflow.compositor.layers.layer import Layer
    from transflow.config import LayerConfig
    from transflow.flow.sources.source import FlowSource
    return Compositor, Layer, LayerConfig, FlowSource


def _random_case(rng, cls):
    h, w = int(rng.integers(1, 26)), int(rng.integers(1, 34))
    cfg = dict(transparent_pixels_can_move=bool(rng.integers(2)), pixels_can_move_to_empty_spot=bool(rng.integers(2)),
               pixels_can_move_to_filled_spot=bool(rng.integers(2)), moving_pixels_leave_empty_spot=bool(rng.integers(2)))
    if cls == "introduction":
        cfg.update({k: bool(rng.integers(2)) for k in INTRO_KEYS})
    else:
        cfg.update(reset_mode=str(rng.choice(["off", "random", "constant", "linear"])),
                   reset_random_factor=float(rng.choice([0.0, 0.3, 1.0])),
                   reset_constant_step=float(rng.choice([0.5, 1.0, 2.5])),
                   reset_linear_factor=float(rng.choice([0.1, 0.5])), reset_source=bool(rng.integers(2)))
    masks = dict(mask_alpha=rng.choice([0.0, 0.5, 1.0], (h, w)).astype(np.float32))
    if cls != "sum":
        masks.update(mask_src=rng.random((h, w)) < 0.85, mask_dst=rng.random((h, w)) < 0.85)
    if cls != "introduction":
        masks.update(reset_mask=rng.random((h, w)).astype(np.float32))
    ns = int(rng.integers(1, 3))
    intro = [rng.random((h, w)) < 0.6 for _ in range(ns)]
    chans = [int(rng.choice([3, 4])) for _ in range(ns)]
    return h, w, cfg, masks, intro, chans


@pytest.mark.parametrize("cls", ["moveref", "sum", "introduction"])
def test_oracle_equals_live_reference_on_random_configurations(ref, cls):
    Compositor, Layer, LayerConfig, FlowSource = ref
    rng = np.random.default_rng({"moveref": 11, "sum": 12, "introduction": 13}[cls])
    orig = np.random.random
    for trial in range(60):
        h, w, cfg, masks, intro, chans = _random_case(rng, cls)
        nframes = 4
        pixmaps = [[rng.integers(0, 256, (h, w, c), dtype=np.uint8) for _ in range(nframes)] for c in chans]
        bg = "#%02x%02x%02x" % tuple(int(v) for v in rng.integers(0, 256, 3))
        layer = Layer.from_args(LayerConfig(0, classname=cls, **cfg), h, w, [])
        for k, v in masks.items():
            setattr(layer, k, v.copy())
        layer.set_sources([FakeSource(pixmaps[s], intro[s]) for s in range(len(intro))])
        comp = Compositor(h, w, [layer], background_color=bg)
        fs = FlowSource(FlowSource.Direction.BACKWARD, w, h, 30.0, None, 0, 0, 0)
        if cls == "introduction":
            prm = R.IntroParams(**{k: v for k, v in cfg.items() if k in PRM_KEYS + INTRO_KEYS})
            ora = R.IntroductionLayer(h, w, prm, introduction_masks=intro, **masks)
        else:
            prm = R.LayerParams(**{k: v for k, v in cfg.items() if k in PRM_KEYS})
            ora = (R.SumLayer if cls == "sum" else R.MoveRefLayer)(h, w, prm, introduction_masks=intro, **masks)
        np.testing.assert_array_equal(layer.data, ora.data)
        for t in range(nframes):
            raw = rng.normal(0, 2.5, (h, w, 2)).astype(np.float32)
            flow = np.asarray(fs.post_process(raw.copy()), dtype=np.float32)
            np.testing.assert_array_equal(flow, R.post_process(raw.copy(), R.BACKWARD))
            u = rng.random((h, w))
            np.random.random = lambda size=None, _u=u: _u.copy()
            try:
                comp.update(flow)
            finally:
                np.random.random = orig
            pms = [pixmaps[s][t] for s in range(len(intro))]
            if cls == "introduction":
                ora.update(flow, pms, frame_numbers=capture_frame_numbers(prm, t, len(intro)))
            else:
                ora.update(flow, pms, u=u)
            msg = f"{cls} trial {trial} {h}x{w} {cfg} frame {t}"
            np.testing.assert_array_equal(layer.data, ora.data, err_msg=msg)
            np.testing.assert_array_equal(np.asarray(layer.rgba), np.asarray(ora.rgba), err_msg=msg)
            frame = comp.render()
            exp = R.composite(np.broadcast_to(np.uint8(comp.background_color), (h, w, 3)), [ora.render()])
            np.testing.assert_array_equal(frame, exp, err_msg=msg)
            np.testing.assert_array_equal(layer.data, ora.data, err_msg="after render " + msg)


def test_flow_ops_oracle_equals_live_reference_on_random_inputs(ref):
    """Flow merging, upscale, post_process with a convolution kernel (both directions, random kernel
    shapes and dtypes), render1d / render2d: oracle/flow_ops_ref.py against the reference's functions."""
    import typing
    import typing_extensions
    if not hasattr(typing, "Self"):
        typing.Self = typing_extensions.Self          # transflow.pipeline wants Python 3.11's
    from oracle import flow_ops_ref as F
    from transflow.output.render import render1d, render2d
    from transflow.pipeline import Pipeline
    from transflow.utils import upscale_array
    _, _, _, FlowSource = ref
    rng = np.random.default_rng(2718)
    for trial in range(40):
        h, w = int(rng.integers(1, 30)), int(rng.integers(1, 40))
        n = int(rng.integers(1, 5))
        flows = [rng.normal(0, 1.5, (h, w, 2)).astype(np.float32) for _ in range(n)]
        for f in flows[1:]:
            f[rng.random((h, w, 2)) < 0.3] = 0
        for kind, fn in Pipeline.FLOW_MERGING_FUNCTIONS.items():
            if kind == "absmax" and n != 2:
                continue
            np.testing.assert_array_equal(F.merge(kind, [f.copy() for f in flows]),
                                          np.asarray(fn([f.copy() for f in flows])), err_msg=f"{kind} n={n} {h}x{w}")
        wf, hf = int(rng.integers(1, 4)), int(rng.integers(1, 4))
        np.testing.assert_array_equal(F.upscale(flows[0], wf, hf), upscale_array(flows[0], wf, hf))
        kh, kw_ = int(rng.integers(1, 7)), int(rng.integers(1, 7))
        kernel = rng.normal(0, 0.3, (kh, kw_))
        kernel = [kernel, kernel.astype(np.float32), np.rint(kernel * 4).astype(np.int64)][int(rng.integers(3))]
        for direction, d in ((FlowSource.Direction.FORWARD, 0), (FlowSource.Direction.BACKWARD, 1)):
            fs = FlowSource(direction, w, h, 30.0, None, 0, 0, 0)
            fs.kernel = kernel
            raw = rng.normal(0, 3.0, (h, w, 2)).astype(np.float32)
            got = F.post_process_with_kernel(raw.copy(), kernel, d)
            exp = np.asarray(fs.post_process(raw.copy()))
            assert got.dtype == exp.dtype, f"kernel {kernel.dtype} {kernel.shape}"
            np.testing.assert_array_equal(got, exp, err_msg=f"kernel {kernel.dtype} {kernel.shape} dir {d} {h}x{w}")
        arr = np.abs(rng.normal(0, 2, (h, w))).astype(np.float32)
        scale, binary = float(rng.choice([0.25, 0.5, 1.0, 2.0])), bool(rng.integers(2))
        np.testing.assert_array_equal(F.render1d(arr, scale=scale, binary=binary), render1d(arr, scale=scale, binary=binary))
        np.testing.assert_array_equal(F.render2d(flows[0], scale=scale), render2d(flows[0], scale=scale))


def test_mask_rules_against_the_reference_live():
    """transflow_amd/masks.py is a rule table with builders of its own (row x column outer products,
    a distance test), not a restatement of utils.py:51-140: walk the argument grammar at random --
    every rule name, 0-5 arguments from pixels / percentages / blanks / junk, dimensions larger than the
    frame (where the reference's results are what numpy slicing makes of negative bounds), ':inv' --
    and require the same array and dtype wherever the reference produces one, an exception wherever it
    raises one."""
    import random
    sys.dont_write_bytecode = True
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from transflow import utils as U
    from transflow_amd import masks as M
    rnd = random.Random(5)
    names = ["border", "border-top", "border-right", "border-bottom", "border-left", "hline", "vline", "circle",
             "rect", "grid", "zeros", "ones", "Border", "CIRCLE", "Rect"]
    tokens = ["", "0", "1", "2", "3", "4", "7", "15", "200", "5%", "50%", "120%", "x"]
    produced = 0
    for _ in range(1500):
        spec = ":".join([rnd.choice(names)] + [rnd.choice(tokens) for _ in range(rnd.choice([0, 1, 1, 2, 2, 3, 4, 5]))])
        spec += rnd.choice(["", ":inv"])
        shape = rnd.choice([(37, 53), (60, 80), (8, 8), (5, 9), (1, 1)])
        try:
            want = U.load_float_mask(spec, shape, 1)
        except Exception:
            with pytest.raises(Exception):
                M.load_float_mask(spec, shape, 1)
            continue
        got = M.load_float_mask(spec, shape, 1)
        assert got.dtype == want.dtype, spec
        np.testing.assert_array_equal(got, want, err_msg=f"{spec} {shape}")
        np.testing.assert_array_equal(M.load_bool_mask(spec, shape, True), U.load_bool_mask(spec, shape, True))
        produced += 1
    assert produced > 400
