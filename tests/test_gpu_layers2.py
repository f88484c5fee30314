"""GPU parity of the sum / static / introduction layer classes: the reference's own
vectors (tests/golden/layer2_*.npz) through HipCompositor, bit for bit."""
import os
import pickle

import numpy as np
import pytest

from tests.helpers import case_cfg, layer2_case_files

pytestmark = pytest.mark.gpu


class FakeSource:
    """PixmapSourceInterface stand-in (pixmap_source_interface.py:12-37), as the capture script's."""

    def __init__(self, frames, introduction_mask):
        self.frames, self.introduction_mask, self.counter = list(frames), introduction_mask, -1

    def next(self, timeout=1):
        self.counter += 1
        return self.frames[self.counter % len(self.frames)]

    @property
    def frame_number(self):
        return self.counter


def _compositor(z):
    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import LayerConfig
    h, w = int(z["h"]), int(z["w"])
    comp = HipCompositor.from_args(h, w, [LayerConfig(0, classname=str(z["classname"]), **case_cfg(z))],
                                   background_color="#%02x%02x%02x" % tuple(int(v) for v in z["background"]))
    layer = comp.layers[0]
    # the capture script installed arbitrary mask arrays on the reference layer: do the same here
    for key in ("mask_src", "mask_dst", "mask_alpha", "reset_mask"):
        if key in z.files:
            setattr(layer, key, z[key])
    ns = int(z["nsources"])
    comp.set_sources({0: [FakeSource(z[f"pixmap_{s}"], z[f"intro_{s}"]) for s in range(ns)]})
    return comp, layer


@pytest.mark.parametrize("path", layer2_case_files(), ids=lambda p: os.path.basename(p)[7:-4])
def test_layer2_matches_reference_vectors(path):
    z = np.load(path)
    comp, layer = _compositor(z)
    has_data = "data_init" in z.files
    if has_data:
        np.testing.assert_array_equal(layer.data, z["data_init"])
        assert layer.data.shape[2] == layer.DEPTH
    np.testing.assert_array_equal(layer.rgba, z["rgba_init"])
    orig = np.random.random
    for t in range(int(z["nframes"])):
        np.random.random = lambda size=None, _u=z[f"u_{t}"]: _u.copy()
        try:
            comp.update(z[f"flow_{t}"])
        finally:
            np.random.random = orig
        if has_data:
            np.testing.assert_array_equal(layer.data, z[f"data_{t}"], err_msg=f"data t={t}")
        np.testing.assert_array_equal(layer.rgba, z[f"rgba_{t}"], err_msg=f"rgba t={t}")
        frame = comp.render()
        np.testing.assert_array_equal(frame, z[f"frame_{t}"], err_msg=f"frame t={t}")
        np.testing.assert_array_equal(layer.rgba, z[f"rgba_after_render_{t}"], err_msg=f"rgba after render t={t}")
        if has_data:
            np.testing.assert_array_equal(layer.data, z[f"data_after_render_{t}"])
    comp.close()


@pytest.mark.parametrize("name", ["intro_once_leave_two_rgba", "sum_reset_random", "static_rgba_two"])
def test_layer2_checkpoint_roundtrip(name):
    """Pickle after two frames, continue on the copy: same frames as the uninterrupted run
    (pipeline.py:225-242, 290-306: sources are stripped before pickling and re-installed after)."""
    z = np.load([p for p in layer2_case_files() if name in p][0])
    comp, layer = _compositor(z)
    orig = np.random.random
    try:
        for t in range(int(z["nframes"])):
            if t == 2:
                sources = layer.sources
                blob = pickle.dumps(comp)
                comp.close()
                comp = pickle.loads(blob)
                layer = comp.layers[0]
                comp.set_sources({0: sources})
            np.random.random = lambda size=None, _u=z[f"u_{t}"]: _u.copy()
            comp.update(z[f"flow_{t}"])
            np.testing.assert_array_equal(comp.render(), z[f"frame_{t}"], err_msg=f"frame t={t}")
    finally:
        np.random.random = orig
    comp.close()


def test_layer2_argument_errors():
    from transflow_amd._lib import TfError
    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import LayerConfig
    from transflow_amd.remap import RemapLayer
    with pytest.raises(ValueError):
        HipCompositor.from_args(4, 4, [LayerConfig(0, classname="nope")])         # layer.py:56
    intro = RemapLayer(4, 5, layer_class="introduction")
    pm = np.zeros((4, 5, 3), np.uint8)
    with pytest.raises((TfError, ValueError, RuntimeError)):
        intro.gather(0, pm)            # introduction layers take pixmaps through introduce()
    with pytest.raises((TfError, ValueError, RuntimeError)):
        intro.introduce(0, pm, 0)      # no sources / no update yet
    mv = RemapLayer(4, 5)
    with pytest.raises((TfError, ValueError, RuntimeError)):
        mv.introduce(0, pm, 0)


def test_random_introduction_and_sum_configurations_vs_oracle():
    """Beyond the captured cases: random flag combinations of the introduction layer (move flags x
    introduce_* flags, masks, 1-2 sources of 3 or 4 channels) and of the sum layer (every reset mode),
    small odd shapes, four frames each, through HipCompositor; bit for bit against the oracle."""
    from oracle import remap_ref as R
    from tests.helpers import INTRO_KEYS, PRM_KEYS, capture_frame_numbers
    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import LayerConfig
    rng = np.random.default_rng(999)
    orig = np.random.random
    for trial in range(30):
        h, w = int(rng.integers(1, 30)), int(rng.integers(1, 45))
        cls = "introduction" if trial % 3 else "sum"
        cfg = dict(transparent_pixels_can_move=bool(rng.integers(2)), pixels_can_move_to_empty_spot=bool(rng.integers(2)),
                   pixels_can_move_to_filled_spot=bool(rng.integers(2)), moving_pixels_leave_empty_spot=bool(rng.integers(2)))
        if cls == "introduction":
            cfg.update({k: bool(rng.integers(2)) for k in INTRO_KEYS})
        else:
            cfg.update(reset_mode=str(rng.choice(["off", "random", "constant", "linear"])),
                       reset_random_factor=float(rng.choice([0.2, 1.0])), reset_source=bool(rng.integers(2)))
        masks = dict(mask_alpha=rng.choice([0.0, 0.5, 1.0], (h, w)).astype(np.float32))
        if cls == "introduction":
            masks.update(mask_src=rng.random((h, w)) < 0.85, mask_dst=rng.random((h, w)) < 0.85)
        else:
            masks.update(reset_mask=rng.random((h, w)).astype(np.float32))
        ns = int(rng.integers(1, 3))
        intro = [rng.random((h, w)) < 0.6 for _ in range(ns)]
        chans = [int(rng.choice([3, 4])) for _ in range(ns)]
        nframes = 4
        pixmaps = [[rng.integers(0, 256, (h, w, c), dtype=np.uint8) for _ in range(nframes)] for c in chans]
        bg = tuple(int(v) for v in rng.integers(0, 256, 3))
        comp = HipCompositor.from_args(h, w, [LayerConfig(0, classname=cls, **cfg)],
                                       background_color="#%02x%02x%02x" % bg)
        layer = comp.layers[0]
        for key, val in masks.items():
            setattr(layer, key, val)
        comp.set_sources({0: [FakeSource(pixmaps[s], intro[s]) for s in range(ns)]})
        if cls == "introduction":
            prm = R.IntroParams(**{k: v for k, v in cfg.items() if k in PRM_KEYS + INTRO_KEYS})
            ora = R.IntroductionLayer(h, w, prm, introduction_masks=intro, **masks)
        else:
            ora = R.SumLayer(h, w, R.LayerParams(**{k: v for k, v in cfg.items() if k in PRM_KEYS}),
                             introduction_masks=intro, **masks)
        for t in range(nframes):
            flow = rng.normal(0, 2.5, (h, w, 2)).astype(np.float32)
            if cls == "introduction":
                flow = R.post_process(flow, R.BACKWARD)            # a moving layer needs in-frame targets
            u = rng.random((h, w))
            np.random.random = lambda size=None, _u=u: _u.copy()
            try:
                comp.update(flow)
            finally:
                np.random.random = orig
            pms = [pixmaps[s][t % nframes] for s in range(ns)]
            if cls == "introduction":
                # a source is read only in frames in which the layer introduces (introduction.py:21-22)
                once_done = prm.introduce_once and t > 0
                if not once_done:
                    ora.update(flow, [pixmaps[s][t] for s in range(ns)], frame_numbers=capture_frame_numbers(prm, t, ns))
                else:
                    ora.update(flow, pms, frame_numbers=capture_frame_numbers(prm, t, ns))
            else:
                ora.update(flow, pms, u=u)
            msg = f"trial {trial} {cls} {h}x{w} {cfg} frame {t}"
            np.testing.assert_array_equal(layer.data, ora.data, err_msg=msg)
            frame = comp.render()
            exp = R.composite(np.broadcast_to(np.uint8(bg), (h, w, 3)), [ora.render()])
            np.testing.assert_array_equal(frame, exp, err_msg=msg)
            np.testing.assert_array_equal(layer.data, ora.data, err_msg="after render " + msg)
        comp.close()


def test_random_layer_stacks_vs_oracle():
    """Two to four layers of random classes in one compositor (compositor.py:27-40: every layer updated with
    the same flow, painted in order over the background), random flags each; three frames; bit for bit."""
    from oracle import remap_ref as R
    from tests.helpers import INTRO_KEYS, PRM_KEYS, capture_frame_numbers
    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import LayerConfig
    rng = np.random.default_rng(1212)
    orig = np.random.random
    for trial in range(12):
        h, w = int(rng.integers(2, 28)), int(rng.integers(2, 40))
        classes = [str(rng.choice(["moveref", "introduction", "static", "sum"])) for _ in range(int(rng.integers(2, 5)))]
        cfgs, oras, srcs = [], [], {}
        nframes = 3
        for li, cls in enumerate(classes):
            cfg = dict(transparent_pixels_can_move=bool(rng.integers(2)), moving_pixels_leave_empty_spot=bool(rng.integers(2)))
            if cls == "introduction":
                cfg.update({k: bool(rng.integers(2)) for k in INTRO_KEYS})
            elif cls in ("moveref", "sum"):
                cfg.update(reset_mode=str(rng.choice(["off", "random", "linear"])), reset_random_factor=0.4)
            intro = [rng.random((h, w)) < 0.6]
            pms = [rng.integers(0, 256, (h, w, int(rng.choice([3, 4]))), dtype=np.uint8) for _ in range(nframes)]
            alpha = rng.choice([0.0, 1.0, 1.0], (h, w)).astype(np.float32)
            cfgs.append(LayerConfig(li, classname=cls, **cfg))
            srcs[li] = (pms, intro, alpha)
            if cls == "introduction":
                prm = R.IntroParams(**{k: v for k, v in cfg.items() if k in PRM_KEYS + INTRO_KEYS})
                oras.append(R.IntroductionLayer(h, w, prm, introduction_masks=intro, mask_alpha=alpha))
            elif cls == "static":
                oras.append(R.StaticLayer(h, w, mask_alpha=alpha, introduction_masks=intro))
            else:
                prm = R.LayerParams(**{k: v for k, v in cfg.items() if k in PRM_KEYS})
                oras.append((R.SumLayer if cls == "sum" else R.MoveRefLayer)(h, w, prm, introduction_masks=intro,
                                                                              mask_alpha=alpha))
        bg = tuple(int(v) for v in rng.integers(0, 256, 3))
        comp = HipCompositor.from_args(h, w, cfgs, background_color="#%02x%02x%02x" % bg)
        for li, layer in enumerate(comp.layers):
            layer.mask_alpha = srcs[li][2]
        comp.set_sources({li: [FakeSource(srcs[li][0], srcs[li][1][0])] for li in range(len(classes))})
        for t in range(nframes):
            flow = R.post_process(rng.normal(0, 2.0, (h, w, 2)).astype(np.float32), R.BACKWARD)
            u = rng.random((h, w))
            np.random.random = lambda size=None, _u=u: _u.copy()
            try:
                comp.update(flow)
            finally:
                np.random.random = orig
            imgs = []
            for li, (cls, ora) in enumerate(zip(classes, oras)):
                pm = [srcs[li][0][t]]
                if cls == "introduction":
                    ora.update(flow, pm, frame_numbers=capture_frame_numbers(ora.prm, t, 1))
                else:
                    ora.update(flow, pm, u=u)
            frame = comp.render()
            for ora in oras:
                imgs.append(ora.render())
            exp = R.composite(np.broadcast_to(np.uint8(bg), (h, w, 3)), imgs)
            np.testing.assert_array_equal(frame, exp, err_msg=f"trial {trial} {classes} {h}x{w} frame {t}")
        comp.close()
