"""The numpy remap oracle against vectors produced by the reference itself
(tools/capture_golden.py) and the reference's own known answers
(reference tests/test_compositor.py:20-54).  CPU only."""
import os

import numpy as np
import pytest

from oracle import remap_ref as R
from tests.helpers import GOLDEN, case_cfg, layer_case_files, oracle_params


def test_post_process_golden():
    z = np.load(os.path.join(GOLDEN, "remap_post_process.npz"))
    n = int(z["count"])
    assert n >= 20
    for i in range(n):
        out = R.post_process(z[f"in_{i}"].copy(), int(z[f"dir_{i}"]))
        np.testing.assert_array_equal(out, z[f"out_{i}"], err_msg=f"case {i}")


def test_post_process_empty_and_single():
    for shape in [(0, 0, 2), (1, 1, 2), (1, 7, 2), (9, 1, 2)]:
        f = np.full(shape, 3.7, np.float32)
        for d in (R.FORWARD, R.BACKWARD):
            out = R.post_process(f.copy(), d)
            assert out.shape == shape
            if shape[0] * shape[1] == 1:
                assert (out == 0).all()


@pytest.mark.parametrize("path", layer_case_files(), ids=lambda p: os.path.basename(p)[12:-4])
def test_layer_sequences_golden(path):
    z = np.load(path)
    h, w = int(z["h"]), int(z["w"])
    cfg = case_cfg(z)
    ns = int(z["nsources"])
    intro = [z[f"intro_{s}"] for s in range(ns)]
    layer = R.MoveRefLayer(h, w, oracle_params(cfg), z["mask_src"], z["mask_dst"],
                           z["mask_alpha"], z["reset_mask"], intro)
    np.testing.assert_array_equal(layer.data, z["data_init"])
    for t in range(int(z["nframes"])):
        layer.update(z[f"flow_{t}"], [z[f"pixmap_{s}"][t] for s in range(ns)], z[f"u_{t}"])
        np.testing.assert_array_equal(layer.data, z[f"data_{t}"], err_msg=f"data t={t}")
        np.testing.assert_array_equal(layer.rgba, z[f"rgba_{t}"], err_msg=f"rgba t={t}")
        frame = R.composite(np.broadcast_to(z["background"], (h, w, 3)), [layer.render()])
        np.testing.assert_array_equal(frame, z[f"frame_{t}"], err_msg=f"frame t={t}")
        np.testing.assert_array_equal(layer.rgba, z[f"rgba_after_render_{t}"])


def test_reference_known_answers():
    """reference tests/test_compositor.py:20-54, re-stated."""
    z = np.load(os.path.join(GOLDEN, "remap_known_answers.npz"))
    assert tuple(z["basic_render"][0, 0]) == (255, 128, 0)
    flow = np.array([[[0, 1], [0, 1], [0, 0]], [[0, 0], [0, 0], [0, 0]]], np.float32)
    layer = R.MoveRefLayer(2, 3)
    layer.update(flow)
    assert tuple(layer.data[0, 0, :2]) == (1, 0) and tuple(layer.data[0, 1, :2]) == (1, 1)
    np.testing.assert_array_equal(layer.data, z["moveref_data"])
    # reset_random_factor=1 with a ones mask resets every pixel whatever u is (u<1)
    layer = R.MoveRefLayer(2, 3, R.LayerParams(reset_mode="random", reset_random_factor=1))
    layer.update(flow, u=np.random.default_rng(0).random((2, 3)))
    assert tuple(layer.data[0, 0, :2]) == (0, 0) and tuple(layer.data[0, 1, :2]) == (0, 1)
    np.testing.assert_array_equal(layer.data, z["moveref_reset_data"])
    layer = R.MoveRefLayer(2, 3, R.LayerParams(reset_mode="random", reset_random_factor=1),
                           reset_mask=z["moveref_reset_mask"])
    layer.update(flow, u=np.random.default_rng(1).random((2, 3)))
    assert tuple(layer.data[0, 0, :2]) == (0, 0) and tuple(layer.data[0, 1, :2]) == (1, 1)
    np.testing.assert_array_equal(layer.data, z["moveref_reset_mask_data"])


def test_multilayer_golden():
    z = np.load(os.path.join(GOLDEN, "remap_multilayer.npz"))
    h, w = int(z["h"]), int(z["w"])
    l0 = R.MoveRefLayer(h, w, introduction_masks=[np.ones((h, w), bool)])
    l1 = R.MoveRefLayer(h, w, R.LayerParams(moving_pixels_leave_empty_spot=True),
                        mask_alpha=z["mask_alpha_l1"], introduction_masks=[np.ones((h, w), bool)])
    for t in range(3):
        l0.update(z[f"flow_{t}"], [z["pixmap_l0"][t]])
        l1.update(z[f"flow_{t}"], [z["pixmap_l1"][t]])
        np.testing.assert_array_equal(l0.data, z[f"data_l0_{t}"])
        np.testing.assert_array_equal(l1.data, z[f"data_l1_{t}"])
        frame = R.composite(np.broadcast_to(z["background"], (h, w, 3)), [l0.render(), l1.render()])
        np.testing.assert_array_equal(frame, z[f"frame_{t}"])


def test_move_rejects_out_of_frame():
    layer = R.MoveRefLayer(4, 4)
    flow = np.zeros((4, 4, 2), np.float32)
    flow[3, 3] = (2, 0)
    with pytest.raises(IndexError):
        layer.update(flow)


def _prestep_cases():
    z = np.load(os.path.join(GOLDEN, "flow_presteps.npz"))
    for i in range(int(z["count"])):
        for direction in (0, 1):
            for use_mask in (0, 1):
                yield z, i, direction, use_mask, f"{i}_{direction}_{use_mask}"


def test_flow_presteps_golden():
    """Flow filters + flow mask + post_process (reference source.py:337-363, filters.py:36-72)."""
    from transflow_amd.flow import FlowFilter
    for z, i, direction, use_mask, key in _prestep_cases():
        filters = [FlowFilter.from_string(p) for p in str(z["specs"][i]).split(";")]
        t = float(z[f"t_{key}"])
        raw = z[f"in_{key}"].copy()
        with np.errstate(all="ignore"):
            flow = R.pre_steps(raw, [(f.name, f.expr(t)) for f in filters], z["mask"] if use_mask else None)
            out = R.post_process(flow, direction)
        np.testing.assert_array_equal(out, z[f"out_{key}"], err_msg=key)
        if not (use_mask == 0):
            # with a mask the reference builds a new array: the raw flow keeps only the filters' effect
            np.testing.assert_array_equal(raw, z[f"raw_after_{key}"], err_msg=key)
