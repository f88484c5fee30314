"""GPU parity, remap half: libtfhip.so (through the C ABI) against vectors the
reference itself produced (tests/golden, tools/capture_golden.py) and against the
numpy oracle on larger seeded inputs.  Bit-exact: integer/byte/index work."""
import os

import numpy as np
import pytest

from oracle import remap_ref as R
from tests.helpers import GOLDEN, PRM_KEYS, case_cfg, layer_case_files, oracle_params

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def tf():
    from transflow_amd import farneback, remap
    return farneback, remap


def _pp(tf, flow, direction):
    farneback, _ = tf
    h, w, _ = flow.shape
    fb = farneback.Farneback(w, h, levels=0)
    out = fb.post_process_host(np.ascontiguousarray(flow.copy()), direction)
    fb.close()
    return out


def test_post_process_golden(tf):
    z = np.load(os.path.join(GOLDEN, "remap_post_process.npz"))
    for i in range(int(z["count"])):
        out = _pp(tf, z[f"in_{i}"], int(z[f"dir_{i}"]))
        np.testing.assert_array_equal(out, z[f"out_{i}"], err_msg=f"case {i}")


@pytest.mark.parametrize("shape,sigma", [((270, 480), 4.0), ((1080, 1920), 6.0), ((37, 1), 2.0), ((1, 1), 1.0)])
def test_post_process_vs_oracle_large(tf, shape, sigma):
    rng = np.random.default_rng(7)
    flow = rng.normal(0, sigma, (*shape, 2)).astype(np.float32)
    for d in (R.FORWARD, R.BACKWARD):
        exp = R.post_process(flow.copy(), d)
        got = _pp(tf, flow, d)
        np.testing.assert_array_equal(got, exp)
        # idempotence of the final clip: a BACKWARD pass over the result changes nothing
        np.testing.assert_array_equal(_pp(tf, got, R.BACKWARD), got)


def _make_layer(remap, h, w, cfg, z):
    kw = {k: v for k, v in cfg.items() if k in PRM_KEYS}
    return remap.RemapLayer(h, w, mask_src=z["mask_src"], mask_dst=z["mask_dst"], mask_alpha=z["mask_alpha"],
                            reset_mask=z["reset_mask"], **kw)


@pytest.mark.parametrize("path", layer_case_files(), ids=lambda p: os.path.basename(p)[12:-4])
def test_layer_sequences_golden(tf, path):
    _, remap = tf
    z = np.load(path)
    h, w = int(z["h"]), int(z["w"])
    cfg = case_cfg(z)
    ns = int(z["nsources"])
    layer = _make_layer(remap, h, w, cfg, z)
    layer.set_sources([z[f"intro_{s}"] for s in range(ns)])
    comp = remap.CompImage(h, w, tuple(int(v) for v in z["background"]))
    data, _ = layer.get_state()
    np.testing.assert_array_equal(data, z["data_init"])
    for t in range(int(z["nframes"])):
        layer.update(z[f"flow_{t}"], z[f"u_{t}"])
        for s in range(ns):
            layer.gather(s, z[f"pixmap_{s}"][t])
        data, rgba = layer.get_state()
        np.testing.assert_array_equal(data, z[f"data_{t}"], err_msg=f"data t={t}")
        np.testing.assert_array_equal(rgba, z[f"rgba_{t}"], err_msg=f"rgba t={t}")
        comp.begin()
        layer.render(comp)
        np.testing.assert_array_equal(comp.download(), z[f"frame_{t}"], err_msg=f"frame t={t}")
        _, rgba = layer.get_state()
        np.testing.assert_array_equal(rgba, z[f"rgba_after_render_{t}"])


def test_random_layer_configurations_vs_oracle(tf):
    """Beyond the 18 captured cases: random moveref configurations -- every move flag, every reset mode,
    masks present or not, 1-3 sources of 3 or 4 channels, shapes down to one pixel -- three frames each,
    through the separate calls and (single-source cases) through the fused step; bit for bit."""
    from transflow_amd.device import DevBuffer
    _, remap = tf
    rng = np.random.default_rng(4242)
    for trial in range(40):
        h, w = int(rng.integers(1, 40)), int(rng.integers(1, 60))
        cfg = dict(transparent_pixels_can_move=bool(rng.integers(2)), pixels_can_move_to_empty_spot=bool(rng.integers(2)),
                   pixels_can_move_to_filled_spot=bool(rng.integers(2)), moving_pixels_leave_empty_spot=bool(rng.integers(2)),
                   reset_mode=str(rng.choice(["off", "random", "constant", "linear"])),
                   reset_random_factor=float(rng.choice([0.0, 0.3, 1.0])), reset_constant_step=float(rng.choice([0.5, 1.0, 2.5])),
                   reset_linear_factor=float(rng.choice([0.1, 0.5])), reset_source=bool(rng.integers(2)))
        masks = {}
        if rng.integers(2):
            masks["mask_src"] = rng.random((h, w)) < 0.8
        if rng.integers(2):
            masks["mask_dst"] = rng.random((h, w)) < 0.8
        if rng.integers(2):
            masks["mask_alpha"] = rng.choice([0.0, 0.5, 1.0], (h, w)).astype(np.float32)
        if rng.integers(2):
            masks["reset_mask"] = rng.random((h, w)).astype(np.float32)
        ns = int(rng.integers(1, 4))
        intro = [rng.random((h, w)) < 0.5 for _ in range(ns)]
        chans = [int(rng.choice([3, 4])) for _ in range(ns)]
        bg = tuple(int(v) for v in rng.integers(0, 256, 3))
        ora = R.MoveRefLayer(h, w, oracle_params(cfg), introduction_masks=intro, **masks)
        layer = remap.RemapLayer(h, w, **cfg, **{k: (v.astype(np.uint8) if v.dtype == bool else v) for k, v in masks.items()})
        layer.set_sources([m.astype(np.uint8) for m in intro])
        fused = remap.RemapLayer(h, w, **cfg, **{k: (v.astype(np.uint8) if v.dtype == bool else v) for k, v in masks.items()})
        fused.set_sources([m.astype(np.uint8) for m in intro])
        comp, comp_f = remap.CompImage(h, w, bg), remap.CompImage(h, w, bg)
        np.testing.assert_array_equal(layer.get_state()[0], ora.data, err_msg=f"trial {trial} init")
        for t in range(3):
            flow = R.post_process(rng.normal(0, 3, (h, w, 2)).astype(np.float32), R.BACKWARD)
            u = rng.random((h, w))
            pms = [rng.integers(0, 256, (h, w, c), dtype=np.uint8) for c in chans]
            ora.update(flow, pms, u)
            layer.update(flow, u)
            for k, pm in enumerate(pms):
                layer.gather(k, pm)
            msg = f"trial {trial} ({h}x{w}, {cfg}, masks {sorted(masks)}, {ns} sources) frame {t}"
            np.testing.assert_array_equal(layer.get_state()[0], ora.data, err_msg=msg)
            np.testing.assert_array_equal(layer.get_state()[1], ora.rgba, err_msg=msg)
            comp.begin()
            layer.render(comp)
            exp = R.composite(np.broadcast_to(np.uint8(bg), (h, w, 3)), [ora.render()])
            np.testing.assert_array_equal(comp.download(), exp, err_msg=msg)
            if ns == 1:
                bufs = [DevBuffer.from_array(a) for a in (flow, pms[0], u)]
                fused.step_dev(comp_f, bufs[0].ptr, bufs[1].ptr, chans[0], uniform_dev=bufs[2].ptr)
                np.testing.assert_array_equal(fused.get_state()[0], ora.data, err_msg="fused " + msg)
                np.testing.assert_array_equal(comp_f.download(), exp, err_msg="fused " + msg)
                for b in bufs:
                    b.close()
        assert not layer.out_of_frame()


def test_reference_known_answers(tf):
    """reference tests/test_compositor.py:20-54 through the C ABI."""
    _, remap = tf
    z = np.load(os.path.join(GOLDEN, "remap_known_answers.npz"))
    comp = remap.CompImage(1, 1, (255, 128, 0))
    assert tuple(comp.download()[0, 0]) == (255, 128, 0)
    flow = np.array([[[0, 1], [0, 1], [0, 0]], [[0, 0], [0, 0], [0, 0]]], np.float32)
    layer = remap.RemapLayer(2, 3)
    layer.update(flow)
    data, _ = layer.get_state()
    assert tuple(data[0, 0, :2]) == (1, 0) and tuple(data[0, 1, :2]) == (1, 1)
    np.testing.assert_array_equal(data, z["moveref_data"])
    layer = remap.RemapLayer(2, 3, reset_mode="random", reset_random_factor=1)
    layer.update(flow, seed=123)   # on-device uniform field: u < 1 always
    np.testing.assert_array_equal(layer.get_state()[0], z["moveref_reset_data"])
    layer = remap.RemapLayer(2, 3, reset_mode="random", reset_random_factor=1, reset_mask=z["moveref_reset_mask"])
    layer.update(flow, seed=5)
    np.testing.assert_array_equal(layer.get_state()[0], z["moveref_reset_mask_data"])


def test_multilayer_golden(tf):
    _, remap = tf
    z = np.load(os.path.join(GOLDEN, "remap_multilayer.npz"))
    h, w = int(z["h"]), int(z["w"])
    l0 = remap.RemapLayer(h, w)
    l1 = remap.RemapLayer(h, w, moving_pixels_leave_empty_spot=True, mask_alpha=z["mask_alpha_l1"])
    ones = np.ones((h, w), np.uint8)
    l0.set_sources([ones])
    l1.set_sources([ones])
    comp = remap.CompImage(h, w, tuple(int(v) for v in z["background"]))
    for t in range(3):
        for layer, pm in ((l0, z["pixmap_l0"]), (l1, z["pixmap_l1"])):
            layer.update(z[f"flow_{t}"])
            layer.gather(0, pm[t])
        np.testing.assert_array_equal(l0.get_state()[0], z[f"data_l0_{t}"])
        np.testing.assert_array_equal(l1.get_state()[0], z[f"data_l1_{t}"])
        comp.begin()
        l0.render(comp)
        l1.render(comp)
        np.testing.assert_array_equal(comp.download(), z[f"frame_{t}"])


@pytest.mark.parametrize("h,w", [(1080, 1920), (2160, 3840), (479, 853)])
def test_full_size_recurrence_vs_oracle(tf, h, w):
    """BASELINE sizes: 3-frame recurrence with every mask, random reset, leave-empty
    and transparent moves, against the numpy oracle; then size-independent
    properties (zero flow is the identity; state round-trips)."""
    _, remap = tf
    rng = np.random.default_rng(h * 7 + w)
    prm = R.LayerParams(transparent_pixels_can_move=True, moving_pixels_leave_empty_spot=True,
                        reset_mode="random", reset_random_factor=0.5)
    msrc = rng.random((h, w)) < 0.9
    mdst = rng.random((h, w)) < 0.9
    malpha = rng.random((h, w)).astype(np.float32)
    rmask = rng.random((h, w)).astype(np.float32)
    ones = np.ones((h, w), bool)
    ora = R.MoveRefLayer(h, w, prm, msrc, mdst, malpha, rmask, [ones])
    gpu = remap.RemapLayer(h, w, transparent_pixels_can_move=True, moving_pixels_leave_empty_spot=True,
                           reset_mode="random", reset_random_factor=0.5, mask_src=msrc, mask_dst=mdst,
                           mask_alpha=malpha, reset_mask=rmask)
    gpu.set_sources([ones])
    comp = remap.CompImage(h, w, (10, 20, 30))
    bg = np.broadcast_to(np.uint8([10, 20, 30]), (h, w, 3))
    for t in range(3):
        flow = R.post_process(rng.normal(0, 5, (h, w, 2)).astype(np.float32), R.BACKWARD)
        u = rng.random((h, w))
        pm = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        ora.update(flow, [pm], u)
        gpu.update(flow, u)
        gpu.gather(0, pm)
        data, rgba = gpu.get_state()
        np.testing.assert_array_equal(data, ora.data)
        np.testing.assert_array_equal(rgba, ora.rgba)
        comp.begin()
        gpu.render(comp)
        np.testing.assert_array_equal(comp.download(), R.composite(bg, [ora.render()]))
    # zero flow + no reset leaves data untouched
    still = remap.RemapLayer(h, w)
    still.set_state(data=ora.data)
    still.update(np.zeros((h, w, 2), np.float32))
    np.testing.assert_array_equal(still.get_state()[0], ora.data)


def test_on_device_uniform_statistics(tf):
    """With uniform=None the field is drawn on the GPU: the reset fraction must match
    reset_random_factor * mean(mask) and differ from frame to frame."""
    _, remap = tf
    h, w = 512, 512
    layer = remap.RemapLayer(h, w, reset_mode="random", reset_random_factor=0.25)
    flow = np.zeros((h, w, 2), np.float32)
    flow[..., 0] = 1.0
    flow = R.post_process(flow, R.BACKWARD)
    base = R.init_data(h, w)
    fracs = []
    prev = None
    for t in range(3):
        layer.set_state(data=np.where(np.ones((h, w, 1), bool), base + np.int32([1, 1, 0, 0]), base))
        layer.update(np.zeros((h, w, 2), np.float32), seed=99)
        data, _ = layer.get_state()
        was_reset = (data[..., 0] == base[..., 0]) & (data[..., 1] == base[..., 1])
        fracs.append(was_reset.mean())
        if prev is not None:
            assert (was_reset != prev).any()
        prev = was_reset
    assert all(abs(f - 0.25) < 0.01 for f in fracs), fracs


def test_out_of_frame_is_index_error(tf):
    _, remap = tf
    layer = remap.RemapLayer(4, 4)
    flow = np.zeros((4, 4, 2), np.float32)
    flow[3, 3] = (2, 0)
    before = layer.get_state()[0]
    with pytest.raises(IndexError):
        layer.update(flow)
    np.testing.assert_array_equal(layer.get_state()[0], before)


def test_argument_errors(tf):
    _, remap = tf
    with pytest.raises(ValueError):
        remap.RemapLayer(4, 4, reset_mode="sometimes")
    layer = remap.RemapLayer(4, 4)
    with pytest.raises(ValueError):
        layer.update(np.zeros((4, 5, 2), np.float32))
    with pytest.raises(ValueError):
        layer.gather(0, np.zeros((4, 4, 2), np.uint8))
    # empty frame: every call is a no-op
    empty = remap.RemapLayer(0, 0)
    empty.update(np.zeros((0, 0, 2), np.float32))
    assert empty.get_state()[0].shape == (0, 0, 4)


@pytest.mark.parametrize("px", [4, 2, 1])
@pytest.mark.parametrize("path", layer_case_files(), ids=lambda p: os.path.basename(p)[12:-4])
def test_fused_step_matches_golden(tf, lib_option, path, px):
    """tf_remap_step_dev (one kernel when the layer allows it, the separate kernels otherwise)
    on the reference's vectors: single-source cases, flow/pixmap/u resident in HBM; every form of the one
    kernel (option "remap_px": 4, 2 or 1 pixels per thread)."""
    from transflow_amd.device import DevBuffer
    _, remap = tf
    z = np.load(path)
    if int(z["nsources"]) != 1:
        pytest.skip("the fused step serves one source")
    lib_option("remap_px", px)
    h, w = int(z["h"]), int(z["w"])
    layer = _make_layer(remap, h, w, case_cfg(z), z)
    layer.set_sources([z["intro_0"]])
    comp = remap.CompImage(h, w, tuple(int(v) for v in z["background"]))
    for t in range(int(z["nframes"])):
        flow = DevBuffer.from_array(z[f"flow_{t}"])
        u = DevBuffer.from_array(z[f"u_{t}"])
        pm = DevBuffer.from_array(z["pixmap_0"][t])
        layer.step_dev(comp, flow.ptr, pm.ptr, channels=z["pixmap_0"].shape[-1], clip_flow=True,
                       uniform_dev=u.ptr)
        data, rgba = layer.get_state()
        np.testing.assert_array_equal(data, z[f"data_{t}"], err_msg=f"data t={t}")
        np.testing.assert_array_equal(rgba, z[f"rgba_after_render_{t}"], err_msg=f"rgba t={t}")
        np.testing.assert_array_equal(comp.download(), z[f"frame_{t}"], err_msg=f"frame t={t}")
    assert not layer.out_of_frame()


def test_fused_step_clips_like_post_process(tf):
    """clip_flow=True on a raw flow == BACKWARD post_process then the separate calls."""
    from transflow_amd.device import DevBuffer
    _, remap = tf
    h, w = 203, 317
    rng = np.random.default_rng(12)
    raw = rng.normal(0, 30, (h, w, 2)).astype(np.float32)
    pm = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    ora = R.MoveRefLayer(h, w, introduction_masks=[np.ones((h, w), bool)])
    ora.update(R.post_process(raw.copy(), R.BACKWARD), [pm])
    layer = remap.RemapLayer(h, w)
    layer.set_sources([np.ones((h, w), np.uint8)])
    comp = remap.CompImage(h, w, (1, 2, 3))
    raw_dev, pm_dev = DevBuffer.from_array(raw), DevBuffer.from_array(pm)  # keep the buffers alive
    layer.step_dev(comp, raw_dev.ptr, pm_dev.ptr, 3, clip_flow=True)
    np.testing.assert_array_equal(layer.get_state()[0], ora.data)
    exp = R.composite(np.broadcast_to(np.uint8([1, 2, 3]), (h, w, 3)), [ora.render()])
    np.testing.assert_array_equal(comp.download(), exp)
    # without the clip the same flow leaves the frame: flagged, offending pixels stay put
    layer.step_dev(comp, raw_dev.ptr, pm_dev.ptr, 3, clip_flow=False)
    assert layer.out_of_frame()


@pytest.mark.parametrize("shape", [(203, 317), (1, 1), (1, 9), (7, 1), (2, 3), (33, 64)])
@pytest.mark.parametrize("leave_empty", [False, True])     # one kernel / the separate kernels
def test_fused_step_finishes_forward_post_process(tf, leave_empty, shape):
    """FORWARD post_process split in two -- tf_fb_post_process_scatter, then step_dev(clip_flow=2) forming
    the flow from the winner map in registers -- equals post_process then the step, and the oracle."""
    import ctypes as C
    from transflow_amd import _lib
    from transflow_amd.device import DevBuffer
    farneback, remap = tf
    h, w = shape
    rng = np.random.default_rng(13)
    raw = rng.normal(0, 12, (h, w, 2)).astype(np.float32)      # collisions and out-of-frame targets
    raw[rng.random((h, w)) < 0.3] = 0
    pm = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    kw = dict(moving_pixels_leave_empty_spot=True) if leave_empty else {}
    ora = R.MoveRefLayer(h, w, oracle_params(kw), introduction_masks=[np.ones((h, w), bool)])
    fb = farneback.Farneback(w, h, levels=0)
    fb.calc_slots([0], [0])                                    # gives the handle a result buffer
    lib = _lib.load()
    pm_dev = DevBuffer.from_array(pm)
    layers = []
    for mode in ("two calls", "split"):
        layer = remap.RemapLayer(h, w, **kw)
        layer.set_sources([np.ones((h, w), np.uint8)])
        comp = remap.CompImage(h, w, (1, 2, 3))
        for t in range(3):                                     # a recurrence, so the state matters
            flow_t = np.ascontiguousarray(np.roll(raw, 7 * t, axis=1))
            _lib.check(lib.tf_dev_upload(C.c_void_p(fb.flow_ptr(0)), C.c_void_p(flow_t.ctypes.data), flow_t.nbytes))
            if mode == "two calls":
                fb.post_process(0, 0)
                layer.step_dev(comp, fb.flow_ptr(0), pm_dev.ptr, 3, clip_flow=False)
            else:
                layer.step_dev(comp, fb.post_process_scatter(0), pm_dev.ptr, 3, clip_flow=2)
            if mode == "split":
                ora.update(R.post_process(flow_t.copy(), R.FORWARD), [pm])
        layers.append((layer.get_state(), comp.download()))
        assert not layer.out_of_frame()
    (d0, r0), i0 = layers[0]
    (d1, r1), i1 = layers[1]
    np.testing.assert_array_equal(d0, d1)
    np.testing.assert_array_equal(r0, r1)
    np.testing.assert_array_equal(i0, i1)
    np.testing.assert_array_equal(d1, ora.data)
    exp = R.composite(np.broadcast_to(np.uint8([1, 2, 3]), (h, w, 3)), [ora.render()])
    np.testing.assert_array_equal(i1, exp)
    fb.close()


@pytest.mark.parametrize("no_pack", [0, 2])
def test_fused_step_state_in_int16_round_trips(tf, lib_option, no_pack):
    """The fused step keeps the layer state as one 32-bit word per pixel (row 13, column 13, alpha 1, source 5 bits; option
    remap_no_pack = 0, the default) or as int16 x 4 (= 2) between its own launches; any other entry point sees the
    reference's int32 layout again.  Interleaves fused steps, a host update, state reads and restored checkpoints whose
    values do not fit the word (an alpha of 3: the step goes on in int16), then not int16 either (it stays on int32)."""
    from transflow_amd.device import DevBuffer
    lib_option("remap_no_pack", no_pack)
    _, remap = tf
    h, w = 120, 173
    rng = np.random.default_rng(14)
    pm = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    pm_dev = DevBuffer.from_array(pm)
    ora = R.MoveRefLayer(h, w, introduction_masks=[np.ones((h, w), bool)])
    layer = remap.RemapLayer(h, w)
    layer.set_sources([np.ones((h, w), np.uint8)])
    comp = remap.CompImage(h, w, (9, 8, 7))

    def flow_t(t):
        return R.post_process(rng.normal(0, 3, (h, w, 2)).astype(np.float32), R.BACKWARD)

    def fused(f):
        buf = DevBuffer.from_array(f)
        layer.step_dev(comp, buf.ptr, pm_dev.ptr, 3)
        ora.update(f, [pm])
        buf.close()

    for t in range(2):
        fused(flow_t(t))
    np.testing.assert_array_equal(layer.get_state()[0], ora.data)          # read in the middle of a run
    fused(flow_t(2))
    f = flow_t(3)                                                          # the separate calls on the same layer
    layer.update(f)
    layer.gather(0, pm)
    ora.update(f, [pm])
    fused(flow_t(4))
    data, rgba = layer.get_state()
    np.testing.assert_array_equal(data, ora.data)
    np.testing.assert_array_equal(rgba, ora.rgba)
    exp = R.composite(np.broadcast_to(np.uint8([9, 8, 7]), (h, w, 3)), [ora.render()])
    np.testing.assert_array_equal(comp.download(), exp)
    # a checkpoint with a value outside int16 (extra/control.py may write anything into layer.data)
    data = data.copy()
    data[5, 6, 0] = 100000
    data[7, 8, 1] = -70000
    layer.set_state(data, rgba)
    ora.data[...] = data
    for t in range(5, 7):
        fused(flow_t(t))
    np.testing.assert_array_equal(layer.get_state()[0], ora.data)
    assert (layer.get_state()[0] == 100000).any()
    # ... and back to values that fit
    layer.set_state(np.clip(layer.get_state()[0], 0, 100), None)
    ora.data[...] = np.clip(ora.data, 0, 100)
    fused(flow_t(7))
    np.testing.assert_array_equal(layer.get_state()[0], ora.data)
    # a checkpoint with an alpha the 32-bit word cannot hold (it has one bit for it): int16 x 4 takes over
    data = layer.get_state()[0].copy()
    data[3, 4, 2] = 3
    data[60:70, 80:90, 2] = 2
    layer.set_state(data, None)
    ora.data[...] = data
    zero = np.zeros((h, w, 2), np.float32)
    fused(zero)                                   # nothing moves: the odd alphas must come through the step unchanged
    got = layer.get_state()[0]
    np.testing.assert_array_equal(got, ora.data)
    assert got[3, 4, 2] == 3 and (got[60:70, 80:90, 2] == 2).all()
    fused(flow_t(8))
    np.testing.assert_array_equal(layer.get_state()[0], ora.data)
    assert not layer.out_of_frame()


@pytest.mark.parametrize("keep", [0, 1])
@pytest.mark.parametrize("case", ["all selected", "holes", "leave empty", "rgba pixmaps"])
def test_steps_call_equals_the_single_steps(tf, lib_option, case, keep):
    """tf_remap_steps_dev = n tf_remap_step_dev calls: layer state, rgba and every frame the same bytes, and both equal to
    the oracle's layer.  The call stores the layer's rgba in its last step alone while every pixel is selected by source 0
    ("all selected": a fresh layer with a random reset through a mask); with pixels that are not ("holes": a checkpoint
    with alpha 0 in a region, whose pixels must keep the colour they had) every step stores it; "leave empty" takes the
    separate kernels; option remap_keep_rgba = 1 is the call without the elision."""
    from transflow_amd.device import DevBuffer
    lib_option("remap_keep_rgba", keep)
    _, remap = tf
    h, w, n = 97, 141, 5
    ch = 4 if case == "rgba pixmaps" else 3
    rng = np.random.default_rng(31)
    pms = [rng.integers(0, 256, (h, w, ch), dtype=np.uint8) for _ in range(n)]
    if ch == 4:
        for pm in pms:
            pm[..., 3] = rng.integers(0, 2, (h, w))
    mask = rng.random((h, w), dtype=np.float32)
    kw = dict(moving_pixels_leave_empty_spot=True) if case == "leave empty" else dict(reset_mode="random", reset_random_factor=0.3)

    def make():
        layer = remap.RemapLayer(h, w, reset_mask=None if case == "leave empty" else mask, **kw)
        layer.set_sources([np.ones((h, w), np.uint8)])
        return layer

    one, many = make(), make()
    prm = R.LayerParams(**kw)
    ora = R.MoveRefLayer(h, w, prm, reset_mask=None if case == "leave empty" else mask, introduction_masks=[np.ones((h, w), bool)])
    if case == "holes":
        data, rgba = one.get_state()
        data = data.copy()
        rgba = rng.integers(0, 256, (h, w, 4), dtype=np.uint8)
        data[20:50, 30:90, 2] = 0
        for layer in (one, many):
            layer.set_state(data, rgba)
        ora.data[...] = data
        ora.rgba[...] = rgba
    flows = [R.post_process(rng.normal(0, 2.5, (h, w, 2)).astype(np.float32), R.BACKWARD) for _ in range(n)]
    fbuf = [DevBuffer.from_array(f) for f in flows]
    pbuf = [DevBuffer.from_array(pm) for pm in pms]
    ubuf = [DevBuffer(h * w * 8) for _ in range(n)]
    comp1 = remap.CompImage(h, w, (3, 2, 1))
    compn = [remap.CompImage(h, w, (3, 2, 1)) for _ in range(n)]
    singles = []
    bg = np.broadcast_to(np.uint8([3, 2, 1]), (h, w, 3))
    for i in range(n):
        one.uniform_dev(77, ubuf[i].ptr)           # the field step i draws: the oracle is handed the same one
        u = ubuf[i].download((h, w), np.float64)
        one.step_dev(comp1, fbuf[i].ptr, pbuf[i].ptr, ch, seed=77)
        singles.append(comp1.download())
        if case == "leave empty":
            ora.update(flows[i], [pms[i]])
        else:
            ora.update(flows[i], [pms[i]], u)
        np.testing.assert_array_equal(singles[-1], R.composite(bg, [ora.render()]), err_msg=f"single step {i} vs oracle")
    many.steps_dev(compn, [b.ptr for b in fbuf], [b.ptr for b in pbuf], ch, seed=77)
    for i in range(n):
        np.testing.assert_array_equal(compn[i].download(), singles[i], err_msg=f"frame {i}")
    (d1, r1), (d2, r2) = one.get_state(), many.get_state()
    np.testing.assert_array_equal(d2, d1)
    np.testing.assert_array_equal(r2, r1)
    np.testing.assert_array_equal(d2, ora.data)
    np.testing.assert_array_equal(r2, ora.rgba)
    # the call again on the layer it left: the state it left is the state the single steps left
    many.steps_dev(compn[:2], [fbuf[0].ptr, fbuf[1].ptr], [pbuf[0].ptr, pbuf[1].ptr], ch, seed=77)
    for i in range(2):
        one.step_dev(comp1, fbuf[i].ptr, pbuf[i].ptr, ch, seed=77)
        np.testing.assert_array_equal(compn[i].download(), comp1.download())
    np.testing.assert_array_equal(many.get_state()[1], one.get_state()[1])
    with pytest.raises(ValueError):
        many.steps_dev(compn[:2], [fbuf[0].ptr], [pbuf[0].ptr, pbuf[1].ptr], ch)
    # a step whose arguments are bad is refused BEFORE the first step runs: an error once rgba stores have been left
    # out would leave the layer's state ahead of its rgba (round 6; a compositor image of another size in step 1)
    before = many.get_state()
    other = remap.CompImage(h + 1, w, (9, 9, 9))
    with pytest.raises(ValueError):
        many.steps_dev([compn[0], other], [fbuf[0].ptr, fbuf[1].ptr], [pbuf[0].ptr, pbuf[1].ptr], ch, seed=77)
    with pytest.raises(ValueError):
        many.steps_dev(compn[:2], [fbuf[0].ptr, 0], [pbuf[0].ptr, pbuf[1].ptr], ch, seed=77)
    after = many.get_state()
    np.testing.assert_array_equal(after[0], before[0])
    np.testing.assert_array_equal(after[1], before[1])
    other.close()


def test_flow_presteps_golden_gpu(tf):
    """scale / threshold / clip filters and the flow mask on the GPU (tf_fb_post_process_host_ex),
    through the FlowSource mirror, against the reference's outputs -- bit for bit, including which
    array the reference mutates in place (NaNs of clip=0 at a zero vector included)."""
    from transflow_amd.flow import FlowFilter, FlowSource
    z = np.load(os.path.join(GOLDEN, "flow_presteps.npz"))
    h, w = z["mask"].shape[:2]
    for i in range(int(z["count"])):
        for direction in (0, 1):
            for use_mask in (0, 1):
                key = f"{i}_{direction}_{use_mask}"
                filters = [FlowFilter.from_string(p) for p in str(z["specs"][i]).split(";")]
                fs = FlowSource(direction, w, h, 30.0, None, 0, 0, 0, mask=z["mask"] if use_mask else None,
                                flow_filters=filters)
                fs.output_frame_index = int(round(float(z[f"t_{key}"]) * fs.framerate))
                assert fs.t == float(z[f"t_{key}"])
                work = z[f"in_{key}"].copy()
                out = fs.post_process(work)
                np.testing.assert_array_equal(out, z[f"out_{key}"], err_msg=key)
                np.testing.assert_array_equal(work, z[f"raw_after_{key}"], err_msg="raw " + key)
                assert (out is work) == (not use_mask)
                fs.close()


def test_flow_presteps_device_path(tf):
    """The resident form (tf_fb_post_process_ex) equals the host form."""
    from transflow_amd.device import DevBuffer
    farneback, _ = tf
    h, w = 120, 160
    rng = np.random.default_rng(8)
    fb = farneback.Farneback(w, h, levels=0)
    a = rng.integers(0, 255, (h, w), dtype=np.uint8)
    b = np.roll(a, 2, axis=1)
    fb.set_frame(0, a)
    fb.set_frame(1, b)
    fb.calc_slots([0], [1])
    raw = fb.get_flow(0)
    mask = rng.random((h, w)).astype(np.float32)
    ops = [("scale", 1.5), ("clip", np.float64(2.0)), ("threshold", 0.2)]
    exp = fb.post_process_host_ex(raw.copy(), 0, ops, mask)
    mdev = DevBuffer.from_array(mask)
    fb.post_process_ex(0, 0, ops, mdev.ptr)
    np.testing.assert_array_equal(fb.get_flow(0), exp)
    with pytest.raises(NotImplementedError):
        fb.post_process_host_ex(raw.copy(), 0, [("scale", np.ones(3))])
    with pytest.raises(ValueError):
        fb.post_process_host_ex(raw.copy(), 0, [("scale", 1.0)] * 9)
    fb.close()


@pytest.mark.parametrize("h,w,px", [(1080, 1920, 4), (2160, 3840, 4), (1080, 1920, 2), (1079, 1918, 4), (1080, 1920, 1)])
@pytest.mark.parametrize("direction", ["backward", "forward"])
def test_timed_remap_kernel_at_full_size_with_its_own_uniform(tf, lib_option, h, w, px, direction):
    """What bench.py times for the remap, at the sizes it times it: the one-kernel step
    (k_remap_step_px<3, int16 state, `px` pixels per thread: option "remap_px", 4 by default) with the uniform field drawn ON THE GPU
    (Philox), random reset p = 0.5 through a float mask, three frames.  tf_remap_uniform_dev hands out
    the very field the kernel is about to draw, so the numpy oracle (reference.py:58-67 with that u)
    must agree bit for bit on data, rgba and the frame.  backward: raw flow, clip folded in
    (clip_flow=1, source.py:361-362); forward: the winner map of tf_fb_post_process_scatter, the rest
    of FORWARD post_process formed in registers (clip_flow=2, source.py:349-362)."""
    from transflow_amd.device import DevBuffer
    farneback, remap = tf
    lib_option("remap_px", px)
    rng = np.random.default_rng(h + 3 * w + len(direction))
    rmask = rng.random((h, w), dtype=np.float32)
    ones = np.ones((h, w), bool)
    prm = R.LayerParams(reset_mode="random", reset_random_factor=0.5)
    ora = R.MoveRefLayer(h, w, prm, reset_mask=rmask, introduction_masks=[ones])
    gpu = remap.RemapLayer(h, w, reset_mode="random", reset_random_factor=0.5, reset_mask=rmask)
    gpu.set_sources([ones])
    comp = remap.CompImage(h, w, (255, 255, 255))
    white = np.full((h, w, 3), 255, np.uint8)
    pm = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    pm_dev = DevBuffer.from_array(pm)
    ubuf = DevBuffer(h * w * 8)
    seed = 20251003
    fb = farneback.Farneback(w, h, levels=0, frame_slots=2, max_pairs=1) if direction == "forward" else None
    seen = []
    for t in range(3):
        raw = rng.normal(0, 4, (h, w, 2)).astype(np.float32)
        raw[:8, :8] = 1e4                                    # vectors far outside the frame: the clip must catch them
        gpu.uniform_dev(seed, ubuf.ptr)
        u = ubuf.download((h, w), np.float64)
        assert 0.0 <= u.min() and u.max() < 1.0
        seen.append(u[::97, ::89].copy())
        if direction == "backward":
            flow_dev = DevBuffer.from_array(raw)
            gpu.step_dev(comp, flow_dev.ptr, pm_dev.ptr, 3, clip_flow=True, seed=seed)
            flow = R.post_process(raw.copy(), R.BACKWARD)
        else:
            # the flow the handle holds for pair 0 is replaced by `raw`, then scattered
            check_ptr = fb.flow_ptr(0)
            import ctypes as C
            from transflow_amd import _lib
            _lib.check(_lib.load().tf_dev_upload(C.c_void_p(check_ptr), C.c_void_p(raw.ctypes.data), raw.nbytes))
            gpu.step_dev(comp, fb.post_process_scatter(0), pm_dev.ptr, 3, clip_flow=2, seed=seed)
            flow = R.post_process(raw.copy(), R.FORWARD)
        ora.update(flow, [pm], u)
        data, rgba = gpu.get_state()
        np.testing.assert_array_equal(data, ora.data, err_msg=f"data t={t}")
        np.testing.assert_array_equal(rgba, ora.rgba, err_msg=f"rgba t={t}")
        np.testing.assert_array_equal(comp.download(), R.composite(white, [ora.render()]), err_msg=f"frame t={t}")
    assert not gpu.out_of_frame()
    assert not np.array_equal(seen[0], seen[1])              # a new field per frame
    reset_fraction = float((u < 0.5 * rmask).mean())
    assert abs(reset_fraction - 0.25) < 0.01


@pytest.mark.parametrize("layer_class", ["moveref", "sum"])
def test_float64_flow_is_rounded_in_float64(tf, layer_class):
    """post_process hands on a float64 flow after a float64 convolution kernel (source.py:344-348); the
    layers round that array itself (movement.py:21, sum.py:10).  Values a float32 cast would push across
    a rounding boundary (1.4999999999 -> 1.5 -> 2; -0.0000000001 -> -0. -> floor 0 instead of -1)."""
    _, remap = tf
    h, w = 12, 16
    rng = np.random.default_rng(64)
    flow = np.zeros((h, w, 2), np.float64)
    flow[4:9, 4:12, 0] = rng.choice([1.4999999999, -1.4999999999, 0.5000000001, 2.5000000001, -1e-10], (5, 8))
    flow[4:9, 4:12, 1] = rng.choice([1.4999999999, -0.4999999999, -2.5000000001, -1e-10], (5, 8))
    assert not np.array_equal(np.rint(flow), np.rint(flow.astype(np.float32)))        # the cast would matter
    ones = np.ones((h, w), bool)
    pm = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
    if layer_class == "moveref":
        ora = R.MoveRefLayer(h, w, R.LayerParams(), introduction_masks=[ones])
    else:
        ora = R.SumLayer(h, w, R.LayerParams(), introduction_masks=[ones])
    gpu = remap.RemapLayer(h, w, layer_class=layer_class)
    gpu.set_sources([ones])
    for _ in range(2):
        ora.update(flow, [pm], None)
        gpu.update(flow)
        gpu.gather(0, pm)
        np.testing.assert_array_equal(gpu.get_state()[0], ora.data)
