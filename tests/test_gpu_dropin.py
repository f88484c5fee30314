"""GPU parity through the reference-shaped surface: HipFlowSource / HipCompositor used
the way transflow/pipeline.py uses FlowSource / Compositor (pipeline.py:562-567:
flow = next(flow_source); compositor.update(flow); compositor.render())."""
import os
import pickle

import numpy as np
import pytest

from oracle import farneback as OF
from oracle import remap_ref as R
from tests.helpers import GOLDEN, PRM_KEYS, case_cfg, layer_case_files, oracle_params, synth_pair

pytestmark = pytest.mark.gpu


class FakeSource:
    """PixmapSourceInterface stand-in (pixmap_source_interface.py:12-37)."""

    def __init__(self, frames, introduction_mask):
        self.frames, self.introduction_mask, self.counter = list(frames), introduction_mask, -1

    def next(self, timeout=1):
        self.counter += 1
        return self.frames[self.counter % len(self.frames)]


def _frames(h, w, n, seed=5):
    out = []
    for i in range(n):
        a, b = synth_pair(h, w, seed=seed, shift=(0.8 * i, 0.5 * i))
        out.append(b)
    return out


@pytest.mark.parametrize("direction", ["forward", "backward"])
def test_flow_source_matches_oracle(direction):
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    h, w = 120, 160
    frames = _frames(h, w, 4)
    builder = HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction=direction)
    with builder as source:
        # what reference tests/test_flow_source.py:22-33 checks
        assert (source.width, source.height, source.framerate, source.length) == (w, h, 25.0, 3)
        flows = list(source)
    assert len(flows) == 3
    for t, flow in enumerate(flows):
        assert isinstance(flow, np.ndarray) and flow.shape == (h, w, 2) and flow.dtype == np.float32
        prev, nxt = (frames[t], frames[t + 1]) if direction == "forward" else (frames[t + 1], frames[t])
        raw = OF.calc(prev, nxt)
        d = R.FORWARD if direction == "forward" else R.BACKWARD
        if d == R.BACKWARD:
            # the clip only touches vectors that leave the frame; compare within tolerance
            exp = R.post_process(raw.copy(), d)
            assert np.abs(flow - exp).max() <= 1e-4 * max(1.0, float(np.abs(exp).max()))
        else:
            # FORWARD output is an integer gather map of the GPU's own flow: it must be exactly
            # post_process(oracle) wherever rounding the two raw flows agrees
            exp = R.post_process(raw.copy(), d)
            assert (flow == np.rint(flow)).all()
            assert (flow != exp).any(axis=2).mean() < 0.01
        pickle.loads(pickle.dumps(flow))   # what pipeline.py:86 puts on the queue


def test_flow_source_with_initial_flow_flag_follows_the_reference_recurrence():
    """fb_flags = OPTFLOW_USE_INITIAL_FLOW: every call starts from a copy of the previous OUTPUT, which __next__ has
    post-processed in place (cv.py:478 with source.py:312-321), zeros for the first frame."""
    from transflow_amd.config import FlowConfig
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    h, w = 120, 160
    frames = _frames(h, w, 5)
    builder = HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward",
                                      cv_config=FlowConfig(fb_flags=4))
    with builder as source:
        flows = [f.copy() for f in source]
    assert len(flows) == 4
    prev = None
    for t, flow in enumerate(flows):
        raw = OF.calc(frames[t + 1], frames[t], flags=4, flow=prev)       # BACKWARD: (current, previous)
        exp = R.post_process(raw.copy(), R.BACKWARD)
        assert np.abs(flow - exp).max() <= 1e-4 * max(1.0, float(np.abs(exp).max())), t
        prev = flow          # the GPU's own previous output, as the reference would feed its own
    # and it is not the plain recurrence
    plain = OF.calc(frames[2], frames[1])
    assert np.abs(flows[1] - R.post_process(plain.copy(), R.BACKWARD)).max() > 1e-3


def test_flow_source_repeat_and_seek():
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    h, w = 64, 96
    frames = _frames(h, w, 6, seed=9)
    with HipFlowSource.from_args(ArrayFrameProvider(frames, 10.0), direction="backward", seek_time=0.2,
                                 repeat=2) as source:
        assert source.length == 2 * 3
        flows = [f.copy() for f in source]
    assert len(flows) == 6
    for t in range(3):                      # second pass replays the first after the rewind
        np.testing.assert_array_equal(flows[t], flows[t + 3])
    single = OF.calc(frames[3], frames[2])
    assert np.abs(flows[0] - R.post_process(single, R.BACKWARD)).max() <= 1e-4 * max(1, np.abs(single).max())


@pytest.mark.parametrize("path", [p for p in layer_case_files()
                                  if any(k in p for k in ("flags05", "holes13", "masks_bool", "reset_random_p05",
                                                          "two_sources_rgb", "rgba_pixmap", "reset_linear"))],
                         ids=lambda p: os.path.basename(p)[12:-4])
def test_compositor_matches_reference_vectors(path):
    """The reference's own vectors through HipCompositor; the random-reset field is drawn
    from numpy's global generator on the host exactly like reference.py:59."""
    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import LayerConfig
    z = np.load(path)
    h, w = int(z["h"]), int(z["w"])
    cfg = case_cfg(z)
    ns = int(z["nsources"])
    comp = HipCompositor.from_args(h, w, [LayerConfig(0, **cfg)],
                                   background_color="#%02x%02x%02x" % tuple(int(v) for v in z["background"]))
    layer = comp.layers[0]
    # the capture script installed arbitrary mask arrays on the reference layer: do the same here
    layer.mask_src, layer.mask_dst = z["mask_src"], z["mask_dst"]
    layer.mask_alpha, layer.reset_mask = z["mask_alpha"], z["reset_mask"]
    comp.set_sources({0: [FakeSource(z[f"pixmap_{s}"], z[f"intro_{s}"]) for s in range(ns)]})
    orig = np.random.random
    for t in range(int(z["nframes"])):
        np.random.random = lambda size=None, _u=z[f"u_{t}"]: _u.copy()
        try:
            comp.update(z[f"flow_{t}"])
        finally:
            np.random.random = orig
        np.testing.assert_array_equal(layer.data, z[f"data_{t}"], err_msg=f"data t={t}")
        frame = comp.render()
        assert frame.dtype == np.uint8 and frame.shape == (h, w, 3)
        np.testing.assert_array_equal(frame, z[f"frame_{t}"], err_msg=f"frame t={t}")


def test_numpy_seed_reproduces_reference_stream():
    """Same numpy seed => same reset decisions as the reference would take (it draws
    numpy.random.random((H, W)) once per update, reference.py:59)."""
    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import LayerConfig
    h, w = 33, 47
    rng = np.random.default_rng(1)
    flows = [R.post_process(rng.normal(0, 2, (h, w, 2)).astype(np.float32), R.BACKWARD) for _ in range(3)]
    comp = HipCompositor.from_args(h, w, [LayerConfig(0, reset_mode="random", reset_random_factor=0.3)])
    ora = R.MoveRefLayer(h, w, R.LayerParams(reset_mode="random", reset_random_factor=0.3))
    np.random.seed(1234)
    for f in flows:
        comp.update(f)
    state = np.random.RandomState(1234)
    for f in flows:
        ora.update(f, u=state.random_sample((h, w)))
    np.testing.assert_array_equal(comp.layers[0].data, ora.data)


def test_checkpoint_pickle_roundtrip_is_bit_identical():
    """reference tests/test_pipeline.py:90-119: resuming from a pickled compositor
    reproduces the following frames bit for bit."""
    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import LayerConfig
    h, w = 40, 56
    rng = np.random.default_rng(2)
    flows = [R.post_process(rng.normal(0, 3, (h, w, 2)).astype(np.float32), R.BACKWARD) for _ in range(6)]
    pix = rng.integers(0, 256, (6, h, w, 3), dtype=np.uint8)
    cfgs = [LayerConfig(0, moving_pixels_leave_empty_spot=True, reset_mode="linear", reset_linear_factor=0.2)]

    def run(comp, lo, hi, frames_seen):
        out = []
        comp.set_sources({0: [FakeSource(pix[frames_seen:], np.ones((h, w), bool))]})
        for t in range(lo, hi):
            comp.update(flows[t])
            out.append(comp.render())
        return out

    full = HipCompositor.from_args(h, w, cfgs, "#336699")
    ref_frames = run(full, 0, 6, 0)
    part = HipCompositor.from_args(h, w, cfgs, "#336699")
    first = run(part, 0, 3, 0)
    for layer in part.layers:           # pipeline.py:236-238 strips the sources before pickling
        layer.sources = []
    resumed = pickle.loads(pickle.dumps(part))
    assert resumed.layers[0]._dev is None
    np.testing.assert_array_equal(resumed.layers[0].data, part.layers[0].data)   # extra/control.py:155-162
    rest = run(resumed, 3, 6, 3)
    for a, b in zip(first + rest, ref_frames):
        np.testing.assert_array_equal(a, b)


def test_pipeline_loop_matches_oracle_sequence():
    """configs[0] plumbing stand-in: 854x480 frames, defaults, BACKWARD,
    moveref with reset off -- the per-frame sequence of pipeline.py:562-567."""
    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import LayerConfig
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    h, w = 480, 854
    frames = _frames(h, w, 4, seed=21)
    pixmap = np.random.default_rng(3).integers(0, 256, (h, w, 3), dtype=np.uint8)
    comp = HipCompositor.from_args(h, w, [LayerConfig(0)], "#ffffff")
    comp.set_sources({0: [FakeSource([pixmap], np.ones((h, w), bool))]})
    ora = R.MoveRefLayer(h, w, introduction_masks=[np.ones((h, w), bool)])
    white = np.full((h, w, 3), 255, np.uint8)
    with HipFlowSource.from_args(ArrayFrameProvider(frames, 50.0), direction="backward") as source:
        for flow in source:
            comp.update(flow)
            frame = comp.render()
            ora.update(flow, [pixmap])          # oracle remap driven by the SAME flow: bit-exact
            np.testing.assert_array_equal(comp.layers[0].data, ora.data)
            np.testing.assert_array_equal(frame, R.composite(white, [ora.render()]))



def _bgr_frames(h, w, n, seed=5):
    """Colour frames whose grey value carries the synthetic texture: three channels with different gains + noise."""
    rng = np.random.default_rng(seed)
    out = []
    for g in _frames(h, w, n, seed):
        f = np.stack([np.clip(g.astype(np.int32) * k // 8 + rng.integers(0, 24, g.shape), 0, 255) for k in (5, 8, 11)], axis=2)
        out.append(f.astype(np.uint8))
    return out


@pytest.mark.parametrize("direction", ["forward", "backward"])
@pytest.mark.parametrize("src_size,size", [((480, 854), None), ((300, 500), (854, 480)), ((961, 1282), (640, 360))])
def test_flow_source_over_bgr_frames_ingests_on_the_device(direction, src_size, size, lib_option):
    """cv.py:461-466 + 479-490 through the drop-in source: decoded BGR frames go up as they are, the nearest-neighbour
    resize to the size the source reports and the BGR -> grey conversion run on the device (one arithmetic: OpenCV 4's
    15-bit weights, the oracle's), then Farnebäck.  Against frames_ref -> the Farnebäck oracle -> post_process: with
    fb_exact_sums bit-identical in both directions; in the default mode within tolerance (BACKWARD) / an integer map
    that agrees wherever rounding the two raw flows agrees (FORWARD)."""
    from oracle import frames_ref
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    sh, sw = src_size
    frames = _bgr_frames(sh, sw, 3)
    w, h = (sw, sh) if size is None else size
    greys = [frames_ref.bgr_to_grey(f, (w, h)) for f in frames]
    d = R.FORWARD if direction == "forward" else R.BACKWARD
    exp = []
    for t in range(2):
        prev, nxt = (greys[t], greys[t + 1]) if direction == "forward" else (greys[t + 1], greys[t])
        exp.append(R.post_process(OF.calc(prev, nxt), d))
    from transflow_amd.config import FlowConfig
    for exact in (1, 0):
        with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0, size=size), direction=direction,
                                     cv_config=FlowConfig(hip_exact_sums=bool(exact))) as source:
            assert (source.width, source.height, source.length) == (w, h, 2)
            np.testing.assert_array_equal(source.prev_gray, greys[0])      # what cv.py:456 keeps
            flows = [f.copy() for f in source]
        assert len(flows) == 2
        for t, flow in enumerate(flows):
            assert flow.shape == (h, w, 2) and flow.dtype == np.float32
            if exact:
                np.testing.assert_array_equal(flow, exp[t])
            elif d == R.BACKWARD:
                assert np.abs(flow - exp[t]).max() <= 1e-4 * max(1.0, float(np.abs(exp[t]).max()))
            else:
                assert (flow == np.rint(flow)).all() and (flow != exp[t]).any(axis=2).mean() < 0.01


def test_prefetching_flow_source_yields_the_same_flows_from_pinned_arrays():
    """FlowConfig.hip_prefetch: the source runs ahead of its consumer in a worker thread with a library stream of its own
    (what the reference's child process + queue give it, pipeline.py:56-64, 85-86) while the consumer's thread works the
    compositor; flows, their order and the end of the iteration are those of the plain source, the compositor's frames
    too.  Arrays handed out come from the page-locked pool and stay intact while the caller holds them."""
    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import FlowConfig, LayerConfig
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    h, w = 180, 256
    frames = _bgr_frames(h, w, 9)
    pix = np.random.default_rng(3).integers(0, 256, (h, w, 3), dtype=np.uint8)

    class Src:
        introduction_mask = np.ones((h, w), bool)

        def next(self, timeout=1):
            return pix

    def run(cfg):
        comp = HipCompositor.from_args(h, w, [LayerConfig(0)])
        comp.set_sources({0: [Src()]})
        flows, images, held = [], [], []
        with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward", cv_config=cfg) as source:
            for flow in source:
                held.append(flow)                 # keep every array: the pool must never hand one out again
                flows.append(flow.copy())
                comp.update(flow)
                images.append(comp.render().copy())
            with pytest.raises(StopIteration):
                next(source)
        for a, b in zip(held, flows):
            np.testing.assert_array_equal(a, b)
        return flows, images
    plain = run(None)
    ahead = run(FlowConfig(hip_prefetch=3))
    assert len(plain[0]) == len(ahead[0]) == 8
    for a, b in zip(plain[0] + plain[1], ahead[0] + ahead[1]):
        np.testing.assert_array_equal(a, b)


def test_views_of_pooled_arrays_survive_more_frames_than_the_pool_holds():
    """What a caller really keeps is `flow[..., 0]` or `frame[:, :, ::-1]`, not the array: numpy points such a view at the
    hidden owner of the page-locked memory, which the pool watches as well (advisor, round 4) -- over more frames than the
    pools hold (4 + prefetch flows, 4 frames) every held view still shows its own frame's values."""
    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import FlowConfig, LayerConfig
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    h, w = 96, 128
    frames = _bgr_frames(h, w, 13)
    pix = np.random.default_rng(4).integers(0, 256, (h, w, 3), dtype=np.uint8)

    class Src:
        introduction_mask = np.ones((h, w), bool)

        def next(self, timeout=1):
            return pix

    for cfg in (None, FlowConfig(hip_prefetch=2)):
        comp = HipCompositor.from_args(h, w, [LayerConfig(0)])
        comp.set_sources({0: [Src()]})
        views, copies = [], []
        with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward", cv_config=cfg) as source:
            for flow in source:
                comp.update(flow)
                frame = comp.render()
                views.append((flow[..., 0], frame[:, :, ::-1]))        # views only: the arrays themselves are let go
                copies.append((flow[..., 0].copy(), frame[:, :, ::-1].copy()))
                del flow, frame
        assert len(views) == 12
        for (fv, iv), (fc, ic) in zip(views, copies):
            np.testing.assert_array_equal(fv, fc)
            np.testing.assert_array_equal(iv, ic)


@pytest.mark.parametrize("shape,no_overlap", [((48, 48), 0), ((120, 160), 1)])
def test_prefetching_flow_source_with_one_result_set(shape, no_overlap, lib_option):
    """A handle with a single result set in rotation -- a frame under 64 pixels (one scale, K == 0) or option fb_no_overlap
    -- under the prefetching worker, which issues flow t + 1 (its download included) before it ends flow t's: round 4 failed
    on the second flow with 'the previous download of this result set has not been ended' (advisor)."""
    from transflow_amd.config import FlowConfig
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    lib_option("fb_no_overlap", no_overlap)
    h, w = shape
    frames = _bgr_frames(h, w, 7)
    with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward") as source:
        plain = [f.copy() for f in source]
    with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward",
                                 cv_config=FlowConfig(hip_prefetch=2)) as source:
        ahead = [f.copy() for f in source]
    assert len(plain) == len(ahead) == 6
    for a, b in zip(plain, ahead):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("direction", ["backward", "forward"])
def test_device_flows_stay_on_the_device_and_give_the_same_frames(direction):
    """FlowConfig.hip_device_flows: the source yields DeviceFlow objects (transflow_amd/deviceflow.py) -- the seam
    pipeline.py:562-567 crosses with a host array, crossed in HBM.  Flows (brought down on demand) and the compositor's
    frames are those of the plain path bit for bit, with and without the prefetching worker; flows the caller keeps --
    more of them than the source's ring of buffers holds -- stay what they were; a flow modified through itself
    (`flow *= 0`) is taken from the host, as the reference would take it; an ordinary pickle is the host array."""
    import pickle

    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import FlowConfig, LayerConfig
    from transflow_amd.deviceflow import DeviceFlow
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    h, w = 120, 168
    frames = _bgr_frames(h, w, 12)
    pix = np.random.default_rng(8).integers(0, 256, (h, w, 3), dtype=np.uint8)

    class Src:
        introduction_mask = np.ones((h, w), bool)

        def next(self, timeout=1):
            return pix

    def run(cfg, zero_at=None):
        comp = HipCompositor.from_args(h, w, [LayerConfig(0, reset_mode="random", reset_random_factor=0.05)], rng="device")
        comp.set_sources({0: [Src()]})
        kept, images, states = [], [], []
        with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction=direction, cv_config=cfg) as source:
            for t, flow in enumerate(source):
                if cfg is not None and cfg.hip_device_flows:
                    assert isinstance(flow, DeviceFlow) and flow.shape == (h, w, 2) and flow.dtype == np.float32
                if t == zero_at:
                    flow *= 0                     # through the flow itself: the compositor must see zeros
                comp.update(flow)
                if t == 5:                        # looking at the layer between update() and render() (a device flow's
                    states.append(comp.layers[0].data.copy())   # frame waits for render() to run as one launch: it is
                if t == 7:                        # run the ordinary way first) -- and a checkpoint at that moment
                    states.append(pickle.loads(pickle.dumps(comp)).layers[0]._pending_state[0].copy())
                images.append(comp.render().copy())
                kept.append(flow)
        all_states.append(states)
        return kept, images
    all_states = []
    plain_flows, plain_images = run(None)
    for cfg in (FlowConfig(hip_device_flows=True), FlowConfig(hip_device_flows=True, hip_prefetch=2),
                FlowConfig(hip_device_flows=True, hip_prefetch=3, hip_batch=3)):
        flows, images = run(cfg)
        assert len(flows) == len(plain_flows) == 11
        assert not any(f._host is not None for f in flows)           # nothing came down: the compositor read HBM
        for a, b in zip(plain_images, images):
            np.testing.assert_array_equal(a, b)
        for a, b in zip(plain_flows, flows):                         # eleven flows held, a ring of four: all intact
            np.testing.assert_array_equal(a, np.asarray(b))
        back = pickle.loads(pickle.dumps(flows[3]))
        assert type(back) is np.ndarray
        np.testing.assert_array_equal(back, plain_flows[3])
    for states in all_states[1:]:                 # the layer's state as seen between the two calls, and as checkpointed
        for a, b in zip(all_states[0], states):
            np.testing.assert_array_equal(a, b)
    zero_plain = run(None, zero_at=4)[1]
    zero_dev = run(FlowConfig(hip_device_flows=True), zero_at=4)[1]
    for a, b in zip(zero_plain, zero_dev):
        np.testing.assert_array_equal(a, b)
    assert not np.array_equal(zero_plain[6], plain_images[6])         # (the zeroed flow did change the recurrence)


@pytest.mark.parametrize("direction", ["backward", "forward"])
def test_batched_look_ahead_gives_the_flows_of_single_calls(direction):
    """FlowConfig.hip_batch: the source reads n frames ahead and runs their pairs in one Farneback call (a ring of n + 1
    frame slots, the last frame of a call the first of the next), never across the wrap of a repeated input
    (source.py:286-291); flows come out one at a time, post-processed with their own t (a filter of t shows it).  Same
    flows as the one-pair-per-call source, bit for bit: host arrays, with the prefetching worker, on the device."""
    from transflow_amd.config import FlowConfig
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    h, w = 96, 136
    frames = _bgr_frames(h, w, 12, seed=9)

    def run(cfg):
        with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction=direction, cv_config=cfg, repeat=2,
                                     flow_filters="scale=1+t") as source:
            assert source.length == 22
            return [np.array(f) for f in source]
    plain = run(None)
    assert len(plain) == 22 and not np.array_equal(plain[0], plain[11])          # (the second pass has another t)
    for cfg in (FlowConfig(hip_batch=4), FlowConfig(hip_batch=5, hip_prefetch=2),
                FlowConfig(hip_batch=3, hip_prefetch=4, hip_device_flows=True), FlowConfig(hip_batch=16)):
        got = run(cfg)
        assert len(got) == 22
        for a, b in zip(plain, got):
            np.testing.assert_array_equal(a, b)


def test_device_flow_that_leaves_the_frame_raises_at_update_like_a_host_flow():
    """movement.py:33, 39: the reference raises IndexError inside update() when a rounded flow vector leaves the frame.
    A DeviceFlow that no post_process clipped (not marked in_frame) is checked there too -- one synchronisation -- so the
    error comes out of HipCompositor.update, not out of a later render (round 6); the compositor goes on afterwards."""
    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import LayerConfig
    from transflow_amd.device import DevBuffer
    from transflow_amd.deviceflow import DeviceFlow
    h, w = 32, 40
    pix = np.zeros((h, w, 3), np.uint8)
    comp = HipCompositor.from_args(h, w, [LayerConfig(0)])
    comp.set_sources({0: [FakeSource([pix], np.ones((h, w), bool))]})
    bad = np.zeros((h, w, 2), np.float32)
    bad[0, 0] = (-3.0, 0.0)
    buf = DevBuffer.from_array(bad)
    with pytest.raises(IndexError):
        comp.update(DeviceFlow(bad.shape, buf.ptr, None))
    comp.render()
    good = DevBuffer.from_array(np.zeros((h, w, 2), np.float32))
    comp.update(DeviceFlow(bad.shape, good.ptr, None))
    comp.render()
    for cls in ("sum", "introduction"):                   # every layer class takes the same path
        other = HipCompositor.from_args(h, w, [LayerConfig(0, classname=cls)])
        other.set_sources({0: [FakeSource([pix], np.ones((h, w), bool))]})
        if cls == "sum":
            other.update(DeviceFlow(bad.shape, buf.ptr, None))       # sum.py adds floor(flow): nothing to leave
        else:
            with pytest.raises(IndexError):
                other.update(DeviceFlow(bad.shape, buf.ptr, None))
        other.close()


def test_prefetching_flow_source_passes_errors_on_and_stops_cleanly():
    """An exception in the worker thread (a provider that fails) surfaces in the consumer's thread at the flow it belongs
    to; closing a source whose worker is blocked on a full queue returns."""
    from transflow_amd.config import FlowConfig
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    h, w = 64, 96
    frames = _bgr_frames(h, w, 12)

    class Failing(ArrayFrameProvider):
        def read(self):
            if self.pos == 4:
                raise OSError("decoder died")
            return ArrayFrameProvider.read(self)
    with HipFlowSource.from_args(Failing(frames, 25.0), direction="backward", cv_config=FlowConfig(hip_prefetch=2)) as source:
        got = []
        with pytest.raises(OSError, match="decoder died"):
            for flow in source:
                got.append(flow)
        assert len(got) == 3
    source = HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward", cv_config=FlowConfig(hip_prefetch=1)).__enter__()
    next(source)                                  # the worker now sits on a full queue
    source.close()


def test_flow_source_host_path_ingests_on_the_device_too(lib_option):
    """A lock expression makes __next__ take the public next() / post_process() pair (host arrays): the frames still
    become grey on the device, and the flows equal the resident path's."""
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    h, w = 120, 160
    frames = _bgr_frames(h, w, 4)
    with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward") as source:
        resident = [f.copy() for f in source]
    with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward", lock_expr="0", lock_mode="skip") as source:
        assert not source._resident_ok()
        host = [f.copy() for f in source]
    assert len(resident) == len(host) == 3
    for a, b in zip(resident, host):
        np.testing.assert_array_equal(a, b)


def test_flow_config_can_ask_for_opencv_identical_flows(lib_option):
    """`"hip_exact_sums": true` in the cv_config JSON: the drop-in source's flows equal the CPU path's bit for bit.
    Exactness is the HANDLE's (tf_fb_set_exact): the source never touches the process-wide option, whatever it says."""
    from transflow_amd import _lib
    from transflow_amd.config import FlowConfig
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    h, w = 120, 160
    frames = _frames(h, w, 3)
    cfg = FlowConfig(hip_exact_sums=True, fb_levels=2)
    assert cfg.to_dict()["hip_exact_sums"] is True and FlowConfig(**cfg.to_dict()).hip_exact_sums
    expected = [R.post_process(OF.calc(frames[t + 1], frames[t], levels=2), R.BACKWARD) for t in range(2)]
    for process_wide in (0, 1):
        lib_option("fb_exact_sums", process_wide)
        with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward", cv_config=cfg) as source:
            flows = [f.copy() for f in source]
            assert _lib.get_option("fb_exact_sums") == process_wide    # untouched while the source is open ...
        assert _lib.get_option("fb_exact_sums") == process_wide        # ... and after
        for flow, exp in zip(flows, expected):
            np.testing.assert_array_equal(flow, exp)


def test_two_sources_of_one_process_disagree_about_exactness(lib_option):
    """One exact source and one default source open TOGETHER, their calls interleaved, the default one prefetching in a
    worker thread on a library stream of its own: each returns its own mode's flows -- bit-identical to the CPU path's, or
    the default mode's (equal to a default source run alone, within tolerance of the oracle).  The two modes' flows are
    often equal to the last bit, so the kernels say which ran: the exact mode's row walker is launched once per level and
    iteration of the exact source's six flows and not once more (the profiler's records are shared by the two threads).
    With the process-wide option saying the opposite of each."""
    from transflow_amd.config import FlowConfig
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    h, w = 270, 480
    frames = _frames(h, w, 7, seed=11)
    oracle = [R.post_process(OF.calc(frames[t + 1], frames[t]), R.BACKWARD) for t in range(6)]
    with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward") as source:
        alone = [f.copy() for f in source]
    from transflow_amd import _lib
    for process_wide in (1, 0):
        lib_option("fb_exact_sums", process_wide)
        _lib.profile(True)
        exact_src = HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward",
                                            cv_config=FlowConfig(hip_exact_sums=True))
        default_src = HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward",
                                              cv_config=FlowConfig(hip_prefetch=2))
        with exact_src as es, default_src as ds:
            for t in range(6):
                if t % 2:
                    d, e = next(ds).copy(), next(es).copy()
                else:
                    e, d = next(es).copy(), next(ds).copy()
                np.testing.assert_array_equal(e, oracle[t], err_msg=f"exact source, flow {t}")
                np.testing.assert_array_equal(d, alone[t], err_msg=f"default source, flow {t}")
                assert np.abs(d - oracle[t]).max() <= 1e-4 * max(1.0, float(np.abs(oracle[t]).max()))
        _lib.profile(False, reset=False)
        rep = _lib.profile_report()
        walkers = sum(n for name, (n, _) in rep.items() if name.startswith("fb_exact_hsolve"))
        assert walkers == 6 * 4 * 3, rep                    # levels=3: four scales, three iterations, six flows
        assert any(name.startswith(("fb_flow_iter", "fb_blur_solve")) for name in rep), rep   # the default source's


def test_lazy_frames_are_the_frames_of_the_synchronous_path():
    """HipCompositor(..., lazy_frames=True): render() returns a DeviceFrame (transflow_amd/deviceframe.py) whose download
    is under way -- frame t comes down (pipeline.py:518, output/ffmpeg.py:32-54) beside frame t + 1's uploads and
    kernels (pipeline.py:565).  The frames are those of the synchronous path bit for bit, whether they are read at once,
    one frame late (an output thread), or all at the end (more frames held than the compositor has images: the pool's
    arrays are never handed out twice); an ordinary pickle is the host array; a compositor pickled with lazy frames
    comes back with them; multi-layer compositors and host-array flows work the same."""
    import pickle

    from transflow_amd.compositor import HipCompositor
    from transflow_amd.config import FlowConfig, LayerConfig
    from transflow_amd.deviceframe import DeviceFrame
    from transflow_amd.flow import ArrayFrameProvider, HipFlowSource
    h, w = 120, 168
    frames = _bgr_frames(h, w, 10)
    pix = [np.random.default_rng(80 + i).integers(0, 256, (h, w, 3), dtype=np.uint8) for i in range(3)]

    class Src:
        introduction_mask = np.ones((h, w), bool)

        def __init__(self):
            self.n = 0

        def next(self, timeout=1):
            self.n += 1
            return pix[self.n % 3]

    def run(lazy, cfg, read, layers=1):
        comp = HipCompositor.from_args(h, w, [LayerConfig(i, reset_mode="random", reset_random_factor=0.05) for i in range(layers)],
                                       rng="device", lazy_frames=lazy)
        comp.set_sources({i: [Src()] for i in range(layers)})
        out, held = [], []
        with HipFlowSource.from_args(ArrayFrameProvider(frames, 25.0), direction="backward", cv_config=cfg) as source:
            for t, flow in enumerate(source):
                comp.update(flow)
                frame = comp.render()
                if lazy:
                    assert isinstance(frame, DeviceFrame) and frame.shape == (h, w, 3) and frame.dtype == np.uint8
                if read == "at once":
                    out.append(np.array(frame))
                elif read == "one late":
                    if held:
                        out.append(np.array(held.pop()))
                    held.append(frame)
                else:
                    held.append(frame)
                if t == 4 and lazy:
                    again = pickle.loads(pickle.dumps(comp))
                    assert again.lazy_frames and again._comp is None and again._comp2 is None
        out.extend(np.array(f) for f in held)
        if lazy and read == "at the end":
            assert sum(f.arrived for f in held) == len(held)
            back = pickle.loads(pickle.dumps(held[2]))
            assert type(back) is np.ndarray
            np.testing.assert_array_equal(back, out[2])
            held[3][0, 0] = (1, 2, 3)                                    # once down it is an ordinary writable array
            assert tuple(np.asarray(held[3])[0, 0]) == (1, 2, 3) and held[3].tobytes()[:3] == bytes([1, 2, 3])
            assert int((held[5] // 2).max()) <= 127 and held[5].mean() > 0
        comp.close()
        return out

    for cfg, layers in ((None, 1), (FlowConfig(hip_device_flows=True, hip_prefetch=2), 1), (FlowConfig(hip_device_flows=True), 2)):
        plain = run(False, cfg, "at once", layers)
        assert len(plain) == 9
        for read in ("at once", "one late", "at the end"):
            got = run(True, cfg, read, layers)
            assert len(got) == len(plain)
            for t, (a, b) in enumerate(zip(plain, got)):
                np.testing.assert_array_equal(a, b, err_msg=f"{read}, frame {t}, layers {layers}")
