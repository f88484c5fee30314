"""Host-side mirror of the reference interface (no GPU): argument parsing, the Builder's
timing arithmetic, lock logic, masks, config round trips, pickling of a never-used
compositor.  Expected values for masks/colours come from the reference itself
(tests/golden/masks.npz, tools/capture_golden.py)."""
import os
import pickle

import numpy as np
import pytest

from tests.helpers import GOLDEN
from transflow_amd import masks
from transflow_amd.compositor import HipCompositor, HipMoveReferenceLayer
from transflow_amd.config import FlowConfig, LayerConfig, parse_bool_arg
from transflow_amd.flow import ArrayFrameProvider, FlowSource, HipFlowSource


def test_masks_and_colors_match_reference():
    z = np.load(os.path.join(GOLDEN, "masks.npz"))
    for shape in [(37, 53), (60, 80)]:
        for i, spec in enumerate(z["specs"]):
            f = masks.load_float_mask(str(spec), shape, 1)
            b = masks.load_bool_mask(str(spec), shape, True)
            np.testing.assert_array_equal(f, z[f"f_{shape[0]}_{i}"], err_msg=str(spec))
            assert f.dtype == z[f"f_{shape[0]}_{i}"].dtype, spec
            np.testing.assert_array_equal(b, z[f"b_{shape[0]}_{i}"], err_msg=str(spec))
    np.testing.assert_array_equal(masks.load_float_mask(None, (4, 5), 1), z["none_float"])
    np.testing.assert_array_equal(masks.load_bool_mask(None, (4, 5), True), z["none_bool"])
    for c, v in zip(z["colors"], z["color_values"]):
        assert masks.parse_color(str(c)) == tuple(int(x) for x in v), c
    with pytest.raises(ValueError):
        masks.load_float_mask("border:1:2:3", (8, 8))


def test_layer_config_defaults_and_roundtrip():
    c = LayerConfig(0)
    assert c.classname == "moveref" and c.reset_mode == "off"
    assert (c.transparent_pixels_can_move, c.pixels_can_move_to_empty_spot, c.pixels_can_move_to_filled_spot,
            c.moving_pixels_leave_empty_spot) == (False, True, True, False)
    assert c.reset_random_factor == 1 and c.reset_constant_step == 1 and c.reset_linear_factor == 0.1
    c = LayerConfig(2, transparent_pixels_can_move="yes", pixels_can_move_to_empty_spot="off", reset_mode="random",
                    reset_random_factor=0.25, mask_src="border:2")
    assert c.transparent_pixels_can_move is True and c.pixels_can_move_to_empty_spot is False
    d = c.todict()
    c2 = LayerConfig.fromdict(d)
    assert c2.todict() == d
    assert LayerConfig.fromdict({"index": 1}).classname == "reference"   # reference config.py:110
    assert parse_bool_arg("OUI", False) and not parse_bool_arg("no", True) and parse_bool_arg(None, True)


def test_flow_config(tmp_path):
    c = FlowConfig()
    assert c.fb_kwargs() == dict(pyr_scale=0.5, levels=3, winsize=15, iterations=3, poly_n=5, poly_sigma=1.2, flags=0)
    c = FlowConfig(fb_levels=5, hs_alpha=2.0)
    p = tmp_path / "cfg.json"
    c.to_file(str(p))
    c2 = FlowConfig.from_file(str(p))
    assert c2.fb_levels == 5 and c2.extra["hs_alpha"] == 2.0
    with pytest.raises(ValueError):
        FlowConfig(method="horn-schunck")


def _builder(n_frames=51, fps=25.0, **kw):
    frames = [np.zeros((4, 6), np.uint8)] * n_frames
    return HipFlowSource.Builder(ArrayFrameProvider(frames, fps), None, **kw)


def test_builder_timing_arithmetic(tmp_path):
    """FlowSource.Builder.build, reference source.py:151-197."""
    b = _builder()
    b.build()
    assert (b.width, b.height, b.framerate, b.base_length) == (6, 4, 25.0, 50)
    assert (b.start_frame, b.end_frame, b.length, b.ckpt_start_frame) == (0, 50, 50, 0)
    b = _builder(seek_time=1.0, duration_time=0.5, repeat=3)
    b.build()
    assert (b.start_frame, b.end_frame, b.length) == (25, 37, 36)
    b = _builder(duration_time=10.0)
    b.build()
    assert b.end_frame == 50 and b.length == 50                      # clipped to the base length
    b = _builder(repeat=0)
    b.build()
    assert b.length is None
    b = _builder(seek_time=0.4, seek_ckpt=47)
    b.build()
    assert b.start_frame == 10 and b.ckpt_start_frame == 10 + 47 % 40
    b = _builder(lock_expr="(0.2,0.4),(1,0.2)", lock_mode="stay")
    b.build()
    assert b.lock_expr_stay == ((0.2, 0.4), (1, 0.2)) and b.length == 50 + 10 + 5
    b = _builder(lock_expr="t > 1", lock_mode="skip")
    b.build()
    assert b.lock_expr_skip(1.5) and not b.lock_expr_skip(0.5)
    kpath = os.path.join(str(tmp_path), "k.npy")
    np.save(kpath, np.full((3, 3), 1 / 9.0))
    b = _builder(kernel_path=kpath)
    b.build()
    assert b.kernel.shape == (3, 3) and b.kwargs()["kernel"] is b.kernel          # source.py:131-132
    b = _builder(flow_filters="scale=2*t; threshold = 0.5")
    b.build()
    assert [f.name for f in b.flow_filters] == ["scale", "threshold"] and b.flow_filters[0].expr(1.5) == 3.0
    b = _builder(flow_filters="polar=r*2:a+t")
    b.build()
    assert b.flow_filters[0].name == "polar" and not b.flow_filters[0].polar.radius.scalar_only
    with pytest.raises(NotImplementedError):
        _builder(flow_filters="polar=r.cumsum():a").build()              # not expressible per pixel
    with pytest.raises(ValueError):
        _builder(flow_filters="polar=r").build()                          # filters.py:29-30
    with pytest.raises(ValueError):
        _builder(flow_filters="blur=3").build()


def test_direction_and_lockmode_parsing():
    D, L = FlowSource.Direction, FlowSource.LockMode
    assert D.from_arg(None) is D.FORWARD and D.from_arg("backward") is D.BACKWARD and D.from_arg(1) is D.BACKWARD
    assert L.from_arg(None) is L.STAY and L.from_arg("skip") is L.SKIP and L.from_arg(0) is L.STAY
    with pytest.raises(ValueError):
        D.from_arg("sideways")
    assert FlowSource.Builder().direction is D.BACKWARD              # Builder default, source.py:61


class _Scripted(FlowSource):
    """Counts next() calls; flows are tagged with their index."""

    def __init__(self, *a, **k):
        self.calls = 0
        super().__init__(*a, **k)

    def next(self):
        self.calls += 1
        return np.full((2, 3, 2), float(self.calls), np.float32)

    def post_process(self, raw):   # keep this test on the CPU
        return raw


def test_iteration_lock_stay_and_skip_and_rewind():
    D, L = FlowSource.Direction, FlowSource.LockMode
    s = _Scripted(D.BACKWARD, 3, 2, 10.0, 6, 0, 0, 4)
    vals = [float(f[0, 0, 0]) for f in s]
    assert vals == [1, 2, 3, 4, 5, 6] and s.calls == 6               # rewinds at end_frame, keeps counting
    # STAY: from t=0.2 hold the flow for 0.2 s (frames 2,3 repeat frame 1's flow)
    s = _Scripted(D.BACKWARD, 3, 2, 10.0, 6, 0, 0, 100, lock_mode=L.STAY, lock_expr_stay=((0.2, 0.2),))
    vals = [float(next(s)[0, 0, 0]) for _ in range(4)]
    assert vals == [1, 2, 2, 2]
    with pytest.raises(IndexError):
        next(s)   # the reference indexes past its last (start, duration) pair here too (source.py:304-307)
    # SKIP: while locked the previous flow is reused AND one input flow is consumed
    s = _Scripted(D.BACKWARD, 3, 2, 10.0, None, 0, 0, 100, lock_mode=L.SKIP, lock_expr_skip=lambda t: 0.15 < t < 0.35)
    vals = [float(next(s)[0, 0, 0]) for _ in range(5)]
    assert vals == [1, 2, 2, 2, 5] and s.calls == 5
    s = _Scripted(D.BACKWARD, 3, 2, 10.0, 2, 0, 0, 100, lock_mode=L.STAY, lock_expr_stay=((0.0, 1.0),))
    with pytest.raises(RuntimeError):
        next(s)                                                      # locked before any flow exists


def test_hip_flow_source_plumbing_without_gpu():
    frames = [np.full((4, 6), i, np.uint8) for i in range(5)]
    with pytest.raises(StopIteration):
        # provider exhausted inside next(): StopIteration like cv.py:462-463
        b = HipFlowSource.Builder(ArrayFrameProvider(frames[:1], 10.0), None, direction="forward")
        b.build()
        src = HipFlowSource(*b.args(), **b.kwargs())
        src.next()
    b = HipFlowSource.Builder(ArrayFrameProvider(frames, 10.0), None, seek_time=0.2)
    b.build()
    src = HipFlowSource(*b.args(), **b.kwargs())
    src.validate()
    assert src.prev_gray[0, 0] == 2 and src.length == 2              # rewind decoded up to the start frame
    # a provider may report another size than its frames have (cv.py:420-427 vs :461): the reported one rules
    p = ArrayFrameProvider([np.zeros((8, 12, 3), np.uint8)] * 3, 10.0, size=(6, 4))
    b = HipFlowSource.Builder(p, None)
    b.build()
    assert (b.width, b.height) == (6, 4)


def test_compositor_surface_and_pickle_without_gpu():
    comp = HipCompositor.from_args(4, 6, [LayerConfig(0, reset_mode="random", mask_src="border:1")], "#102030")
    assert comp.background_color == (16, 32, 48) and comp.background.shape == (4, 6, 3)
    layer = comp.layers[0]
    assert isinstance(layer, HipMoveReferenceLayer)
    assert (layer.INDEX_I, layer.INDEX_J, layer.INDEX_ALPHA, layer.INDEX_SOURCE, layer.DEPTH) == (0, 1, 2, 3, 4)
    assert layer.mask_src.dtype == bool and layer.mask_src.sum() == 4 * 6 - 2 * 4

    class Src:
        introduction_mask = np.ones((4, 6), bool)
    comp.set_sources({0: [Src()]})
    assert len(layer.sources) == 1
    blob = pickle.dumps(comp)                                         # never touched the GPU
    back = pickle.loads(blob)
    assert back.layers[0].sources == [] and back.layers[0].config.reset_mode == "random"
    from transflow_amd.compositor import HipIntroductionLayer, HipStaticLayer, HipSumLayer
    others = HipCompositor.from_args(4, 6, [LayerConfig(0, classname="sum"), LayerConfig(1, classname="static"),
                                            LayerConfig(2, classname="introduction")])
    assert [type(x) for x in others.layers] == [HipSumLayer, HipStaticLayer, HipIntroductionLayer]
    intro = others.layers[2]
    assert (intro.INDEX_I, intro.INDEX_J, intro.INDEX_ALPHA, intro.INDEX_SOURCE, intro.DEPTH) == (5, 6, 3, 4, 8)
    assert not hasattr(others.layers[0], "mask_src") and not hasattr(others.layers[1], "reset_mask")
    with pytest.raises(ValueError):
        HipCompositor.from_args(4, 6, [LayerConfig(0, classname="reference")])     # layer.py:56
    with pytest.raises(ValueError):
        HipCompositor.from_args(4, 6, [LayerConfig(0, reset_mode="never")])


@pytest.mark.skipif(not os.path.isdir("/root/reference/transflow"), reason="reference tree not present")
def test_dropin_install_patches_reference_factories():
    """transflow_amd.dropin against the importable half of the real reference package."""
    import sys
    sys.path.insert(0, "/root/reference")
    try:
        from transflow.compositor import Compositor
        from transflow.compositor.layers.move_reference import MoveReferenceLayer
        from transflow.config import LayerConfig as RefLayerConfig
        from transflow.flow.sources.source import FlowSource as RefFlowSource

        from transflow_amd import dropin
        dropin.install()
        try:
            comp = Compositor.from_args(4, 6, [RefLayerConfig(0, reset_mode="random", reset_random_factor=0.5)],
                                        background_color="#ff8000")
            assert isinstance(comp, HipCompositor) and comp.layers[0].config.reset_random_factor == 0.5
            from transflow.compositor.layers.data import DataLayer as RefDataLayer
            assert isinstance(comp.layers[0], RefDataLayer)          # what extra/control.py:155 checks
            assert comp.background_color == (255, 128, 0)
            other = Compositor.from_args(4, 6, [RefLayerConfig(0, classname="sum"),
                                                RefLayerConfig(1, classname="introduction")])
            assert isinstance(other, HipCompositor)                  # every layer class is served
            b = RefFlowSource.from_args("clip.mp4", direction=RefFlowSource.Direction.BACKWARD)
            assert isinstance(b, HipFlowSource.Builder) and b.direction is FlowSource.Direction.BACKWARD
            with pytest.raises(Exception):
                RefFlowSource.from_args("clip.flow.zip").build()     # falls through to the reference's archive source
        finally:
            dropin.uninstall()
        assert isinstance(Compositor.from_args(2, 3, [RefLayerConfig(0)]).layers[0], MoveReferenceLayer)
    finally:
        sys.path.remove("/root/reference")
        for m in [m for m in sys.modules if m == "transflow" or m.startswith("transflow.")]:
            del sys.modules[m]


@pytest.mark.skipif(not os.path.isdir("/root/reference/transflow"), reason="reference tree not present")
def test_flow_archive_format_is_the_references(tmp_path):
    """.flow.zip written here is read by the reference's ArchiveFlowSource and vice versa
    (output/zip.py, output/numpy.py, flow/sources/archive.py)."""
    import sys
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from transflow.flow.sources.archive import ArchiveFlowSource as RefArchive
    from transflow.output.numpy import NumpyOutput as RefNumpyOutput
    from transflow.utils import find_unique_path as ref_unique_path
    from transflow_amd.archive import ArchiveFlowSource, NumpyOutput, flow_export_meta, unique_path
    rng = np.random.default_rng(3)
    flows = [rng.normal(0, 2, (6, 9, 2)).astype(np.float32) for _ in range(3)]
    ours, theirs = str(tmp_path / "a.flow.zip"), str(tmp_path / "b.flow.zip")
    for cls, path in ((NumpyOutput, ours), (RefNumpyOutput, theirs)):
        out = cls(path, True)
        out.write_meta(flow_export_meta("clip.mp4", 9, 6, 25.0, "backward"))
        for f in flows:
            out.write_array(f)
        out.close()
    # byte-identical members
    import zipfile
    with zipfile.ZipFile(ours) as a, zipfile.ZipFile(theirs) as b:
        assert a.namelist() == b.namelist() == ["meta.json", "000000000.npy", "000000001.npy", "000000002.npy"]
        for name in a.namelist():
            assert a.read(name) == b.read(name), name
    # cross reading: builders agree on everything they derive
    rb, ob = RefArchive.Builder(ours), ArchiveFlowSource.Builder(theirs)
    rb.build()
    ob.build()
    for attr in ("width", "height", "framerate", "base_length", "length", "start_frame", "end_frame"):
        assert getattr(rb, attr) == getattr(ob, attr), attr
    assert rb.direction.value == ob.direction.value == 1
    src = ArchiveFlowSource(*ob.args(), **ob.kwargs())
    src.validate()
    for t in range(3):
        np.testing.assert_array_equal(src.next(), flows[t])
        src.input_frame_index += 1
    with pytest.raises(KeyError):
        src.next()                                   # how an archive ends in the reference
    src.archive.close()
    rb.archive.close()
    assert unique_path(ours) == str(tmp_path / "a.000.flow.zip")
    # the naming of a taken path agrees with the reference's on every shape of name
    for name in ("a.flow.zip", "clip.mp4", "x.000.flow.zip", "x.004.map.zip", "noext", "b.7.txt", "c.123", "d.tar.gz"):
        (tmp_path / "names").mkdir(exist_ok=True)
        target = tmp_path / "names" / name
        assert unique_path(str(target)) == ref_unique_path(str(target)) == str(target)      # free: unchanged
        target.write_bytes(b"")
        assert unique_path(str(target)) == ref_unique_path(str(target)), name
        (tmp_path / "names" / os.path.basename(unique_path(str(target)))).write_bytes(b"")
        assert unique_path(str(target)) == ref_unique_path(str(target)), name


@pytest.mark.skipif(not os.path.isdir("/root/reference/transflow"), reason="needs the reference tree (build container only)")
def test_iteration_logic_matches_the_reference_class_on_random_schedules():
    """Differential check of FlowSource.__next__ (locks, rewinds, lengths: source.py:286-332) against
    the reference's own class driven by the same scripted next(): random frame ranges, lengths, STAY
    schedules and SKIP predicates; same values, same input-frame bookkeeping, same exceptions."""
    import sys
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from transflow.flow.sources.source import FlowSource as Ref

    def scripted(base):
        class S(base):
            def __init__(self, *a, **k):
                self.calls, self.rewinds = 0, 0
                base.__init__(self, *a, **k)

            def next(self):
                self.calls += 1
                return np.full((2, 3, 2), float(self.calls), np.float32)

            def rewind(self):
                self.rewinds += 1
                base.rewind(self)

            def post_process(self, raw):
                return raw
        return S

    Mine, Theirs = scripted(FlowSource), scripted(Ref)
    rng = np.random.default_rng(31337)
    for trial in range(300):
        fps = float(rng.choice([1.0, 10.0, 24.0, 29.97]))
        start = int(rng.integers(0, 4))
        end = start + int(rng.integers(1, 8))
        ckpt = int(rng.integers(start, end + 1))
        length = None if rng.random() < 0.2 else int(rng.integers(0, 25))
        mode = int(rng.integers(3))
        kw_m, kw_t = {}, {}
        if mode == 1:
            pairs, t0 = [], 0.0
            for _ in range(int(rng.integers(1, 4))):
                t0 += float(rng.uniform(0, 6 / fps))
                d = float(rng.uniform(0, 5 / fps))
                pairs.append((t0, d))
                t0 += d
            kw_m = dict(lock_mode=FlowSource.LockMode.STAY, lock_expr_stay=tuple(pairs))
            kw_t = dict(lock_mode=Ref.LockMode.STAY, lock_expr_stay=tuple(pairs))
        elif mode == 2:
            a, b = sorted(float(v) for v in rng.uniform(0, 20 / fps, 2))
            pred = (lambda t, a=a, b=b: a < t < b)
            kw_m = dict(lock_mode=FlowSource.LockMode.SKIP, lock_expr_skip=pred)
            kw_t = dict(lock_mode=Ref.LockMode.SKIP, lock_expr_skip=pred)
        m = Mine(FlowSource.Direction.BACKWARD, 3, 2, fps, length, start, ckpt, end, **kw_m)
        r = Theirs(Ref.Direction.BACKWARD, 3, 2, fps, length, start, ckpt, end, **kw_t)
        for step in range(30):
            out = []
            for obj in (m, r):
                try:
                    out.append(("value", float(next(obj)[0, 0, 0])))
                except (StopIteration, IndexError, RuntimeError) as e:
                    out.append((type(e).__name__, None))
            ctx = f"trial {trial} step {step}: fps={fps} range=({start},{ckpt},{end}) length={length} mode={mode}"
            assert out[0] == out[1], ctx
            assert (m.calls, m.rewinds, m.input_frame_index, m.output_frame_index) == \
                   (r.calls, r.rewinds, r.input_frame_index, r.output_frame_index), ctx
            if out[0][0] != "value":
                break


@pytest.mark.skipif(not os.path.isdir("/root/reference/transflow"), reason="needs the reference tree (build container only)")
def test_builder_arithmetic_matches_the_reference_class_on_random_arguments():
    """Differential check of FlowSource.Builder.build (source.py:125-197): random seek / duration /
    repeat / checkpoint / lock arguments over random base lengths and frame rates, streams included."""
    import sys
    import warnings
    if "/root/reference" not in sys.path:
        sys.path.insert(0, "/root/reference")
    from transflow.flow.sources.source import FlowSource as Ref
    rng = np.random.default_rng(8086)
    fields = ("start_frame", "end_frame", "length", "ckpt_start_frame", "is_stream", "repeat", "seek_time",
              "base_length", "lock_expr_stay")
    for trial in range(400):
        kw = dict(direction=str(rng.choice(["forward", "backward"])),
                  seek_time=None if rng.random() < 0.4 else float(rng.choice([0.0, 0.25, 1.0, 2.5, 3.3333])),
                  duration_time=None if rng.random() < 0.4 else float(rng.choice([0.1, 0.5, 1.0, 7.0])),
                  repeat=int(rng.choice([0, 1, 1, 2, 5])),
                  seek_ckpt=None if rng.random() < 0.6 else int(rng.integers(0, 200)))
        r = rng.random()
        if r < 0.3:
            kw.update(lock_mode="stay", lock_expr=str(rng.choice(["0.5,0.2", "(0.2,0.4),(1,0.2)", "(0,1)"])))
        elif r < 0.5:
            kw.update(lock_mode="skip", lock_expr="t > 1")
        base_length = int(rng.choice([-1, 0, 1, 7, 50, 300]))
        fps = float(rng.choice([10.0, 25.0, 29.97, 60.0]))
        out = []
        for cls in (FlowSource, Ref):
            b = cls.Builder(**kw)
            b.width, b.height, b.framerate, b.base_length = 6, 4, fps, base_length
            try:
                with warnings.catch_warnings():
                    warnings.simplefilter("ignore")
                    b.build()
                out.append(tuple(getattr(b, f) for f in fields))
            except Exception as e:                      # e.g. a stream with a duration-less range: both must agree
                out.append(type(e).__name__)
        assert out[0] == out[1], f"trial {trial}: {kw} base_length={base_length} fps={fps}: {out}"


@pytest.mark.skipif(not os.path.isdir("/root/reference/transflow"), reason="needs the reference tree (build container only)")
def test_checkpointed_layers_pass_controls_datalayer_test(tmp_path):
    """extra/control.py:146-162 unpickles a checkpoint's compositor, insists on isinstance(layer, DataLayer) and reads
    layer.data / layer.INDEX_*: in a fresh process that has transflow on its path, a pickled HipCompositor's data
    layers are DataLayer instances, the static layer is not (as in the reference), and no GPU is touched."""
    import subprocess
    import sys
    comp = HipCompositor.from_args(6, 8, [LayerConfig(0), LayerConfig(1, classname="sum"),
                                          LayerConfig(2, classname="introduction"), LayerConfig(3, classname="static")])
    rng = np.random.default_rng(4)
    for layer, depth in zip(comp.layers[:3], (4, 4, 8)):
        data = rng.integers(0, 6, (6, 8, depth), dtype=np.int32)
        state = layer.__getstate__()
        state["_saved_state"] = (data, np.zeros((6, 8, 4), np.uint8))
        layer.__setstate__(state)
    blob = tmp_path / "compositor.bin"
    blob.write_bytes(pickle.dumps(comp))
    code = f"""
import pickle, sys, numpy
sys.path.insert(0, "/root/reference"); sys.path.insert(0, {str(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))!r})
from transflow.compositor.layers.data import DataLayer          # what extra/control.py imports
comp = pickle.load(open({str(blob)!r}, "rb"))
flags = [isinstance(l, DataLayer) for l in comp.layers]
assert flags == [True, True, True, False], flags
layer = comp.layers[0]
mapping = numpy.concatenate([layer.data[:, :, layer.INDEX_J][:, :, numpy.newaxis],
                             layer.data[:, :, layer.INDEX_I][:, :, numpy.newaxis]], axis=2).astype(int)   # control.py:158-160
alpha = layer.data[:, :, layer.INDEX_ALPHA]
intro = comp.layers[2]
assert (intro.INDEX_I, intro.INDEX_J, intro.INDEX_ALPHA) == (5, 6, 3) and intro.data.shape == (6, 8, 8)
import transflow_amd._lib as L
assert L._lib is None, "the GPU library was loaded"
print("OK", mapping.shape, int(alpha.sum()))
"""
    env = dict(os.environ, PYTHONDONTWRITEBYTECODE="1")
    out = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert out.returncode == 0, out.stderr[-2000:]
    assert out.stdout.startswith("OK (6, 8, 2)")
