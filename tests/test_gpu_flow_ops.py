"""GPU parity of the flow-array steps either side of the path against the
reference's own vectors (tests/golden/flow_ops.npz) and, at full size, against the oracle."""
import os

import numpy as np
import pytest

from oracle import flow_ops_ref as F
from oracle import frames_ref
from tests.helpers import GOLDEN

pytestmark = pytest.mark.gpu
Z = np.load(os.path.join(GOLDEN, "flow_ops.npz"))


def test_merge_golden_gpu():
    from transflow_amd.flowops import merge_flows
    for i in range(int(Z["merge_cases"])):
        kind, n = str(Z[f"merge_{i}_kind"]), int(Z[f"merge_{i}_n"])
        out = merge_flows(kind, [Z[f"merge_{i}_in{j}"] for j in range(n)])
        np.testing.assert_array_equal(out, Z[f"merge_{i}_out"], err_msg=f"{kind} n={n}")
    with pytest.raises(ValueError):
        merge_flows("absmax", [Z["merge_0_in0"]] * 3)
    with pytest.raises(KeyError):
        merge_flows("median", [Z["merge_0_in0"]])


def test_merge_full_size_vs_oracle():
    from transflow_amd.flowops import MERGE_KINDS, merge_flows
    rng = np.random.default_rng(5)
    flows = [rng.normal(0, 2, (1080, 1920, 2)).astype(np.float32) for _ in range(3)]
    flows[1][rng.random(flows[1].shape) < 0.3] = 0
    flows[2][0, 0] = np.nan
    for kind in MERGE_KINDS:
        use = flows[:2] if kind == "absmax" else flows
        np.testing.assert_array_equal(merge_flows(kind, use), F.merge(kind, use), err_msg=kind)


def test_upscale_golden_gpu():
    from transflow_amd.flowops import upscale_array
    for i in range(int(Z["up_cases"])):
        wf, hf = (int(v) for v in Z[f"up_{i}_f"])
        np.testing.assert_array_equal(upscale_array(Z[f"up_{i}_in"], wf, hf), Z[f"up_{i}_out"])
    a = np.random.default_rng(6).normal(0, 3, (540, 960, 2)).astype(np.float32)
    np.testing.assert_array_equal(upscale_array(a, 2, 2), F.upscale(a, 2, 2))


def test_kernel_post_process_golden_gpu():
    from transflow_amd.flowops import convolve_post_process
    for i in range(int(Z["conv_cases"])):
        k, exp = Z[f"conv_{i}_kernel"], Z[f"conv_{i}_out"]
        out = convolve_post_process(Z[f"conv_{i}_in"], k, int(Z[f"conv_{i}_dir"]))
        assert out.dtype == exp.dtype, (i, k.dtype)
        np.testing.assert_array_equal(out, exp, err_msg=f"case {i} kernel {k.dtype}{k.shape}")


def test_kernel_through_the_flow_source():
    """HipFlowSource.post_process with mask + filter + kernel, against the oracle's chain."""
    from oracle import remap_ref as R
    from transflow_amd.flow import FlowFilter, FlowSource
    h, w = 45, 64
    rng = np.random.default_rng(8)
    kernel = rng.normal(0, 0.3, (3, 5))
    mask = rng.random((h, w, 1)).astype(np.float32)
    for direction in ("forward", "backward"):
        src = FlowSource(direction, w, h, 25.0, 10, 0, 0, 10, mask=mask, kernel=kernel,
                         flow_filters=[FlowFilter.from_string("scale=1.5")])
        raw = rng.normal(0, 3, (h, w, 2)).astype(np.float32)
        out = src.post_process(raw.copy())
        pre = R.pre_steps(raw.copy(), [("scale", 1.5)], mask)
        exp = F.post_process_with_kernel(pre, kernel, R.FORWARD if direction == "forward" else R.BACKWARD)
        assert out.dtype == np.float64
        np.testing.assert_array_equal(out, exp)
        src.close()


def test_render_golden_gpu():
    from transflow_amd.flowops import render1d, render2d
    for i in range(int(Z["r1_cases"])):
        out = render1d(Z[f"r1_{i}_in"], float(Z[f"r1_{i}_scale"]), tuple(str(c) for c in Z[f"r1_{i}_colors"]),
                       bool(Z[f"r1_{i}_binary"]))
        np.testing.assert_array_equal(out, Z[f"r1_{i}_out"], err_msg=f"render1d {i}")
    for i in range(int(Z["r2_cases"])):
        out = render2d(Z[f"r2_{i}_in"], float(Z[f"r2_{i}_scale"]), tuple(str(c) for c in Z[f"r2_{i}_colors"]))
        np.testing.assert_array_equal(out, Z[f"r2_{i}_out"], err_msg=f"render2d {i}")
    big = np.random.default_rng(9).normal(0, 5, (1080, 1920, 2)).astype(np.float32)
    np.testing.assert_array_equal(render2d(big, 0.1), F.render2d(big, 0.1))
    np.testing.assert_array_equal(render1d(np.abs(big[:, :, 0]), 0.3), F.render1d(np.abs(big[:, :, 0]), 0.3))


@pytest.mark.parametrize("size", [None, (854, 480), (333, 77), (3840, 2160)])
def test_bgr_to_grey_vs_oracle(size):
    from transflow_amd.flowops import bgr_to_grey
    frame = np.random.default_rng(10).integers(0, 256, (1080, 1920, 3), dtype=np.uint8)
    np.testing.assert_array_equal(bgr_to_grey(frame, size), frames_ref.bgr_to_grey(frame, size))


@pytest.mark.parametrize("rounded", [False, True])
def test_archive_source_post_processes_on_the_gpu(tmp_path, rounded):
    """A .flow.zip read back through ArchiveFlowSource: every frame equals the oracle's post_process of
    the stored array (dtype kept for a rounded archive, pipeline.py:506); the archive ends with the
    KeyError of the first missing frame, as in the reference (archive.py:45-48)."""
    from oracle import remap_ref as R
    from transflow_amd.archive import NumpyOutput, flow_export_meta
    from transflow_amd.flow import HipFlowSource
    rng = np.random.default_rng(11)
    h, w = 40, 56
    flows = [rng.normal(0, 4, (h, w, 2)).astype(np.float32) for _ in range(3)]
    if rounded:
        flows = [np.round(f).astype(int) for f in flows]
    path = str(tmp_path / "clip.flow.zip")
    out = NumpyOutput(path, True)
    out.write_meta(flow_export_meta("clip.mp4", w, h, 25.0, "forward"))
    for f in flows:
        out.write_array(f)
    out.close()
    got = []
    with HipFlowSource.from_args(path) as source:
        assert (source.width, source.height, source.framerate, source.length) == (w, h, 25.0, None)
        with pytest.raises(KeyError):
            for flow in source:
                got.append(flow)
    assert len(got) == 3
    for f, g in zip(flows, got):
        exp = R.post_process(f.astype(np.float32), R.FORWARD)
        assert g.dtype == f.dtype
        np.testing.assert_array_equal(g, exp.astype(f.dtype))


@pytest.mark.parametrize("shape,kshape", [((64, 80), (3, 3)), ((45, 150), (9, 4)), ((24, 31), (70, 70)),
                                           ((130, 67), (1, 33)), ((17, 64), (16, 1))])
@pytest.mark.parametrize("dtype", [np.float64, np.float32])
def test_convolution_tiled_and_fallback_vs_oracle(shape, kshape, dtype):
    """Both convolution kernels (LDS-tiled; per-pixel fallback for kernels whose tile exceeds 64 KB)
    against the oracle's scipy-ordered sum, bit for bit, on tile-edge-unfriendly sizes."""
    from transflow_amd.flowops import convolve_post_process
    rng = np.random.default_rng(12)
    flow = rng.normal(0, 3, (*shape, 2)).astype(np.float32)
    kernel = rng.normal(0, 0.3, kshape).astype(dtype)
    out = convolve_post_process(flow, kernel, None)
    exp = np.stack([F.convolve_same(flow[:, :, 0], kernel), F.convolve_same(flow[:, :, 1], kernel)], axis=-1)
    assert out.dtype == exp.dtype == dtype
    np.testing.assert_array_equal(out, exp)


ZP = np.load(os.path.join(GOLDEN, "flow_polar.npz"))


def _close(out, exp):
    """Device sin/cos/atan2/sqrt/pow differ from numpy's by a few ulp; the flow tolerance of the path
    is 1e-4 relative (BASELINE north_star): ask for 20x better."""
    scale = max(1.0, float(np.nanmax(np.abs(exp))))
    assert np.isnan(out).sum() == np.isnan(exp).sum()
    np.testing.assert_allclose(np.nan_to_num(out), np.nan_to_num(exp), rtol=0, atol=5e-6 * scale)


@pytest.mark.parametrize("i", range(int(ZP["cases"])))
def test_polar_filter_golden_gpu(i):
    from transflow_amd.exprs import PolarFilter
    from transflow_amd.flowops import polar_filter
    flow = ZP[f"in_{i}"].copy()
    out = polar_filter(flow, PolarFilter(str(ZP[f"er_{i}"]), str(ZP[f"ea_{i}"])), float(ZP["t"]))
    assert out is flow
    _close(out, ZP[f"out_{i}"])


def test_polar_inside_post_process_chain():
    """scale -> polar -> clip -> BACKWARD clip, as the reference chained them (source.py:339-341)."""
    from transflow_amd.flow import FlowFilter, FlowSource
    src = FlowSource("backward", 34, 21, 30.0, 100, 0, 0, 100,
                     flow_filters=[FlowFilter.from_string("scale=1.5"), FlowFilter.from_string("polar=r+1:a*2"),
                                   FlowFilter.from_string("clip=4")])
    src.output_frame_index = 21
    out = src.post_process(ZP["chain_in"].copy())
    _close(out, ZP["chain_out"])
    src.close()


def test_polar_full_size_vs_oracle():
    from transflow_amd.exprs import PolarFilter
    from transflow_amd.flowops import polar_filter
    flow = np.random.default_rng(13).normal(0, 3, (1080, 1920, 2)).astype(np.float32)
    exp = F.polar(flow.copy(), "numpy.sqrt(r)*(1+t)", "a+numpy.sin(r)", 0.25)
    _close(polar_filter(flow, PolarFilter("numpy.sqrt(r)*(1+t)", "a+numpy.sin(r)"), 0.25), exp)


@pytest.mark.parametrize("shape", [(1, 1), (1, 7), (9, 1), (3, 5), (0, 4)])
def test_tiny_and_empty_inputs(shape):
    """One pixel, one row, one column, and the empty frame through every flow-array entry point."""
    from transflow_amd.exprs import PolarFilter
    from transflow_amd.flowops import (bgr_to_grey, convolve_post_process, merge_flows, polar_filter, render1d,
                                       render2d, upscale_array)
    h, w = shape
    rng = np.random.default_rng(14)
    f0 = rng.normal(0, 2, (h, w, 2)).astype(np.float32)
    f1 = rng.normal(0, 2, (h, w, 2)).astype(np.float32)
    np.testing.assert_array_equal(merge_flows("difference", [f0, f1]), F.merge("difference", [f0, f1]))
    np.testing.assert_array_equal(merge_flows("absmax", [f0, f1]), F.merge("absmax", [f0, f1]))
    np.testing.assert_array_equal(upscale_array(f0, 3, 2), F.upscale(f0, 3, 2))
    np.testing.assert_array_equal(render2d(f0, 0.4), F.render2d(f0, 0.4))
    np.testing.assert_array_equal(render1d(np.abs(f0[:, :, 0]), 0.4), F.render1d(np.abs(f0[:, :, 0]), 0.4))
    if h * w:
        k = rng.normal(0, 0.4, (3, 2))
        for d in (0, 1):
            np.testing.assert_array_equal(convolve_post_process(f0, k, d), F.post_process_with_kernel(f0, k, d))
        exp = F.polar(f0.copy(), "r+1", "a*2", 0.0)
        np.testing.assert_allclose(polar_filter(f0.copy(), PolarFilter("r+1", "a*2"), 0.0), exp, rtol=0, atol=2e-5)
        frame = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        np.testing.assert_array_equal(bgr_to_grey(frame), frames_ref.bgr_to_grey(frame))
        np.testing.assert_array_equal(bgr_to_grey(frame, (2 * w + 1, h + 2)), frames_ref.bgr_to_grey(frame, (2 * w + 1, h + 2)))


@pytest.mark.parametrize("cls", ["sum", "static", "introduction"])
@pytest.mark.parametrize("shape", [(1, 1), (1, 6), (5, 1)])
def test_layer_classes_on_degenerate_frames(cls, shape):
    """The other layer classes on one-pixel / one-row / one-column frames against the oracle."""
    from oracle import remap_ref as R
    from transflow_amd.remap import CompImage, RemapLayer
    h, w = shape
    rng = np.random.default_rng(15)
    intro = [rng.random((h, w)) < 0.8]
    layer = RemapLayer(h, w, layer_class=cls)
    layer.set_sources(intro)
    comp = CompImage(h, w, (9, 8, 7))
    if cls == "sum":
        ref = R.SumLayer(h, w, introduction_masks=intro)
    elif cls == "static":
        ref = R.StaticLayer(h, w, introduction_masks=intro)
    else:
        ref = R.IntroductionLayer(h, w, introduction_masks=intro)
    for t in range(3):
        flow = R.post_process(rng.normal(0, 1.5, (h, w, 2)).astype(np.float32), R.BACKWARD)
        pm = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)
        layer.update(flow)
        if cls == "introduction":
            layer.introduce(0, pm, t)
            ref.update(flow, [pm], frame_numbers=[t])
        else:
            layer.gather(0, pm)
            ref.update(flow, [pm])
        comp.begin()
        layer.render(comp)
        img = ref.render()
        exp = R.composite(np.broadcast_to(np.array([9, 8, 7], np.uint8), (h, w, 3)), [img])
        np.testing.assert_array_equal(comp.download(), exp)
        data, rgba = layer.get_state()
        if data is not None:
            np.testing.assert_array_equal(data, ref.data)
        np.testing.assert_array_equal(rgba, np.asarray(ref.rgba))
    layer.close()
