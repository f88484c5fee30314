"""The flow-array steps either side of the hot path, on the GPU (tf_flow_* / tf_frame_* of
libtfhip.so).  Host-array functions with the reference's names and argument meaning; every one
uploads, runs the kernel and downloads (the `_dev` entry points they call take device pointers,
for callers that keep the arrays resident).

  merge_flows(kind, flows)        <- Pipeline.FLOW_MERGING_FUNCTIONS[kind](flows), pipeline.py:149-158
  upscale_array(arr, wf, hf)      <- utils.upscale_array, utils.py:417-418
  convolve_post_process(...)      <- the kernel step and what follows it in FlowSource.post_process,
                                     flow/sources/source.py:344-362
  render1d / render2d             <- output/render.py:9-48
  bgr_to_grey(frame, size)        <- cv2.resize(INTER_NEAREST) + cv2.cvtColor(BGR2GRAY), cv.py:461-466
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check
from .device import DevBuffer
from .masks import parse_color

MERGE_KINDS = {"first": 0, "sum": 1, "average": 2, "difference": 3, "product": 4, "maskbin": 5, "masklin": 6,
               "absmax": 7}
MAX_MERGE = 8


def _flow32(a) -> np.ndarray:
    a = np.ascontiguousarray(a, dtype=np.float32)
    if a.ndim != 3 or a.shape[2] != 2:
        raise ValueError(f"expected a flow of shape (H, W, 2), got {a.shape}")
    return a


def merge_flows(kind: str, flows) -> np.ndarray:
    if kind not in MERGE_KINDS:
        raise KeyError(kind)                                    # what the reference's dict lookup raises
    flows = [_flow32(f) for f in flows]
    if not flows or len(flows) > MAX_MERGE:
        raise ValueError(f"merge_flows takes 1..{MAX_MERGE} flows, got {len(flows)}")
    if any(f.shape != flows[0].shape for f in flows):
        raise ValueError("flows differ in shape")
    if kind == "absmax" and len(flows) != 2:
        raise ValueError("cannot reshape: absmax merges exactly two flows")   # utils.py:378 fails in reshape
    lib = _lib.load()
    bufs = [DevBuffer.from_array(f) for f in flows]
    out = DevBuffer(flows[0].nbytes)
    ptrs = (C.c_void_p * len(bufs))(*[b.ptr for b in bufs])
    check(lib.tf_flow_merge_dev(MERGE_KINDS[kind], len(bufs), ptrs, C.c_void_p(out.ptr), flows[0].size))
    res = out.download(flows[0].shape, np.float32)
    for b in bufs + [out]:
        b.close()
    return res


def upscale_array(arr, wf: int, hf: int) -> np.ndarray:
    a = _flow32(arr)
    wf, hf = int(wf), int(hf)
    h, w, _ = a.shape
    lib = _lib.load()
    src, dst = DevBuffer.from_array(a), DevBuffer(a.nbytes * wf * hf)
    check(lib.tf_flow_upscale_dev(C.c_void_p(src.ptr), C.c_void_p(dst.ptr), w, h, wf, hf))
    res = dst.download((h * hf, w * wf, 2), np.float32)
    src.close()
    dst.close()
    return res


def kernel_result_type(kernel: np.ndarray):
    """numpy.result_type(float32 flow, kernel): what scipy.signal.convolve2d computes in."""
    rt = np.result_type(np.float32, np.asarray(kernel).dtype)
    if rt not in (np.dtype(np.float32), np.dtype(np.float64)):
        raise NotImplementedError(f"convolution kernels of type {np.asarray(kernel).dtype} (result {rt})")
    return rt


def convolve_post_process(flow, kernel, direction: int | None) -> np.ndarray:
    """source.py:344-362: both channels through scipy.signal.convolve2d(mode="same", boundary="fill",
    fillvalue=0), then -- in the convolution's type -- the FORWARD inversion and the clip to the
    frame (direction 0 FORWARD, 1 BACKWARD, None: the convolution alone)."""
    f = _flow32(flow)
    kernel = np.asarray(kernel)
    if kernel.ndim != 2 or kernel.size == 0:
        raise ValueError("the convolution kernel must be a non-empty 2-D array")
    rt = kernel_result_type(kernel)
    wide = int(rt == np.float64)
    k = np.ascontiguousarray(kernel, dtype=rt)
    h, w, _ = f.shape
    lib = _lib.load()
    src, kb = DevBuffer.from_array(f), DevBuffer.from_array(k)
    dst = DevBuffer(max(1, f.size * rt.itemsize))
    scratch = DevBuffer(max(4, h * w * 4))
    check(lib.tf_flow_convolve_dev(C.c_void_p(src.ptr), C.c_void_p(kb.ptr), k.shape[0], k.shape[1], wide,
                                   C.c_void_p(dst.ptr), w, h))
    if direction is not None:
        check(lib.tf_flow_post_process_dev(C.c_void_p(dst.ptr), wide, w, h, int(direction), C.c_void_p(scratch.ptr)))
    res = dst.download((h, w, 2), rt)
    for b in (src, kb, dst, scratch):
        b.close()
    return res


def polar_filter(flow, polar, t: float) -> np.ndarray:
    """PolarFlowFilter.apply (filters.py:81-88), in place on a C-contiguous float32 flow; `polar` is a
    transflow_amd.exprs.PolarFilter."""
    from ._lib import TfPolarStep
    if not (isinstance(flow, np.ndarray) and flow.dtype == np.float32 and flow.flags.c_contiguous
            and flow.ndim == 3 and flow.shape[2] == 2):
        raise ValueError("the polar filter works in place on a C-contiguous float32 (H, W, 2) array")
    sr, st, wide_trig, wide_product = polar.programs(t)
    ar = (TfPolarStep * len(sr))(*[TfPolarStep(o, w, v) for o, w, v in sr])
    at = (TfPolarStep * len(st))(*[TfPolarStep(o, w, v) for o, w, v in st])
    lib = _lib.load()
    buf = DevBuffer.from_array(flow)
    check(lib.tf_flow_polar_dev(C.c_void_p(buf.ptr), flow.shape[0] * flow.shape[1], len(sr), ar, len(st), at,
                                int(wide_trig), int(wide_product)))
    flow[...] = buf.download(flow.shape, np.float32)
    buf.close()
    return flow


def _colors(colors, n):
    arr = np.array([parse_color(c) for c in colors], dtype=np.float32)   # render.py:17 / :37
    if arr.shape != (n, 3):
        raise ValueError(f"expected {n} colours")
    return (C.c_float * (3 * n))(*arr.ravel())


def render1d(arr, scale: float = 1, colors=None, binary: bool = False) -> np.ndarray:
    a = np.ascontiguousarray(arr, dtype=np.float32)
    h, w = a.shape[:2]
    a = a.reshape(h, w)
    lib = _lib.load()
    src, dst = DevBuffer.from_array(a), DevBuffer(max(1, a.size * 3))
    check(lib.tf_flow_render1d_dev(C.c_void_p(src.ptr), C.c_void_p(dst.ptr), a.size, float(scale),
                                   _colors(colors or ("#000000", "#ffffff"), 2), int(bool(binary))))
    res = dst.download((h, w, 3), np.uint8)
    src.close()
    dst.close()
    return res


def render2d(arr, scale: float = 1, colors=None) -> np.ndarray:
    a = _flow32(arr)
    h, w, _ = a.shape
    lib = _lib.load()
    src, dst = DevBuffer.from_array(a), DevBuffer(max(1, h * w * 3))
    check(lib.tf_flow_render2d_dev(C.c_void_p(src.ptr), C.c_void_p(dst.ptr), h * w, float(scale),
                                   _colors(colors or ("#ffff00", "#0000ff", "#ff00ff", "#00ff00"), 4)))
    res = dst.download((h, w, 3), np.uint8)
    src.close()
    dst.close()
    return res


def bgr_to_grey(frame, size=None) -> np.ndarray:
    """cv.py:461-466: nearest-neighbour resize to size = (width, height), then BGR -> grey."""
    f = np.ascontiguousarray(frame, dtype=np.uint8)
    if f.ndim != 3 or f.shape[2] != 3:
        raise ValueError(f"expected a BGR frame (H, W, 3), got {f.shape}")
    sh, sw, _ = f.shape
    w, h = (sw, sh) if size is None else (int(size[0]), int(size[1]))
    lib = _lib.load()
    src, dst = DevBuffer.from_array(f), DevBuffer(max(1, w * h))
    check(lib.tf_frame_grey_dev(C.c_void_p(src.ptr), sw, sh, C.c_void_p(dst.ptr), w, h))
    res = dst.download((h, w), np.uint8)
    src.close()
    dst.close()
    return res
