"""ctypes binding of oracle/libfbref.so (the C restatement in farneback_ref.c).

TEST INFRASTRUCTURE ONLY -- see the header of farneback_ref.c.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None


def build(force: bool = False) -> str:
    alt = os.environ.get("FBREF_LIBRARY")      # tools/oracle_asan.sh: the sanitizer build of the same source
    if alt:
        return alt
    so = os.path.join(_HERE, "libfbref.so")
    src = os.path.join(_HERE, "farneback_ref.c")
    if force or not os.path.exists(so) or (
            os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libfbref.so"], stdout=subprocess.DEVNULL)
    return so


VARIANTS = ("fma", "area", "polyf32")     # oracle/Makefile, target `variants`: sensitivity builds, never parity targets
_VARIANT_LIBS = {}


def variant_lib(name):
    """A sensitivity build of the same source (oracle/Makefile `variants`): "fma" = FMA contraction as an AVX2/FMA
    OpenCV wheel contracts, "area" = [VERIFY] 4's scalar-tail order on a half-size level, "polyf32" = the
    polynomial expansion's horizontal part in float.  Used to measure how far the default build (the parity
    target) and the HIP library sit from builds a real OpenCV could be."""
    if name not in VARIANTS:
        raise ValueError(name)
    if name not in _VARIANT_LIBS:
        so = os.path.join(_HERE, f"libfbref_{name}.so")
        src = os.path.join(_HERE, "farneback_ref.c")
        if not os.path.exists(so) or (os.path.exists(src) and os.path.getmtime(src) > os.path.getmtime(so)):
            subprocess.check_call(["make", "-C", _HERE, "-B", f"libfbref_{name}.so"], stdout=subprocess.DEVNULL)
        _VARIANT_LIBS[name] = C.CDLL(so)
    return _VARIANT_LIBS[name]


def lib():
    global _LIB
    if _LIB is None:
        _LIB = C.CDLL(build())
        _LIB.fbref_level_geometry.restype = C.c_double
    return _LIB


def _f32(a):
    return np.ascontiguousarray(a, dtype=np.float32)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


def gaussian_kernel(n, sigma):
    out = np.zeros(n, np.float32)
    lib().fbref_gaussian_kernel(C.c_int(n), C.c_double(sigma), _p(out))
    return out


def prepare_gaussian(n, sigma):
    g = np.zeros(2 * n + 1, np.float32)
    xg = np.zeros(2 * n + 1, np.float32)
    xxg = np.zeros(2 * n + 1, np.float32)
    ig = np.zeros(4, np.float64)
    off = n * 4
    lib().fbref_prepare_gaussian(C.c_int(n), C.c_double(sigma), C.c_void_p(g.ctypes.data + off),
                                 C.c_void_p(xg.ctypes.data + off), C.c_void_p(xxg.ctypes.data + off), _p(ig))
    return g, xg, xxg, ig


def num_levels(w, h, pyr_scale, levels):
    return int(lib().fbref_num_levels(C.c_int(w), C.c_int(h), C.c_double(pyr_scale), C.c_int(levels)))


def level_geometry(w, h, pyr_scale, k):
    wk, hk, ksz = C.c_int(), C.c_int(), C.c_int()
    sigma = lib().fbref_level_geometry(C.c_int(w), C.c_int(h), C.c_double(pyr_scale), C.c_int(k),
                                       C.byref(wk), C.byref(hk), C.byref(ksz))
    return wk.value, hk.value, ksz.value, float(sigma)


def level_image(img, pyr_scale, k):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    wk, hk, _, _ = level_geometry(w, h, pyr_scale, k)
    out = np.zeros((hk, wk), np.float32)
    lib().fbref_level_image(_p(img), C.c_int(w), C.c_int(h), C.c_double(pyr_scale), C.c_int(k), _p(out))
    return out


def gaussian_blur_u8(img, ksz, sigma):
    img = np.ascontiguousarray(img, np.uint8)
    h, w = img.shape
    out = np.zeros((h, w), np.float32)
    lib().fbref_gaussian_blur_u8(_p(img), C.c_int(w), C.c_int(h), C.c_int(ksz), C.c_double(sigma), _p(out))
    return out


def resize_linear(src, dw, dh):
    src = _f32(src)
    if src.ndim == 2:
        sh, sw = src.shape
        cn = 1
        out = np.zeros((dh, dw), np.float32)
    else:
        sh, sw, cn = src.shape
        out = np.zeros((dh, dw, cn), np.float32)
    lib().fbref_resize_linear(_p(src), C.c_int(sw), C.c_int(sh), C.c_int(cn), _p(out), C.c_int(dw), C.c_int(dh))
    return out


def polyexp(img, n, sigma):
    """float32 [H,W] -> [H,W,5] (OpenCV channel order: y, x, yy, xx, xy)."""
    img = _f32(img)
    h, w = img.shape
    out = np.zeros((h, w, 5), np.float32)
    lib().fbref_polyexp(_p(img), C.c_int(w), C.c_int(h), C.c_int(n), C.c_double(sigma), _p(out))
    return out


def update_matrices(r0, r1, flow):
    r0, r1, flow = _f32(r0), _f32(r1), _f32(flow)
    h, w, _ = flow.shape
    m = np.zeros((h, w, 5), np.float32)
    lib().fbref_update_matrices(_p(r0), _p(r1), _p(flow), _p(m), C.c_int(w), C.c_int(h), C.c_int(0), C.c_int(h))
    return m


def update_flow_blur(r0, r1, flow, m, winsize, update):
    """Returns (new_flow, new_M); inputs are not modified."""
    r0, r1 = _f32(r0), _f32(r1)
    flow, m = _f32(flow).copy(), _f32(m).copy()
    h, w, _ = flow.shape
    lib().fbref_update_flow_blur(_p(r0), _p(r1), _p(flow), _p(m), C.c_int(w), C.c_int(h), C.c_int(winsize),
                                 C.c_int(1 if update else 0))
    return flow, m


OPTFLOW_USE_INITIAL_FLOW = 4
OPTFLOW_FARNEBACK_GAUSSIAN = 256


def resize_area(src, dw, dh):
    """cv2.resize(src, (dw, dh), interpolation=INTER_AREA) for a shrinking float image [H,W,C]."""
    src = _f32(src)
    sh, sw, cn = src.shape
    out = np.zeros((dh, dw, cn), np.float32)
    lib().fbref_resize_area(_p(src), C.c_int(sw), C.c_int(sh), C.c_int(cn), _p(out), C.c_int(dw), C.c_int(dh))
    return out


def area_tab(ssize, dsize):
    """computeResizeAreaTab's (source index, destination index, weight) triples for one axis."""
    si = np.zeros(2 * ssize + 2, np.int32)
    di = np.zeros(2 * ssize + 2, np.int32)
    al = np.zeros(2 * ssize + 2, np.float32)
    n = lib().fbref_area_tab(C.c_int(ssize), C.c_int(dsize), _p(si), _p(di), _p(al))
    return si[:n].copy(), di[:n].copy(), al[:n].copy()


def gaussian_window(m):
    k = np.zeros(m + 1, np.float32)
    lib().fbref_gaussian_window(C.c_int(m), _p(k))
    return k


def update_flow_gaussian(r0, r1, flow, m, winsize, update):
    """FarnebackUpdateFlow_GaussianBlur: returns (new_flow, new_M); inputs are not modified."""
    r0, r1 = _f32(r0), _f32(r1)
    flow, m = _f32(flow).copy(), _f32(m).copy()
    h, w, _ = flow.shape
    lib().fbref_update_flow_gaussian(_p(r0), _p(r1), _p(flow), _p(m), C.c_int(w), C.c_int(h), C.c_int(winsize),
                                     C.c_int(1 if update else 0))
    return flow, m


def calc(prev, nxt, pyr_scale=0.5, levels=3, winsize=15, iterations=3, poly_n=5, poly_sigma=1.2, flags=0, flow=None,
         variant=None):
    """Same argument meaning as cv2.calcOpticalFlowFarneback (reference cv.py:479-490); `flow` is the
    initial flow read when flags has OPTFLOW_USE_INITIAL_FLOW (never modified: a new array is returned).
    `variant`: one of VARIANTS -- the same call through a sensitivity build (variant_lib); None = the parity target."""
    prev = np.ascontiguousarray(prev, np.uint8)
    nxt = np.ascontiguousarray(nxt, np.uint8)
    assert prev.shape == nxt.shape and prev.ndim == 2
    h, w = prev.shape
    if flow is not None:
        flow = np.array(flow, dtype=np.float32, order="C", copy=True)
        assert flow.shape == (h, w, 2)
    else:
        flow = np.zeros((h, w, 2), np.float32)
    rc = (variant_lib(variant) if variant else lib()).fbref_calc(_p(prev), _p(nxt), C.c_int(w), C.c_int(h), _p(flow), C.c_double(pyr_scale), C.c_int(levels),
                          C.c_int(winsize), C.c_int(iterations), C.c_int(poly_n), C.c_double(poly_sigma), C.c_int(flags))
    if rc != 0:
        raise ValueError("fbref_calc: unsupported arguments")
    return flow


def calc_batch(prevs, nexts, threads, pyr_scale=0.5, levels=3, winsize=15, iterations=3, poly_n=5, poly_sigma=1.2, flags=0):
    """n independent pairs, one per OpenMP thread; each result is bit-identical to calc()'s."""
    prevs = [np.ascontiguousarray(a, np.uint8) for a in prevs]
    nexts = [np.ascontiguousarray(a, np.uint8) for a in nexts]
    n = len(prevs)
    assert n == len(nexts) and n > 0
    h, w = prevs[0].shape
    assert all(a.shape == (h, w) for a in prevs + nexts)
    flows = [np.zeros((h, w, 2), np.float32) for _ in range(n)]
    arr = lambda xs: (C.c_void_p * n)(*[x.ctypes.data for x in xs])
    rc = lib().fbref_calc_batch(arr(prevs), arr(nexts), C.c_int(n), C.c_int(int(threads)), C.c_int(w), C.c_int(h),
                                arr(flows), C.c_double(pyr_scale), C.c_int(levels), C.c_int(winsize), C.c_int(iterations),
                                C.c_int(poly_n), C.c_double(poly_sigma), C.c_int(flags))
    if rc != 0:
        raise ValueError("fbref_calc_batch: unsupported arguments")
    return flows
