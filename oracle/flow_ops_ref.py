"""CPU oracle for the flow-array steps around the hot path: flow merging,
integer upscaling, the convolution-kernel pre-step of post_process, flow visualisation.

TEST INFRASTRUCTURE ONLY (see oracle/remap_ref.py's header).  Written as explicit per-element
formulas in the arithmetic numpy / scipy use, each function citing the reference lines it follows;
pinned by tests/golden/flow_ops.npz, produced by running the reference (tools/capture_golden.py
--flowops-only).
"""
from __future__ import annotations

import numpy as np

from . import remap_ref

MERGE_KINDS = ("first", "sum", "average", "difference", "product", "maskbin", "masklin", "absmax")


def merge(kind: str, flows) -> np.ndarray:
    """Pipeline.FLOW_MERGING_FUNCTIONS, transflow/pipeline.py:149-158 (+ utils.py:359-381).
    All float32, operands combined strictly left to right (numpy.sum over the leading axis of a
    stacked list adds the arrays one after the other)."""
    f = [np.asarray(a, np.float32) for a in flows]
    if kind == "first":                                             # :150
        return f[0]
    if kind in ("sum", "average"):                                  # :151-152
        acc = f[0].copy()
        for a in f[1:]:
            acc = acc + a
        return acc / np.float32(len(f)) if kind == "average" else acc
    if kind == "difference":                                        # :153  flows[0] - sum(flows[1:])
        if len(f) == 1:
            return f[0] - np.float32(0)
        acc = f[1].copy()
        for a in f[2:]:
            acc = acc + a
        return f[0] - acc
    if kind in ("product", "maskbin", "masklin"):                   # :154-156, utils.py:359-373
        rest = f[1:]
        if kind == "maskbin":                                       # |x| > 0.2 as float32 -> 1 else 0
            rest = [(np.abs(a) > np.float32(0.2)).astype(np.float32) for a in rest]
        elif kind == "masklin":
            rest = [np.abs(a) for a in rest]
        out = f[0]
        for a in rest:
            out = out * a
        return out
    if kind == "absmax":                                            # :157, utils.py:376-381 (two arrays)
        if len(f) != 2:
            raise ValueError("absmax merges exactly two flows")
        a0, a1 = np.abs(f[0]), np.abs(f[1])                         # argmax: first maximum, NaN counts as one
        return np.where((a1 > a0) | (np.isnan(a1) & ~np.isnan(a0)), f[1], f[0])
    raise ValueError(kind)


def upscale(arr: np.ndarray, wf: int, hf: int) -> np.ndarray:
    """utils.upscale_array, utils.py:417-418: nearest-neighbour repeat by (hf, wf) of
    (x * wf, y * hf); the float64 detour of numpy.kron rounds once, like a float32 multiply."""
    a = np.asarray(arr, np.float32) * np.array([wf, hf], np.float32)
    return np.repeat(np.repeat(a, hf, axis=0), wf, axis=1)


def convolve_same(channel: np.ndarray, kernel: np.ndarray) -> np.ndarray:
    """scipy.signal.convolve2d(channel, kernel, mode="same", boundary="fill", fillvalue=0)
    (source.py:346-347).  Result type = numpy.result_type(float32, kernel dtype); every output is
    sum over kernel rows j then columns k, in that order, of kernel[j, k] * in[m + (Kh-1)//2 - j,
    n + (Kw-1)//2 - k] with zeros outside, accumulated in the result type, multiply and add
    rounded separately (scipy/signal/_firfilter.c, pylab_convolve_2d)."""
    rt = np.result_type(np.float32, kernel.dtype)
    a = np.asarray(channel).astype(rt)
    k = np.asarray(kernel).astype(rt)
    h, w = a.shape
    kh, kw = k.shape
    oy, ox = (kh - 1) // 2, (kw - 1) // 2
    pad = np.zeros((h + 2 * kh, w + 2 * kw), rt)
    pad[kh:kh + h, kw:kw + w] = a
    out = np.zeros((h, w), rt)
    for j in range(kh):
        for kk in range(kw):
            y0, x0 = kh + oy - j, kw + ox - kk
            out = out + k[j, kk] * pad[y0:y0 + h, x0:x0 + w]
    return out


def post_process_with_kernel(flow: np.ndarray, kernel: np.ndarray, direction: int) -> np.ndarray:
    """source.py:344-363 from the kernel step on: both channels convolved, stacked (the flow is now
    of the convolution's type, float64 unless the kernel is float32), then the direction handling
    and the clip in that type."""
    out = np.stack([convolve_same(flow[:, :, 0], kernel), convolve_same(flow[:, :, 1], kernel)], axis=-1)
    return post_process_any(out, direction)


def post_process_any(flow: np.ndarray, direction: int) -> np.ndarray:
    """remap_ref.post_process for a flow of any float type (source.py:349-362)."""
    h, w, _ = flow.shape
    jj = np.arange(w, dtype=np.int32)[None, :]
    ii = np.arange(h, dtype=np.int32)[:, None]

    def clip(f):
        np.clip(f[:, :, 0], -jj, w - 1 - jj, out=f[:, :, 0])
        np.clip(f[:, :, 1], -ii, h - 1 - ii, out=f[:, :, 1])

    if direction == remap_ref.FORWARD:
        clip(flow)
        fi = np.rint(flow).astype(np.int32)
        d = (fi[:, :, 1] * w + fi[:, :, 0]).ravel()
        p = np.arange(h * w, dtype=np.int64)
        moving = d != 0
        winner = np.full(h * w, -1, np.int64)
        tgt = np.clip(p[moving] + d[moving], 0, h * w - 1)
        np.maximum.at(winner, tgt, p[moving])                      # ascending put: the largest source wins
        src = np.where(winner >= 0, winner, p)
        flow[:, :, 0] = (src % w - p % w).reshape(h, w)
        flow[:, :, 1] = (src // w - p // w).reshape(h, w)
    clip(flow)
    return flow


def _parse_color(c: str):
    c = c.lstrip("#")
    return tuple(int(c[i:i + 2], 16) for i in (0, 2, 4))


def render1d(arr: np.ndarray, scale=1, colors=("#000000", "#ffffff"), binary=False) -> np.ndarray:
    """output/render.py:9-27.  float32 throughout for a float32 input and a Python scalar scale."""
    a = np.asarray(arr, np.float32)
    ca, cb = (np.array(_parse_color(c), np.float32) for c in colors)
    s = np.float32(scale)
    if binary:
        coeff = np.clip(np.rint(s * a), 0, 1)[..., None]           # :20
        coeff_a, coeff_b = np.float32(1) - coeff, coeff
    else:
        coeff_a = np.clip(np.float32(1) - s * a, 0, 1)[..., None]  # :24
        coeff_b = np.clip(s * a, 0, 1)[..., None]                  # :25
    frame = coeff_a * ca + coeff_b * cb                            # :26
    return np.clip(frame, 0, 255).astype(np.uint8)


def render2d(arr: np.ndarray, scale=1, colors=("#ffff00", "#0000ff", "#ff00ff", "#00ff00")) -> np.ndarray:
    """output/render.py:30-48."""
    a = np.asarray(arr, np.float32)
    cy, cb, cm, cg = (np.array(_parse_color(c), np.float32) for c in colors)
    s, one = np.float32(scale), np.float32(1)
    k_y = np.clip(one + s * a[:, :, 0], 0, 1)[..., None]
    k_b = np.clip(one - s * a[:, :, 0], 0, 1)[..., None]
    k_m = np.clip(one + s * a[:, :, 1], 0, 1)[..., None]
    k_g = np.clip(one - s * a[:, :, 1], 0, 1)[..., None]
    frame = np.float32(0.5) * (((k_y * cy + k_b * cb) + k_m * cm) + k_g * cg)   # :43-47, left to right
    return np.clip(frame, 0, 255).astype(np.uint8)


def polar(flow: np.ndarray, expr_radius: str, expr_theta: str, t: float) -> np.ndarray:
    """PolarFlowFilter.apply, flow/filters.py:81-88, in place: the two user expressions are evaluated by
    Python on numpy arrays exactly as the reference does (utils.parse_lambda_expression, utils.py:409-414;
    `math`, `numpy`, `random`, `re`, `os` in scope)."""
    import math
    import os
    import random
    import re
    scope = {"math": math, "numpy": np, "random": random, "re": re, "os": os}
    h, w, _ = flow.shape
    radius = np.sqrt(flow[:, :, 0] * flow[:, :, 0] + flow[:, :, 1] * flow[:, :, 1])   # :83, float32 pair norm
    theta = np.arctan2(flow[:, :, 1], flow[:, :, 0])                                  # :84
    new_radius = eval("lambda t, r, a: " + expr_radius, scope)(t, radius, theta)
    new_theta = eval("lambda t, r, a: " + expr_theta, scope)(t, radius, theta)
    flow[:, :, 1] = new_radius * np.sin(new_theta)                                    # :87
    flow[:, :, 0] = new_radius * np.cos(new_theta)                                    # :88
    return flow
