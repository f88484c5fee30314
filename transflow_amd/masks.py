"""Mask and colour arguments of the compositor, host side.

Same argument grammar and results as the reference's helpers
(transflow/utils.py:13-144 load_float_mask / load_bool_mask, :316-324 parse_color):
these only PARSE user arguments into the arrays the kernels consume; they run once
at setup, not per frame.  Pinned by tests/golden/masks.npz.
"""
from __future__ import annotations

import re

import numpy as np


def _dim(arg: str, parent: int) -> int:
    """'12' -> 12, '40%' -> int(0.4*parent), '' -> 0   (utils.py:13-18)"""
    arg = arg.strip()
    if arg == "":
        return 0
    if arg.endswith("%"):
        return int(float(arg[:-1]) / 100 * parent)
    return int(arg)


def _border(spec: str, h: int, w: int):
    name, args = spec.lower().split(":", 1)
    top = right = bottom = left = 0
    if name == "border":
        vals = [_dim(a, h if i % 2 == 0 else w) for i, a in enumerate(args.split(":"))]
        if len(vals) == 1:
            top = right = bottom = left = vals[0]
        elif len(vals) == 2:
            top = bottom = vals[0]
            right = left = vals[1]
        elif len(vals) == 4:
            top, right, bottom, left = vals
        else:
            raise ValueError(f"Invalid number of argument {len(vals)} for border mask")
    elif name == "border-top":
        top = _dim(args, h)
    elif name == "border-right":
        right = _dim(args, w)
    elif name == "border-bottom":
        bottom = _dim(args, h)
    elif name == "border-left":
        left = _dim(args, w)
    else:
        raise ValueError(f"Invalid border rule name {name}")
    return top, right, bottom, left


_RE_BORDER = re.compile(r"^border(\-(top|right|bottom|left))?:(\d+%?:|:|\d+%?$){1,4}$", re.I)
_RE_LINE = re.compile(r"^[hv]line:\d+%?$", re.I)
_RE_CIRCLE = re.compile(r"circle:\d+%?", re.I)
_RE_RECT = re.compile(r"rect:\d+%?(:\d+%?)?", re.I)
_RE_GRID = re.compile(r"grid:\d+:\d+:\d+?", re.I)


def load_float_mask(mask_path: str | None, shape=(0, 0), default: float = 0) -> np.ndarray:
    h, w = shape
    if mask_path is None:
        return np.full(shape, default, dtype=np.float32)
    inverse = mask_path.endswith(":inv")
    if inverse:
        mask_path = mask_path[:-4]
    low = mask_path.lower()
    if low == "zeros":
        arr = np.zeros(shape, np.float32)
    elif low == "ones":
        arr = np.ones(shape, np.float32)
    elif low == "random":
        arr = np.random.rand(*shape).astype(np.float32)
    elif _RE_BORDER.match(mask_path):
        top, right, bottom, left = _border(mask_path, h, w)
        arr = np.zeros(shape, np.float32)
        if top:
            arr[:top, :] = 1
        if right:
            arr[:, -right:] = 1
        if bottom:
            arr[-bottom:, :] = 1
        if left:
            arr[:, :left] = 1
    elif _RE_LINE.match(mask_path):
        name, a = low.split(":")
        arr = np.zeros(shape, np.float32)
        if name == "hline":
            t = _dim(a, h)
            i = (h - t) // 2
            arr[i:i + t, :] = 1
        else:
            t = _dim(a, w)
            j = (w - t) // 2
            arr[:, j:j + t] = 1
    elif _RE_CIRCLE.match(mask_path):
        radius = _dim(low.split(":")[1], min(shape))
        ii = np.arange(h)[:, None] - h // 2
        jj = np.arange(w)[None, :] - w // 2
        arr = jj ** 2 + ii ** 2 < radius ** 2          # bool, like the reference (utils.py:92)
    elif _RE_RECT.match(mask_path):
        args = mask_path[mask_path.index(":") + 1:].split(":")
        if len(args) == 1:
            rw, rh = _dim(args[0], w), _dim(args[0], h)
        elif len(args) == 2:
            rw, rh = _dim(args[0], w), _dim(args[1], h)
        else:
            raise ValueError(f"Invalid number of argument {len(args)} for rect mask")
        arr = np.ones(shape, np.float32)
        arr[:h // 2 - rh // 2, :] = 0
        arr[h // 2 + rh // 2:, :] = 0
        arr[:, :w // 2 - rw // 2] = 0
        arr[:, w // 2 + rw // 2:] = 0
    elif _RE_GRID.match(mask_path):
        nrows, ncols, radius = (int(a) for a in mask_path[mask_path.index(":") + 1:].split(":"))
        dia = 2 * radius
        k = np.arange(dia) - radius
        disc = k[None, :] ** 2 + k[:, None] ** 2 < radius ** 2
        arr = np.zeros(shape, np.float32)
        ch, cw = h // nrows, w // ncols
        for i in range(nrows):
            for j in range(ncols):
                i0, j0 = ch * i + ch // 2 - radius, cw * j + cw // 2 - radius
                arr[i0:i0 + dia, j0:j0 + dia] = disc
    else:
        import PIL.Image
        with PIL.Image.open(mask_path) as image:
            arr = np.array(image).astype(np.float32)
        if arr.ndim == 2:
            arr /= 255
        elif arr.ndim == 3:
            arr = np.mean(arr[:, :, :3], axis=2) / 255
        else:
            raise ValueError(f"Image has wrong number of dimensions {arr.ndim}, expected 2 or 3")
    if inverse:
        arr = 1.0 - arr
    return arr


def load_bool_mask(mask_path: str | None, shape=(0, 0), default: bool = False) -> np.ndarray:
    """utils.py:143-144: round (half-even) the float mask, then to bool."""
    return np.round(load_float_mask(mask_path, shape, float(default))).astype(bool)


def parse_color(string: str) -> tuple[int, int, int]:
    """utils.py:316-324: CSS name, 'rgb(r, g, b)' / '(r,g,b)', or hex with optional '#'/'0x'."""
    from PIL import ImageColor
    name = string.lower()
    if name in ImageColor.colormap and name.isalpha():
        v = ImageColor.colormap[name]
        rgb = ImageColor.getrgb(v) if isinstance(v, str) else v
        return tuple(int(c) for c in rgb[:3])
    m = re.match(r"^(?:rgb)?\((\d+), ?(\d+), ?(\d+)\)$", string, re.I)
    if m is not None:
        return int(m.group(1)), int(m.group(2)), int(m.group(3))
    x = int(string.replace("#", "").replace("0x", "").replace("x", ""), 16)
    return (x >> 16) & 255, (x >> 8) & 255, x & 255
