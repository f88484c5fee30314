"""Mask and colour arguments of the compositor, host side.

Same argument grammar and results as the reference's helpers
(transflow/utils.py:13-144 load_float_mask / load_bool_mask, :316-324 parse_color):
these only PARSE user arguments into the arrays the kernels consume; they run once
at setup, not per frame.  Pinned by tests/golden/masks.npz.
"""
from __future__ import annotations

import re

import numpy as np


_LENGTH = re.compile(r"\d+%?\Z")


def _length(token: str, extent: int) -> int:
    """A mask dimension: pixels ('12'), a share of the axis it is measured along ('40%',
    truncated), or nothing ('' = 0).  Argument grammar of utils.py:13-18."""
    token = token.strip()
    if not token:
        return 0
    if token[-1] == "%":
        return int(float(token[:-1]) / 100 * extent)
    return int(token)


def _picked(n: int, start, stop) -> np.ndarray:
    """Which of n rows (columns) the Python slice start:stop picks, as a bool vector: negative
    and overshooting bounds mean what they mean to numpy indexing, which is what the reference's
    results are made of when a dimension exceeds the frame."""
    v = np.zeros(n, bool)
    v[slice(start, stop)] = True
    return v


def _edges(n: int, head: int, tail: int) -> np.ndarray:
    """The first `head` and the last `tail` of n positions."""
    k = np.arange(n)
    return (k < head) | (k >= n - tail)


# ---- shape builders: (args, h, w) -> mask.  Stripes, frames and boxes are outer products of
# one row vector and one column vector; discs are a distance test. ----------------------------
def _frame(sides):
    """border / border-<side>: `sides` maps the argument list to (top, right, bottom, left)."""
    def build(args, h, w):
        top, right, bottom, left = sides(args, h, w)
        return (_edges(h, top, bottom)[:, None] | _edges(w, left, right)[None, :]).astype(np.float32)
    return build


def _all_sides(args, h, w):
    if len(args) not in (1, 2, 4):
        raise ValueError(f"Invalid number of argument {len(args)} for border mask")
    # vertical extents (top, bottom) are measured along the height, horizontal ones along the width
    v = [_length(a, (h, w)[i % 2]) for i, a in enumerate(args)]
    return {1: lambda: (v[0],) * 4, 2: lambda: (v[0], v[1], v[0], v[1]), 4: lambda: tuple(v)}[len(v)]()


def _one_side(index):
    def sides(args, h, w):
        out = [0, 0, 0, 0]
        out[index] = _length(":".join(args), (h, w)[index % 2])
        return out
    return sides


def _stripe(axis):
    """hline / vline: a centred band of the given thickness across the frame."""
    def build(args, h, w):
        n = (h, w)[axis]
        t = _length(args[0], n)
        first = (n - t) // 2
        band = _picked(n, first, first + t)
        rows, cols = (band, np.ones(w, bool)) if axis == 0 else (np.ones(h, bool), band)
        return (rows[:, None] & cols[None, :]).astype(np.float32)
    return build


def _disc(args, h, w):
    """circle: the open disc around (h // 2, w // 2).  A BOOL array, as utils.py:92 leaves it
    (so ':inv' of it is float64) -- tests/golden/masks.npz pins the dtypes."""
    r = _length(args[0], min(h, w))
    dy = np.arange(h)[:, None] - h // 2
    dx = np.arange(w)[None, :] - w // 2
    return dx * dx + dy * dy < r * r


def _box(args, h, w):
    """rect: centred box, width[:height]; one argument serves both axes (each against its own)."""
    if len(args) > 2:
        raise ValueError(f"Invalid number of argument {len(args)} for rect mask")
    bw, bh = _length(args[0], w), _length(args[-1], h)
    rows = ~(_picked(h, None, h // 2 - bh // 2) | _picked(h, h // 2 + bh // 2, None))
    cols = ~(_picked(w, None, w // 2 - bw // 2) | _picked(w, w // 2 + bw // 2, None))
    return (rows[:, None] & cols[None, :]).astype(np.float32)


def _dots(args, h, w):
    """grid: rows:cols:radius -- one disc stamped at the centre of every cell, cells in raster
    order (a stamp is a whole 2r x 2r block, so where blocks overlap the later one stands)."""
    nrows, ncols, r = (int(a) for a in args)
    k = np.arange(2 * r) - r
    stamp = k[None, :] ** 2 + k[:, None] ** 2 < r * r
    out = np.zeros((h, w), np.float32)
    ch, cw = h // nrows, w // ncols
    for cell in range(nrows * ncols):
        i0 = ch * (cell // ncols) + ch // 2 - r
        j0 = cw * (cell % ncols) + cw // 2 - r
        out[i0:i0 + 2 * r, j0:j0 + 2 * r] = stamp
    return out


# ---- the rule table: name -> (do these arguments make it a rule?, builder).  A name whose
# arguments do not fit is not a rule and the string is a file name -- the reference's patterns
# (utils.py:64-112) send the same strings to PIL.  circle / rect / grid are recognised by their
# leading arguments alone there; what follows is ignored (circle) or counted (rect, grid). -------
_DIGITS = re.compile(r"\d")


def _lengths(args, blank_ok=False):
    return all(_LENGTH.match(a) or (blank_ok and a == "") for a in args)


def _is_frame(args):
    return 1 <= len(args) <= 4 and any(args) and _lengths(args, blank_ok=True)


def _leading(n):
    return lambda args: len(args) >= n and all(_DIGITS.match(a) for a in args[:n])


def _bare(args):
    return not args


_SHAPES = {
    "zeros": (_bare, lambda a, h, w: np.zeros((h, w), np.float32)),
    "ones": (_bare, lambda a, h, w: np.ones((h, w), np.float32)),
    "random": (_bare, lambda a, h, w: np.random.rand(h, w).astype(np.float32)),
    "border": (_is_frame, _frame(_all_sides)),
    "border-top": (_is_frame, _frame(_one_side(0))),
    "border-right": (_is_frame, _frame(_one_side(1))),
    "border-bottom": (_is_frame, _frame(_one_side(2))),
    "border-left": (_is_frame, _frame(_one_side(3))),
    "hline": (lambda args: len(args) == 1 and _lengths(args), _stripe(0)),
    "vline": (lambda args: len(args) == 1 and _lengths(args), _stripe(1)),
    "circle": (_leading(1), _disc),
    "rect": (_leading(1), _box),
    "grid": (_leading(3), _dots),
}


def _shape_rule(spec: str):
    name, *args = spec.split(":")
    fits, build = _SHAPES.get(name.lower(), (None, None))
    if fits is None or not fits(args):
        return None
    return lambda h, w: build(args, h, w)


def _from_image(path: str) -> np.ndarray:
    import PIL.Image
    with PIL.Image.open(path) as image:
        px = np.asarray(image).astype(np.float32)
    if px.ndim == 2:
        return px / 255
    if px.ndim == 3:
        return px[:, :, :3].mean(axis=2) / 255          # an alpha channel is ignored
    raise ValueError(f"Image has wrong number of dimensions {px.ndim}, expected 2 or 3")


def load_float_mask(mask_path: str | None, shape=(0, 0), default: float = 0) -> np.ndarray:
    if mask_path is None:
        return np.full(shape, default, dtype=np.float32)
    inverse = mask_path.endswith(":inv")
    spec = mask_path.removesuffix(":inv")
    rule = _shape_rule(spec)
    mask = rule(*shape) if rule else _from_image(spec)
    return 1.0 - mask if inverse else mask


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
