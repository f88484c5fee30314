"""CPU oracle for the remap half of the hot path (numpy, integer/byte exact).

TEST INFRASTRUCTURE ONLY.  Nothing under ``transflow_amd/`` may import this
module; only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s
``cpu_baseline`` leg use it, and only as the checker.

This is a restatement, written as per-pixel gather formulas, of what the
reference computes with index lists and ``numpy.put``.  Every function cites
the reference lines it follows (paths relative to ``/root/reference``).
Pinned by ``tests/golden/remap_*.npz`` -- vectors produced by the reference's
own code (script: ``tools/capture_golden.py``) -- and by the known answers of
``tests/test_compositor.py:20-54``.

Conventions: flat index ``p = i*W + j``; flow channel
0 is dx (columns), channel 1 is dy (rows); ``data[p] = (i_ref, j_ref, alpha,
source)`` int32.
"""
from __future__ import annotations

import numpy as np

FORWARD = 0
BACKWARD = 1

INDEX_I, INDEX_J, INDEX_ALPHA, INDEX_SOURCE = 0, 1, 2, 3


# --------------------------------------------------------------------------
# B1  FlowSource.post_process        transflow/flow/sources/source.py:337-363
# --------------------------------------------------------------------------
def _clip_to_frame(flow: np.ndarray) -> None:
    """In-place clip so that p + flow stays inside the frame.

    source.py:361-362 with the tables of source.py:250-263
    (fx in [-j, W-1-j], fy in [-i, H-1-i]); float32 clip against int bounds.
    """
    h, w, _ = flow.shape
    jj = np.arange(w, dtype=np.int32)[None, :]
    ii = np.arange(h, dtype=np.int32)[:, None]
    np.clip(flow[:, :, 0], -jj, w - 1 - jj, out=flow[:, :, 0])
    np.clip(flow[:, :, 1], -ii, h - 1 - ii, out=flow[:, :, 1])


def post_process(flow: np.ndarray, direction: int) -> np.ndarray:
    """source.py:337-363 without the optional filter/mask/kernel pre-steps.

    Works in place on ``flow`` (the reference mutates its argument) and
    returns it.  FORWARD turns the push field into a gather map: every source
    p with a non-zero rounded displacement d writes its own coordinates at
    target p+d, ascending p, last write wins (source.py:349-360).
    """
    assert flow.dtype == np.float32 and flow.ndim == 3 and flow.shape[2] == 2
    h, w, _ = flow.shape
    if direction == FORWARD:
        _clip_to_frame(flow)                                  # :350-351
        fi = np.rint(flow).astype(np.int32)                   # :352 (half-even)
        d = (fi[:, :, 1] * w + fi[:, :, 0]).ravel()           # :353
        p = np.arange(h * w, dtype=np.int64)
        moving = d != 0                                       # :354
        # winner[t] = largest source p that lands on t (ascending put order)
        winner = np.full(h * w, -1, dtype=np.int64)
        tgt = np.clip(p[moving] + d[moving], 0, h * w - 1)    # mode="clip" :357
        np.maximum.at(winner, tgt, p[moving])
        src = np.where(winner >= 0, winner, p)                # untouched keep base
        flow[:, :, 0] = ((src % w) - (p % w)).reshape(h, w)   # :359
        flow[:, :, 1] = ((src // w) - (p // w)).reshape(h, w)  # :360
    _clip_to_frame(flow)                                      # :361-362
    return flow


# --------------------------------------------------------------------------
# N1  optional pre-steps of post_process: filters and mask
#     transflow/flow/filters.py:36-72, transflow/flow/sources/source.py:339-343
# --------------------------------------------------------------------------
def _pair_norm(flow: np.ndarray) -> np.ndarray:
    """numpy.linalg.norm(flow.reshape(N, 2), axis=1) for float32: sqrt(x*x + y*y) in float32."""
    x, y = flow[:, :, 0], flow[:, :, 1]
    return np.sqrt(x * x + y * y)


def filter_scale(flow: np.ndarray, value) -> None:
    """filters.py:41-42, in place.  `value` keeps its Python/numpy type: a Python float is a weak
    scalar (float32 arithmetic), a numpy.float64 is not (float64 arithmetic, cast back)."""
    flow *= value


def filter_threshold(flow: np.ndarray, value) -> None:
    """filters.py:50-55, in place."""
    flow[_pair_norm(flow) <= value] = 0


def filter_clip(flow: np.ndarray, value) -> None:
    """filters.py:63-71, in place: float64 factors (1 below the threshold), value/norm above."""
    norm = _pair_norm(flow)
    factors = np.ones(norm.shape)
    hit = norm >= value
    factors[hit] = value / norm[hit]
    flow[:, :, 0] *= factors
    flow[:, :, 1] *= factors


FILTERS = {"scale": filter_scale, "threshold": filter_threshold, "clip": filter_clip}


def pre_steps(flow: np.ndarray, ops=(), mask=None) -> np.ndarray:
    """source.py:339-343: filters in place, then (if a mask is set) a NEW array mask*flow."""
    for name, value in ops:
        FILTERS[name](flow, value)
    if mask is not None:
        flow = np.multiply(np.asarray(mask, np.float32).reshape(flow.shape[0], flow.shape[1], 1), flow)
    return flow


# --------------------------------------------------------------------------
# Layer parameters                    transflow/config.py:57-104
# --------------------------------------------------------------------------
class LayerParams:
    """The subset of LayerConfig (config.py:88-98) the moveref layer reads."""

    def __init__(self, transparent_pixels_can_move=False,
                 pixels_can_move_to_empty_spot=True,
                 pixels_can_move_to_filled_spot=True,
                 moving_pixels_leave_empty_spot=False,
                 reset_mode="off", reset_random_factor=1.0,
                 reset_constant_step=1.0, reset_linear_factor=0.1,
                 reset_source=False):
        self.transparent_pixels_can_move = bool(transparent_pixels_can_move)
        self.pixels_can_move_to_empty_spot = bool(pixels_can_move_to_empty_spot)
        self.pixels_can_move_to_filled_spot = bool(pixels_can_move_to_filled_spot)
        self.moving_pixels_leave_empty_spot = bool(moving_pixels_leave_empty_spot)
        self.reset_mode = reset_mode
        self.reset_random_factor = reset_random_factor
        self.reset_constant_step = reset_constant_step
        self.reset_linear_factor = reset_linear_factor
        self.reset_source = bool(reset_source)


def base_indices(h: int, w: int) -> np.ndarray:
    """DataLayer.base, data.py:15: [H,W,2] int32 (i, j)."""
    ii, jj = np.meshgrid(np.arange(h, dtype=np.int32),
                         np.arange(w, dtype=np.int32), indexing="ij")
    return np.stack([ii, jj], axis=-1)


def source_index_map(h: int, w: int, introduction_masks) -> np.ndarray:
    """reference.py:46-52: highest source whose introduction mask is set, else 0."""
    src = np.zeros((h, w), dtype=np.int32)
    for s, m in enumerate(introduction_masks):
        src[np.asarray(m, dtype=bool)] = s
    return src


def init_data(h: int, w: int, introduction_masks=()) -> np.ndarray:
    """ReferenceLayer.__init__, reference.py:38-44 (+ set_sources :54-56)."""
    data = np.zeros((h, w, 4), dtype=np.int32)
    data[:, :, 0:2] = base_indices(h, w)
    data[:, :, INDEX_ALPHA] = 1
    data[:, :, INDEX_SOURCE] = source_index_map(h, w, introduction_masks)
    return data


# --------------------------------------------------------------------------
# B2 + B3  MovementLayer.update        compositor/layers/movement.py:20-64
# --------------------------------------------------------------------------
def flow_to_offsets(flow: np.ndarray) -> np.ndarray:
    """movement.py:20-23: round half-even, flat offset fy*W + fx (int32)."""
    h, w, _ = flow.shape
    fi = np.rint(flow).astype(np.int32)
    return (fi[:, :, 1] * w + fi[:, :, 0]).ravel()


def move(data: np.ndarray, flow: np.ndarray, mask_src: np.ndarray,
         mask_dst: np.ndarray, prm: LayerParams, alpha_index: int = INDEX_ALPHA) -> np.ndarray:
    """movement.py:25-60 as a per-target gather.  Returns the new data array.

    The reference indexes ``flat[shift]`` with python semantics (negative
    offsets wrap, offsets >= H*W raise IndexError, movement.py:33,39); flows
    that went through post_process never leave the frame.  Here an
    out-of-frame source raises IndexError for both cases' superset that the
    HIP path rejects: s < 0 or s >= H*W.
    """
    h, w, depth = data.shape
    n = h * w
    d = flow_to_offsets(flow).astype(np.int64)
    t = np.arange(n, dtype=np.int64)
    s = t + d
    if s.min(initial=0) < 0 or s.max(initial=0) >= n:
        raise IndexError("flow leaves the frame; run post_process first")
    old = data.reshape(n, depth)
    a_old = old[:, alpha_index]
    ms = np.asarray(mask_src, dtype=bool).ravel()[s]          # :39
    src_filled = a_old[s] != 0
    if not prm.transparent_pixels_can_move:                   # :35-38
        ms &= src_filled
    md = np.asarray(mask_dst, dtype=bool).ravel().copy()      # :41
    if not prm.pixels_can_move_to_empty_spot:                 # :42-43
        md &= a_old != 0
    if not prm.pixels_can_move_to_filled_spot:                # :44-45
        md &= a_old == 0
    in_t = (d != 0) & ms & md                                 # :47-48
    new = old.copy()
    new[in_t] = old[s[in_t]]                                  # :51-52
    if prm.moving_pixels_leave_empty_spot:                    # :53-54
        new[s[in_t], alpha_index] = 0
    if prm.transparent_pixels_can_move:                       # :55-58
        new[in_t & src_filled, alpha_index] = 1
    else:                                                     # :59-60
        new[in_t, alpha_index] = 1
    return new.reshape(h, w, depth)


# --------------------------------------------------------------------------
# B4  ReferenceLayer._update_reset_*   compositor/layers/reference.py:58-91
# --------------------------------------------------------------------------
def reset_random(data: np.ndarray, u: np.ndarray, reset_mask: np.ndarray,
                 prm: LayerParams, introduction_masks=()) -> None:
    """reference.py:58-67, in place.  ``u`` is the float64 uniform field the
    reference draws with numpy.random.random (:59); the threshold is
    ``reset_random_factor * reset_mask`` evaluated as numpy does for a python
    scalar times a float32 array (float32), compared in float64 (:61)."""
    h, w, _ = data.shape
    thr = prm.reset_random_factor * np.asarray(reset_mask, dtype=np.float32)
    sel = u < thr
    base = base_indices(h, w)
    data[:, :, INDEX_I][sel] = base[:, :, 0][sel]
    data[:, :, INDEX_J][sel] = base[:, :, 1][sel]
    data[:, :, INDEX_ALPHA][sel] = 1
    if prm.reset_source:                                      # :66-67
        for s, m in enumerate(introduction_masks):
            data[:, :, INDEX_SOURCE][np.asarray(m, dtype=bool) & sel] = s


def reset_constant(data: np.ndarray, reset_mask: np.ndarray, prm: LayerParams) -> None:
    """reference.py:69-79, in place."""
    h, w, _ = data.shape
    base = base_indices(h, w)
    dij_base = (base - data[:, :, 0:2]).astype(np.float32)    # :70
    dij = dij_base.copy()
    norm_base = np.max(np.abs(dij), axis=2)                   # inf-norm :72
    nz = norm_base != 0
    dij[nz] /= norm_base[..., None][nz]                       # :74
    dij *= prm.reset_constant_step * np.asarray(reset_mask, np.float32)[..., None]  # :75
    norm_scaled = np.max(np.abs(dij), axis=2)                 # :76
    over = norm_scaled > norm_base
    dij[over] = dij_base[over]                                # :78
    data[:, :, 0:2] += np.rint(dij).astype(np.int32)          # :79


def reset_linear(data: np.ndarray, reset_mask: np.ndarray, prm: LayerParams) -> None:
    """reference.py:81-83, in place (python float factor times int32: float64)."""
    h, w, _ = data.shape
    base = base_indices(h, w)
    dij = prm.reset_linear_factor * (base - data[:, :, 0:2])  # float64
    data[:, :, 0:2] += np.rint(
        np.asarray(reset_mask, np.float32)[..., None] * dij).astype(np.int32)


def reset(data, prm, reset_mask, u=None, introduction_masks=()):
    """Dispatch of reference.py:85-91."""
    if prm.reset_mode == "random":
        assert u is not None
        reset_random(data, u, reset_mask, prm, introduction_masks)
    elif prm.reset_mode == "constant":
        reset_constant(data, reset_mask, prm)
    elif prm.reset_mode == "linear":
        reset_linear(data, reset_mask, prm)
    elif prm.reset_mode != "off":
        raise ValueError(f"Unknown reset mode {prm.reset_mode}")


# --------------------------------------------------------------------------
# B5  ReferenceLayer._update_rgba      compositor/layers/reference.py:93-105
# --------------------------------------------------------------------------
def gather_rgba(rgba: np.ndarray, data: np.ndarray, source_index: int,
                pixmap: np.ndarray) -> None:
    """One iteration of the per-source loop (reference.py:94-105), in place."""
    h, w, _ = data.shape
    c = pixmap.shape[2]
    sel = (data[:, :, INDEX_SOURCE] == source_index) & (data[:, :, INDEX_ALPHA] != 0)
    mi = np.clip(data[:, :, 0], 0, h - 1)[sel]                # :100
    mj = np.clip(data[:, :, 1], 0, w - 1)[sel]                # :101
    rgba[:, :, :c][sel] = pixmap[mi, mj]                      # :102
    if c == 3:                                                # :103-105
        rgba[:, :, 3] = 0
        rgba[:, :, 3][sel] = 1


# --------------------------------------------------------------------------
# B6  Layer.render + Compositor.render layer.py:32-34, compositor.py:31-40
# --------------------------------------------------------------------------
def layer_render(rgba: np.ndarray, mask_alpha: np.ndarray) -> np.ndarray:
    """layer.py:32-34: alpha := uint8(mask_alpha * alpha) IN PLACE, returns copy."""
    rgba[:, :, 3] = np.asarray(mask_alpha, np.float32) * rgba[:, :, 3]
    return np.clip(rgba, 0, 255).astype(np.uint8)


def composite(background_rgb, layer_images) -> np.ndarray:
    """compositor.py:35-40: paint opaque pixels of each layer, in order."""
    image = np.array(background_rgb, dtype=np.uint8, copy=True)
    for li in layer_images:
        opaque = li[:, :, 3] != 0
        image[opaque] = li[:, :, :3][opaque]
    return image


# --------------------------------------------------------------------------
# A whole moveref layer, for sequence tests (move_reference.py:6-14)
# --------------------------------------------------------------------------
class MoveRefLayer:
    def __init__(self, h, w, prm: LayerParams | None = None, mask_src=None,
                 mask_dst=None, mask_alpha=None, reset_mask=None,
                 introduction_masks=()):
        self.h, self.w = h, w
        self.prm = prm or LayerParams()
        self.mask_src = np.ones((h, w), bool) if mask_src is None else np.asarray(mask_src, bool)
        self.mask_dst = np.ones((h, w), bool) if mask_dst is None else np.asarray(mask_dst, bool)
        self.mask_alpha = np.ones((h, w), np.float32) if mask_alpha is None else np.asarray(mask_alpha, np.float32)
        self.reset_mask = np.ones((h, w), np.float32) if reset_mask is None else np.asarray(reset_mask, np.float32)
        self.introduction_masks = list(introduction_masks)
        self.data = init_data(h, w, self.introduction_masks)
        self.rgba = np.zeros((h, w, 4), np.uint8)

    def update(self, flow, pixmaps=(), u=None):
        """move_reference.py:12-14: Movement.update then Reference.update."""
        self.data = move(self.data, flow, self.mask_src, self.mask_dst, self.prm)
        reset(self.data, self.prm, self.reset_mask, u, self.introduction_masks)
        for s, pm in enumerate(pixmaps):
            gather_rgba(self.rgba, self.data, s, pm)

    def render(self):
        return layer_render(self.rgba, self.mask_alpha)


# --------------------------------------------------------------------------
# N2  the other layer classes (layer.py:44-56)
# --------------------------------------------------------------------------
class SumLayer(MoveRefLayer):
    """compositor/layers/sum.py:7-14: a ReferenceLayer (no move) whose (i, j) accumulate
    floor(flow) -- flow channel 0 goes to i and channel 1 to j, as written (:10) -- followed
    by ReferenceLayer.update (reset, then the rgba gather: reference.py:107-109)."""

    def update(self, flow, pixmaps=(), u=None):
        self.data[:, :, 0:2] += np.floor(flow).astype(np.int32)    # sum.py:10
        reset(self.data, self.prm, self.reset_mask, u, self.introduction_masks)
        for s, pm in enumerate(pixmaps):
            gather_rgba(self.rgba, self.data, s, pm)


class StaticLayer:
    """compositor/layers/static.py:7-17: alpha starts at 1 (:11); every update copies each
    source's pixmap where its introduction mask is set (:14-17)."""

    def __init__(self, h, w, mask_alpha=None, introduction_masks=()):
        self.h, self.w = h, w
        self.mask_alpha = np.ones((h, w), np.float32) if mask_alpha is None else np.asarray(mask_alpha, np.float32)
        self.introduction_masks = [np.asarray(m, bool) for m in introduction_masks]
        self.rgba = np.zeros((h, w, 4), np.uint8)
        self.rgba[:, :, 3] = 1

    def update(self, flow, pixmaps=(), u=None):
        for m, pm in zip(self.introduction_masks, pixmaps):
            self.rgba[:, :, :pm.shape[2]][m] = pm[m]

    def render(self):
        return layer_render(self.rgba, self.mask_alpha)


INTRO_DEPTH = 8  # r, g, b, alpha, source, i, j, frame   (introduction.py:10-14)
INTRO_ALPHA = 3


class IntroParams(LayerParams):
    """LayerParams + the introduce_* fields of LayerConfig (config.py:99-105)."""

    def __init__(self, introduce_pixels_on_empty_spots=True, introduce_pixels_on_filled_spots=True,
                 introduce_moving_pixels=True, introduce_unmoving_pixels=True, introduce_once=False,
                 introduce_on_all_filled_spots=False, introduce_on_all_empty_spots=False, **kw):
        LayerParams.__init__(self, **kw)
        self.introduce_pixels_on_empty_spots = bool(introduce_pixels_on_empty_spots)
        self.introduce_pixels_on_filled_spots = bool(introduce_pixels_on_filled_spots)
        self.introduce_moving_pixels = bool(introduce_moving_pixels)
        self.introduce_unmoving_pixels = bool(introduce_unmoving_pixels)
        self.introduce_once = bool(introduce_once)
        self.introduce_on_all_filled_spots = bool(introduce_on_all_filled_spots)
        self.introduce_on_all_empty_spots = bool(introduce_on_all_empty_spots)


def introduction_mask(data: np.ndarray, offsets: np.ndarray, prm: IntroParams) -> np.ndarray:
    """introduction.py:24-44: which targets may receive an introduced pixel.

    As written in the reference, `where_empty` (:27) and the unmoving selection (:37) are the
    comparison `numpy.where(...) == 0` of a TUPLE with 0, i.e. the Python value False; indexing
    with False selects nothing, so introduce_pixels_on_empty_spots, introduce_unmoving_pixels and
    the mask write of introduce_on_all_empty_spots (:43) have no effect on the mask (the latter
    still switches `consider_flow` off, :40).  Pinned by tests/golden/layer2_intro_*.npz.
    """
    h, w, _ = data.shape
    filled = data[:, :, INTRO_ALPHA] != 0                          # :28
    mask = np.ones((h, w), dtype=bool)                             # :25
    if not prm.introduce_pixels_on_filled_spots:                   # :32-33
        mask[filled] = False
    if not prm.introduce_moving_pixels:                            # :34-35
        mask[(offsets != 0).reshape(h, w)] = False
    if prm.introduce_on_all_filled_spots:                          # :41-42
        mask[filled] = True
    return mask


def introduce(data: np.ndarray, mask: np.ndarray, offsets: np.ndarray, prm: IntroParams, source_index: int,
              pixmap: np.ndarray, intro_mask: np.ndarray, frame_number: int) -> None:
    """One iteration of the per-source loop, introduction.py:46-63, in place: target t takes the
    8-channel record (pixmap[s], [1 if RGB], source index, base(s), frame number) of
    s = t + offset(t) -- or of t itself when either introduce_on_all_* flag is set (:40, :50-53)."""
    h, w, _ = data.shape
    n = h * w
    sel = (mask & np.asarray(intro_mask, bool)).ravel()            # :48
    t = np.nonzero(sel)[0]
    consider_flow = not (prm.introduce_on_all_filled_spots or prm.introduce_on_all_empty_spots)
    s = t + offsets.astype(np.int64)[t] if consider_flow else t
    if s.size and (s.min() < 0 or s.max() >= n):
        raise IndexError("flow leaves the frame; run post_process first")
    c = pixmap.shape[2]
    rec = np.zeros((n, INTRO_DEPTH), dtype=np.int32)
    rec[:, :c] = pixmap.reshape(n, c)
    if c == 3:
        rec[:, 3] = 1                                              # :61-62
    rec[:, 4] = source_index
    rec[:, 5:7] = base_indices(h, w).reshape(n, 2)
    rec[:, 7] = frame_number
    flat = data.reshape(n, INTRO_DEPTH)
    flat[t] = rec[s]                                               # :63


class IntroductionLayer:
    """compositor/layers/introduction.py:8-73 (a MovementLayer with an 8-channel canvas)."""

    def __init__(self, h, w, prm: IntroParams | None = None, mask_src=None, mask_dst=None, mask_alpha=None,
                 introduction_masks=()):
        self.h, self.w = h, w
        self.prm = prm or IntroParams()
        self.mask_src = np.ones((h, w), bool) if mask_src is None else np.asarray(mask_src, bool)
        self.mask_dst = np.ones((h, w), bool) if mask_dst is None else np.asarray(mask_dst, bool)
        self.mask_alpha = np.ones((h, w), np.float32) if mask_alpha is None else np.asarray(mask_alpha, np.float32)
        self.introduction_masks = list(introduction_masks)
        self.data = np.zeros((h, w, INTRO_DEPTH), np.int32)        # data.py:17: an empty canvas
        self.introduced_once = False

    @property
    def rgba(self):
        return self.data[:, :, :4]                                 # :65-66: a VIEW of data (int32)

    def update(self, flow, pixmaps=(), frame_numbers=None, u=None):
        self.data = move(self.data, flow, self.mask_src, self.mask_dst, self.prm, alpha_index=INTRO_ALPHA)
        if self.prm.introduce_once and self.introduced_once:       # :21-22 (no source.next() either)
            return
        self.introduced_once = True
        offsets = flow_to_offsets(flow)
        mask = introduction_mask(self.data, offsets, self.prm)
        for s, pm in enumerate(pixmaps):
            fn = 0 if frame_numbers is None else frame_numbers[s]
            introduce(self.data, mask, offsets, self.prm, s, pm, self.introduction_masks[s], fn)

    def render(self):
        """layer.py:32-34 on the int32 view: alpha := int32(mask_alpha * alpha) written back into
        `data` (float32 times int32 is float64 in numpy; the store truncates), result clipped to u8."""
        a = self.mask_alpha.astype(np.float64) * self.data[:, :, INTRO_ALPHA]
        self.data[:, :, INTRO_ALPHA] = a.astype(np.int32)
        return np.clip(self.data[:, :, :4], 0, 255).astype(np.uint8)
