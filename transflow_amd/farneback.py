"""Farnebäck handle: thin object over the tf_fb_* entry points of libtfhip.so.

`Farneback.calc(prev, next)` has the argument meaning of
cv2.calcOpticalFlowFarneback as the reference calls it
(transflow/flow/sources/cv.py:479-490).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import TfFbParams, TfFlowOp, check

OPTFLOW_USE_INITIAL_FLOW = 4      # cv2.OPTFLOW_USE_INITIAL_FLOW
OPTFLOW_FARNEBACK_GAUSSIAN = 256  # cv2.OPTFLOW_FARNEBACK_GAUSSIAN

FORWARD = 0   # FlowSource.Direction.FORWARD  (flow/sources/source.py:21)
BACKWARD = 1  # FlowSource.Direction.BACKWARD (flow/sources/source.py:22)


def _ptr(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


class Farneback:
    def __init__(self, width: int, height: int, pyr_scale: float = 0.5, levels: int = 3, winsize: int = 15,
                 iterations: int = 3, poly_n: int = 5, poly_sigma: float = 1.2, flags: int = 0,
                 frame_slots: int = 2, max_pairs: int = 1, device: int | None = None, lanes: int = 1,
                 exact: bool | None = None):
        """lanes = 2: calc_slots calls go alternately to two handles that share the frame slots and queue on the
        library's two call streams (tf_fb_create_lane), so consecutive batches are in flight together; results
        (get_flow, flow_ptr, post_process ...) are those of the latest call.  For callers that issue batch after
        batch without reading each back first (the resident path); not with keep_expansions.
        exact: True / False = this handle's calls sum the box window in OpenCV's own order (flows bit-identical to the
        CPU path's) or the default way, whatever other handles of the process do (tf_fb_set_exact); None = as the
        process-wide option fb_exact_sums says at each call."""
        if lanes not in (1, 2):
            raise ValueError("lanes must be 1 or 2")
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self._h2 = C.c_void_p()
        if device is not None:
            check(self._lib.tf_init(int(device)))
        self.width, self.height = int(width), int(height)
        self.frame_slots, self.max_pairs = int(frame_slots), int(max_pairs)
        self.flags = int(flags)
        prm = TfFbParams(float(pyr_scale), int(levels), int(winsize), int(iterations), int(poly_n),
                         float(poly_sigma), int(flags))
        check(self._lib.tf_fb_create(C.byref(self._h), self.width, self.height, C.byref(prm),
                                     self.frame_slots, self.max_pairs))
        self._handles = [self._h]
        if lanes == 2:
            check(self._lib.tf_fb_create_lane(C.byref(self._h2), self._h))
            self._handles.append(self._h2)
        self._calls = 0
        self._last = self._h   # the handle whose results the reading methods return
        if exact is not None:
            self.set_exact(exact)

    def set_exact(self, exact: bool | None) -> None:
        """tf_fb_set_exact on every lane: takes effect with the next call."""
        for h in self._handles:
            check(self._lib.tf_fb_set_exact(h, -1 if exact is None else int(bool(exact))))

    @property
    def _next(self):
        return self._handles[self._calls % len(self._handles)]

    def close(self):
        if getattr(self, "_h2", None) is not None and self._h2.value:
            self._lib.tf_fb_destroy(self._h2)
            self._h2 = C.c_void_p()
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.tf_fb_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _grey(self, frame) -> np.ndarray:
        a = np.asarray(frame)
        if a.dtype != np.uint8 or a.shape != (self.height, self.width):
            raise ValueError(f"expected uint8 grey frame {(self.height, self.width)}, got {a.dtype} {a.shape}")
        if a.strides[1] != 1 or a.strides[0] < a.shape[1]:
            a = np.ascontiguousarray(a)
        return a

    # -- one pair, host in / host out -------------------------------------------------
    def calc(self, prev, nxt, flow=None) -> np.ndarray:
        """`flow`: the initial flow, read when the handle has OPTFLOW_USE_INITIAL_FLOW (zeros if None, as
        cv.py:478 passes before the first frame); never modified, a new array is returned."""
        p, n = self._grey(prev), self._grey(nxt)
        if self.flags & OPTFLOW_USE_INITIAL_FLOW:
            if flow is None:
                flow = np.zeros((self.height, self.width, 2), np.float32)
            else:
                flow = np.array(flow, dtype=np.float32, order="C", copy=True)
                if flow.shape != (self.height, self.width, 2):
                    raise ValueError(f"initial flow shape {flow.shape} != {(self.height, self.width, 2)}")
        else:
            flow = np.empty((self.height, self.width, 2), np.float32)
        check(self._lib.tf_fb_calc(self._h, _ptr(p), p.strides[0], _ptr(n), n.strides[0], _ptr(flow)))
        self._last = self._h
        return flow

    # -- resident path ----------------------------------------------------------------
    def set_frame(self, slot: int, frame) -> None:
        a = self._grey(frame)
        check(self._lib.tf_fb_set_frame(self._h, int(slot), _ptr(a), a.strides[0]))

    def set_frame_bgr(self, slot: int, frame) -> None:
        """cv.py:461-466 on the device: a decoded BGR frame of any size -> nearest-neighbour resize to the
        handle's size -> grey, straight into the slot."""
        a = np.asarray(frame)
        if a.dtype != np.uint8 or a.ndim != 3 or a.shape[2] != 3 or a.shape[0] < 1 or a.shape[1] < 1:
            raise ValueError(f"expected a uint8 BGR frame (H, W, 3), got {a.dtype} {a.shape}")
        if a.strides[2] != 1 or a.strides[1] != 3 or a.strides[0] < 3 * a.shape[1]:
            a = np.ascontiguousarray(a)   # (a vertically flipped view, frame[::-1], or a broadcast row: rows that do not advance)
        check(self._lib.tf_fb_set_frame_bgr(self._h, int(slot), _ptr(a), a.shape[1], a.shape[0], a.strides[0]))

    def set_initial_flow(self, pair: int, flow) -> None:
        """Resident path, OPTFLOW_USE_INITIAL_FLOW: the initial flow of `pair` for the next calc_slots."""
        f = np.ascontiguousarray(flow, dtype=np.float32)
        if f.shape != (self.height, self.width, 2):
            raise ValueError(f"initial flow shape {f.shape} != {(self.height, self.width, 2)}")
        check(self._lib.tf_fb_set_initial_flow(self._next, int(pair), _ptr(f)))

    def initial_flow_ptr(self, pair: int) -> int:
        p = C.c_void_p()
        check(self._lib.tf_fb_initial_flow_ptr(self._next, int(pair), C.byref(p)))
        return p.value

    def stage_initial_flow(self, flow) -> np.ndarray:
        f = np.ascontiguousarray(flow, dtype=np.float32)
        w, h = self.level_sizes()[-1]
        out = np.empty((h, w, 2), np.float32)
        check(self._lib.tf_fb_stage_initial_flow(self._h, _ptr(f), _ptr(out)))
        return out

    def frame_ptr(self, slot: int) -> int:
        p = C.c_void_p()
        check(self._lib.tf_fb_frame_ptr(self._h, int(slot), C.byref(p)))
        return p.value

    def calc_slots(self, prev_slots, next_slots) -> None:
        n = len(prev_slots)
        if n != len(next_slots):
            raise ValueError("prev_slots and next_slots differ in length")
        a = (C.c_int * n)(*[int(v) for v in prev_slots])
        b = (C.c_int * n)(*[int(v) for v in next_slots])
        h = self._next
        check(self._lib.tf_fb_calc_slots(h, n, a, b))
        self._calls += 1
        self._last = h

    def get_flow(self, pair: int = 0) -> np.ndarray:
        flow = np.empty((self.height, self.width, 2), np.float32)
        check(self._lib.tf_fb_get_flow(self._last, int(pair), _ptr(flow)))
        return flow

    def get_flow_into(self, pair: int, out: np.ndarray) -> np.ndarray:
        if out.dtype != np.float32 or not out.flags.c_contiguous or out.shape != (self.height, self.width, 2):
            raise ValueError("get_flow_into needs a C-contiguous float32 array of shape (H, W, 2)")
        check(self._lib.tf_fb_get_flow(self._last, int(pair), _ptr(out)))
        return out

    def async_io(self, on: bool = True) -> None:
        """Frames go up and flows come down on copy streams of the library's (tf_fb_async_io): for a caller that streams
        one frame per call and reads each flow back."""
        check(self._lib.tf_fb_async_io(self._h, int(bool(on))))

    def get_flow_begin(self, pair: int, out: np.ndarray) -> int:
        """Start the download of the last call's flow into `out` (page-locked, C-contiguous float32 (H, W, 2)) beside
        whatever is queued next; returns the token for get_flow_end.  `out` must not be touched in between."""
        if out.dtype != np.float32 or not out.flags.c_contiguous or out.shape != (self.height, self.width, 2):
            raise ValueError("get_flow_begin needs a C-contiguous float32 array of shape (H, W, 2)")
        tok = C.c_int()
        check(self._lib.tf_fb_get_flow_begin(self._last, int(pair), _ptr(out), C.byref(tok)))
        return tok.value

    def get_flow_end(self, token: int) -> None:
        check(self._lib.tf_fb_get_flow_end(self._last, int(token)))

    def flow_ptr(self, pair: int = 0) -> int:
        p = C.c_void_p()
        check(self._lib.tf_fb_flow_ptr(self._last, int(pair), C.byref(p)))
        return p.value

    def post_process(self, pair: int, direction: int) -> None:
        check(self._lib.tf_fb_post_process(self._last, int(pair), int(direction)))

    def keep_expansions(self, on: bool = True) -> None:
        """Streaming: a slot's pyramid and polynomial expansion stay valid until set_frame writes it."""
        check(self._lib.tf_fb_keep_expansions(self._h, int(bool(on))))

    def post_process_scatter(self, pair: int) -> int:
        """First half of FORWARD post_process: device address of the int32 [H, W] winner map
        (RemapLayer.step_dev(..., clip_flow=2) does the rest)."""
        p = C.c_void_p()
        check(self._lib.tf_fb_post_process_scatter(self._last, int(pair), C.byref(p)))
        return p.value

    def post_process_host(self, flow: np.ndarray, direction: int) -> np.ndarray:
        """In place on a float32 C-contiguous [H,W,2] array, like the reference."""
        if (not isinstance(flow, np.ndarray) or flow.dtype != np.float32 or not flow.flags.c_contiguous
                or flow.shape != (self.height, self.width, 2)):
            raise ValueError("post_process needs a C-contiguous float32 array of shape (H, W, 2)")
        check(self._lib.tf_fb_post_process_host(self._h, _ptr(flow), int(direction)))
        return flow

    FLOW_OPS = {"scale": 0, "threshold": 1, "clip": 2}

    @staticmethod
    def _ops_array(ops):
        """ops: iterable of (name, value); value typed as the filter's lambda returned it."""
        items = []
        for name, value in ops:
            if isinstance(value, np.generic):
                if not np.issubdtype(value.dtype, np.floating) and not np.issubdtype(value.dtype, np.integer):
                    raise NotImplementedError(f"flow filter value of type {value.dtype}")
                wide = int(value.dtype == np.float64)          # strong float64 scalar: float64 arithmetic
            elif isinstance(value, (int, float)):
                wide = 0                                        # weak Python scalar: float32 arithmetic
            else:
                raise NotImplementedError(f"flow filter value of type {type(value).__name__} (array-valued "
                                          "expressions are not supported on the device)")
            items.append(TfFlowOp(Farneback.FLOW_OPS[name], wide, float(value)))
        arr = (TfFlowOp * max(1, len(items)))(*items)
        return arr, len(items)

    def post_process_host_ex(self, flow: np.ndarray, direction: int | None, ops=(), mask=None) -> np.ndarray:
        """Filters + mask + (direction handling unless direction is None), in place."""
        if (not isinstance(flow, np.ndarray) or flow.dtype != np.float32 or not flow.flags.c_contiguous
                or flow.shape != (self.height, self.width, 2)):
            raise ValueError("post_process needs a C-contiguous float32 array of shape (H, W, 2)")
        arr, n = self._ops_array(ops)
        m = None
        if mask is not None:
            m = np.ascontiguousarray(mask, dtype=np.float32).reshape(self.height, self.width)
        check(self._lib.tf_fb_post_process_host_ex(self._h, _ptr(flow), -1 if direction is None else int(direction), n,
                                                   arr, None if m is None else _ptr(m)))
        return flow

    def post_process_ex(self, pair: int, direction: int, ops=(), mask_dev: int | None = None) -> None:
        arr, n = self._ops_array(ops)
        check(self._lib.tf_fb_post_process_ex(self._last, int(pair), int(direction), n, arr,
                                              C.c_void_p(mask_dev) if mask_dev else None))

    # -- geometry / stage entry points (parity tests) -----------------------------------
    def level_sizes(self):
        n = C.c_int()
        check(self._lib.tf_fb_level_count(self._h, C.byref(n)))
        out = []
        for k in range(n.value):
            w, h = C.c_int(), C.c_int()
            check(self._lib.tf_fb_level_size(self._h, k, C.byref(w), C.byref(h)))
            out.append((w.value, h.value))
        return out

    def stage_level_image(self, frame, level: int) -> np.ndarray:
        a = self._grey(frame)
        w, h = self.level_sizes()[level]
        out = np.empty((h, w), np.float32)
        check(self._lib.tf_fb_stage_level_image(self._h, _ptr(a), a.strides[0], int(level), _ptr(out)))
        return out

    def stage_level_polyexp(self, frame, level: int) -> np.ndarray:
        a = self._grey(frame)
        w, h = self.level_sizes()[level]
        out = np.empty((h, w, 5), np.float32)
        check(self._lib.tf_fb_stage_level_polyexp(self._h, _ptr(a), a.strides[0], int(level), _ptr(out)))
        return out

    def stage_polyexp(self, img) -> np.ndarray:
        img = np.ascontiguousarray(img, np.float32)
        h, w = img.shape
        out = np.empty((h, w, 5), np.float32)
        check(self._lib.tf_fb_stage_polyexp(self._h, _ptr(img), w, h, _ptr(out)))
        return out

    def stage_update_matrices(self, r0, r1, flow) -> np.ndarray:
        r0 = np.ascontiguousarray(r0, np.float32)
        r1 = np.ascontiguousarray(r1, np.float32)
        flow = np.ascontiguousarray(flow, np.float32)
        h, w, _ = flow.shape
        out = np.empty((h, w, 5), np.float32)
        check(self._lib.tf_fb_stage_update_matrices(self._h, _ptr(r0), _ptr(r1), _ptr(flow), w, h, _ptr(out)))
        return out

    def stage_upsampled_matrices(self, level: int, r0, r1, coarse_flow) -> np.ndarray:
        """A5 + A3 at `level`: the flow of level + 1 upsampled inside the matrix kernel."""
        sizes = self.level_sizes()
        (w, h), (wc, hc) = sizes[level], sizes[level + 1]
        r0 = np.ascontiguousarray(r0, np.float32)
        r1 = np.ascontiguousarray(r1, np.float32)
        cf = np.ascontiguousarray(coarse_flow, np.float32)
        if r0.shape != (h, w, 5) or r1.shape != (h, w, 5) or cf.shape != (hc, wc, 2):
            raise ValueError("stage_upsampled_matrices: array shapes do not match the handle's levels")
        out = np.empty((h, w, 5), np.float32)
        check(self._lib.tf_fb_stage_upsampled_matrices(self._h, int(level), _ptr(r0), _ptr(r1), _ptr(cf), _ptr(out)))
        return out

    def stage_blur_solve(self, m) -> np.ndarray:
        m = np.ascontiguousarray(m, np.float32)
        h, w, _ = m.shape
        out = np.empty((h, w, 2), np.float32)
        check(self._lib.tf_fb_stage_blur_solve(self._h, _ptr(m), w, h, _ptr(out)))
        return out
