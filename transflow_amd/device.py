"""Raw device buffers (tf_dev_*): for harnesses that keep inputs resident in HBM."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import check


class DevBuffer:
    def __init__(self, nbytes: int):
        self._lib = _lib.load()
        self.nbytes = int(nbytes)
        p = C.c_void_p()
        check(self._lib.tf_dev_alloc(C.byref(p), self.nbytes))
        self.ptr = p.value

    @classmethod
    def from_array(cls, a: np.ndarray) -> "DevBuffer":
        a = np.ascontiguousarray(a)
        buf = cls(a.nbytes)
        buf.upload(a)
        return buf

    def upload(self, a: np.ndarray) -> None:
        a = np.ascontiguousarray(a)
        if a.nbytes > self.nbytes:
            raise ValueError("array larger than the device buffer")
        if a.nbytes:
            check(self._lib.tf_dev_upload(C.c_void_p(self.ptr), C.c_void_p(a.ctypes.data), a.nbytes))

    def download(self, shape, dtype, offset: int = 0) -> np.ndarray:
        """The bytes at `offset` as an array of `shape` / `dtype`."""
        out = np.empty(shape, dtype)
        if offset < 0 or offset + out.nbytes > self.nbytes:
            raise ValueError("requested bytes outside the device buffer")
        if out.nbytes:
            check(self._lib.tf_dev_download(C.c_void_p(out.ctypes.data), C.c_void_p(self.ptr + int(offset)), out.nbytes))
        return out

    def close(self):
        if getattr(self, "ptr", None):
            self._lib.tf_dev_free(C.c_void_p(self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Pinned:
    """One page-locked host allocation (tf_host_alloc); freed when the last array over it is gone."""

    def __init__(self, nbytes: int):
        self._lib = _lib.load()
        p = C.c_void_p()
        check(self._lib.tf_host_alloc(C.byref(p), int(nbytes)))
        self.ptr = p.value

    def __del__(self):
        try:
            if getattr(self, "ptr", None):
                self._lib.tf_host_free(C.c_void_p(self.ptr))
                self.ptr = None
        except Exception:
            pass


def pinned_empty(shape, dtype) -> np.ndarray:
    """numpy.empty over page-locked memory: copies to and from the device run at the link's rate and asynchronously
    (transflow/output/ffmpeg.py:32-54 writes such a frame to the encoder; cv.py:490's flow is one).  The allocation
    lives as long as any array or view over it."""
    dtype = np.dtype(dtype)
    n = int(np.prod(shape)) * dtype.itemsize
    own = _Pinned(max(1, n))
    buf = (C.c_char * max(1, n)).from_address(own.ptr)
    buf._owner = own                      # the array's base keeps the ctypes view, the view keeps the allocation
    return np.frombuffer(buf, dtype=dtype, count=int(np.prod(shape))).reshape(shape)


class ArrayPool:
    """Host arrays handed out again once nobody else holds them.  A fresh 66 MB numpy array costs
    ~5 ms of page faults at 4K -- four times the copy that fills it -- so per-frame outputs (flows,
    rendered frames) come from a small pool; an array the caller still references is never reused.
    `pinned`: the arrays are page-locked (pinned_empty)."""

    def __init__(self, shape, dtype, limit: int = 4, pinned: bool = False):
        self.shape, self.dtype, self.limit = tuple(shape), np.dtype(dtype), int(limit)
        self.pinned = bool(pinned)
        self._arrays: list = []

    def take(self) -> np.ndarray:
        import sys
        for a, root in self._arrays:
            # nobody but the pool holds `a` or any view of its memory.  References counted: the pool's tuple, the loop
            # variable, getrefcount's argument -- and for the owner of a page-locked array's memory also `a.base`.
            if sys.getrefcount(a) == 3 and (root is None or sys.getrefcount(root) == 4):
                return a
        a = pinned_empty(self.shape, self.dtype) if self.pinned else np.empty(self.shape, self.dtype)
        # A page-locked array is itself a view (frombuffer -> reshape), and numpy points every view of a view at the
        # array that owns the memory: a caller's `flow[..., 0]` references that hidden 1-D array, not `a`.  The pool
        # watches both.
        root = a.base if isinstance(a.base, np.ndarray) else None
        if len(self._arrays) < self.limit:
            self._arrays.append((a, root))
        return a


def sync() -> None:
    check(_lib.load().tf_sync())
