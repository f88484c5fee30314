"""A rendered frame that is still on its way down.

`Compositor.render()` returns the finished frame as a host array (transflow/compositor/compositor.py:31-40), the main
loop hands it to the outputs' queue and goes on to the next `update` (pipeline.py:518, 565); the output process writes its
bytes to the encoder's pipe (output/ffmpeg.py:32-54).  With a synchronous download the 25 MB of a 4K frame cross the link
while nothing else of the consumer's runs -- in particular not the 25 MB upload of the next frame's pixmap, which uses
the link's other direction.  `HipCompositor(..., lazy_frames=True).render()` returns a `DeviceFrame` instead:

* the frame is rendered into one of two device images in turn and its download is STARTED (tf_comp_download_begin: the
  library's download stream, a page-locked array of the compositor's pool as the target) before `render()` returns;
* the object is the uint8 (H, W, 3) array it stands for to everything numpy -- `numpy.asarray`, indexing, arithmetic,
  `tobytes()`, any ndarray attribute -- and the first such use waits for the transfer to end (tf_comp_download_end);
  afterwards it IS that array: reads and writes go straight to it (a rendered frame has no device copy to keep current);
* any pickle of it -- a checkpoint, the outputs' multiprocessing queue (whose feeder thread then does the waiting, beside
  the main loop's next update) -- is the pickle of the host array.

Plain `ndarray` frames stay the default.
"""
from __future__ import annotations

import numpy as np
from numpy.lib.mixins import NDArrayOperatorsMixin


class DeviceFrame(NDArrayOperatorsMixin):
    """uint8 (H, W, 3) frame whose download may still be running; see the module text."""

    dtype = np.dtype(np.uint8)
    ndim = 3
    __array_priority__ = 0.0

    def __init__(self, target: np.ndarray, image):
        self.shape = tuple(target.shape)
        self._target = target          # page-locked; being filled until _image.download_end() has returned
        self._image = image            # the CompImage whose download was begun into `target`
        self._done = False

    @property
    def arrived(self) -> bool:
        """True once something has waited for the transfer (the frame's values are in host memory)."""
        return self._done

    def host(self) -> np.ndarray:
        if not self._done:
            self._image.download_end()
            self._done = True
            self._image = None
        return self._target

    @property
    def size(self) -> int:
        return int(np.prod(self.shape))

    @property
    def nbytes(self) -> int:
        return self.size

    def __len__(self) -> int:
        return self.shape[0]

    def __array__(self, dtype=None, copy=None):
        a = self.host()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        args = [x.host() if isinstance(x, DeviceFrame) else x for x in inputs]
        if out is None:
            return getattr(ufunc, method)(*args, **kwargs)
        res = getattr(ufunc, method)(*args, out=tuple(o.host() if isinstance(o, DeviceFrame) else o for o in out), **kwargs)
        if isinstance(res, tuple):
            return tuple(o if isinstance(o, DeviceFrame) else r for o, r in zip(out, res))
        return out[0] if isinstance(out[0], DeviceFrame) else res

    def __array_function__(self, func, types, args, kwargs):
        def down(x):
            if isinstance(x, DeviceFrame):
                return x.host()
            if isinstance(x, (list, tuple)):
                return type(x)(down(v) for v in x)
            return x
        return func(*down(args), **{k: down(v) for k, v in kwargs.items()})

    def __getitem__(self, key):
        return self.host()[key]

    def __setitem__(self, key, value):
        self.host()[key] = value

    def __iter__(self):
        return iter(self.host())

    def __getattr__(self, name):
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self.host(), name)          # copy, astype, tobytes, reshape, T, mean ...: the host array's

    def __repr__(self):
        return f"DeviceFrame(shape={self.shape}, uint8, {'in host memory' if self._done else 'on its way down'})"

    def __reduce__(self):
        return (np.array, (self.host(),))           # checkpoints, queues, deepcopy: the array and nothing else

    def __del__(self):
        # the page-locked target goes back to its pool when this object dies: not before the transfer into it has ended
        try:
            if not self._done and self._image is not None:
                self._image.download_end()
        except Exception:
            pass
