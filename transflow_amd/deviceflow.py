"""A flow that stays in HBM between the flow source and the compositor.

The reference hands every flow from `FlowSource.__next__` to `Compositor.update` as a host array -- through a
multiprocessing queue when the source runs in its child process (transflow/pipeline.py:85-86, 326), directly otherwise
(pipeline.py:562-567).  With a GPU on both sides of that seam the array is 66 MB per 4K frame down the link and up again.
`DeviceFlow` is what `HipFlowSource` yields instead when its configuration says `hip_device_flows`:

* it looks like the float32 (H, W, 2) array it stands for -- `shape`, `dtype`, indexing, arithmetic, `numpy.asarray(flow)`,
  anything numpy does with an object that has `__array__` -- and comes down (once, into a page-locked array) only when
  something actually reads it on the host; readers get read-only views, writing goes through the flow itself
  (`flow[...] = v`, `flow *= 2`) and marks the device copy stale;
* `HipCompositor.update` (every layer class) takes its device address: no transfer at all, the consumer's stream waits on
  the device for the event the producer recorded behind the flow's last kernel;
* `pickle.dumps(flow)` is the pickle of the host array (a checkpoint never holds a device address);
* through a `multiprocessing` queue -- whose pickler is `multiprocessing.reduction.ForkingPickler`, and only that one -- it
  travels as a 64-byte HIP IPC handle where `hip_device_flows = "ipc"` asked for it and the exporter can make one; the
  consumer's process opens the allocation, takes a private copy of the flow (device to device, ~30 us at 4K) and is done
  with the producer's buffer before `queue.get()` returns, so the producer's small ring of buffers never waits for a
  consumer.  Anything else (plain `hip_device_flows`, an export that fails) crosses a process boundary as the host array.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np
from numpy.lib.mixins import NDArrayOperatorsMixin

from . import _lib
from ._lib import check


class _Event:
    """tf_event: recorded on the calling thread's library stream; other streams / the host wait for it."""

    def __init__(self):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        check(self._lib.tf_event_create(C.byref(self._h)))

    def record(self):
        check(self._lib.tf_event_record(self._h))

    def stream_wait(self):
        check(self._lib.tf_stream_wait_event(self._h))

    def synchronize(self):
        check(self._lib.tf_event_synchronize(self._h))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.tf_event_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class _Slot:
    """One device buffer of a flow ring: the allocation, the event behind the flow that was last written into it
    (`ready`) and the event behind the last kernel that read it (`used`; None until somebody did)."""

    def __init__(self, nbytes: int, index: int):
        from .device import DevBuffer
        self.buf = DevBuffer(nbytes)
        self.index = index
        self.ready = _Event()
        self.used = None
        self.ipc_handle = None          # exported once, on first use
        self.exported_at = None         # the ring's export count when a flow in this buffer last left as an IPC token

    def close(self):
        self.buf.close()
        self.ready.close()
        if self.used is not None:
            self.used.close()


class FlowRing:
    """The device buffers a flow source's DeviceFlows live in.  A buffer goes back into rotation when the DeviceFlow over
    it is garbage; the producer's stream then waits (on the device) for the last kernel that read it.  A flow that left
    as an IPC token is garbage here as soon as it is pickled, long before the other process has read it (it copies the flow
    out inside its queue.get(), see the module text): such a buffer stays out of rotation until EXPORT_HOLD more flows
    have left the same way -- a producer can only be that far ahead of its consumer through a queue deeper than
    EXPORT_HOLD - 2 (transflow/pipeline.py:326: maxsize = 1)."""

    EXPORT_HOLD = 8

    def __init__(self, shape, slots: int = 4):
        self.shape = tuple(int(v) for v in shape)
        self.nbytes = int(np.prod(self.shape)) * 4
        self.slots = max(2, int(slots))
        self._all: list[_Slot] = []
        self._free: list[_Slot] = []
        self.exports = 0

    def exported(self, slot: "_Slot") -> None:
        slot.exported_at = self.exports
        self.exports += 1

    def take(self) -> _Slot:
        slot = next((s for s in self._free
                     if s.exported_at is None or self.exports - s.exported_at >= self.EXPORT_HOLD), None)
        if slot is not None:
            self._free.remove(slot)
        else:
            slot = _Slot(self.nbytes, len(self._all))
            self._all.append(slot)           # (more flows held -- or on their way to another process -- than `slots`: the ring grows)
        if slot.used is not None:
            slot.used.stream_wait()          # the producer's writes stay behind the consumer's last read
        return slot

    def give_back(self, slot: _Slot) -> None:
        if slot in self._all and slot not in self._free:
            self._free.append(slot)

    def close(self):
        for s in self._all:
            s.close()
        self._all, self._free = [], []


class DeviceFlow(NDArrayOperatorsMixin):
    """float32 (H, W, 2) flow in HBM; see the module text."""

    dtype = np.dtype(np.float32)
    ndim = 3
    __array_priority__ = 0.0
    _HANDLED = (np.ndarray, np.generic, int, float, complex, bool, list, tuple)

    def __init__(self, shape, dev_ptr: int, ready: _Event | None, *, ring: FlowRing | None = None, slot=None, owner=None,
                 cross_process: str | None = None):
        self.shape = tuple(int(v) for v in shape)
        self._ptr = int(dev_ptr)
        self._ready = ready
        self._ring, self._slot, self._owner = ring, slot, owner
        self._host = None
        self._dirty = False                # the host values were modified in place: the device copy is stale
        self._cross = cross_process        # "ipc": a multiprocessing queue carries the IPC handle, not the array
        self.in_frame = False              # set by a source whose post_process clipped the flow on the device: no rounded
                                           # vector of it can leave the frame (the compositor need not look for one)

    # ---- what the compositor uses -------------------------------------------------------------------------------
    @property
    def dev_ptr(self) -> int:
        return self._ptr

    def wait_on_stream(self) -> None:
        """The calling thread's library stream waits, on the device, for the flow to be complete."""
        if self._ready is not None:
            self._ready.stream_wait()

    def mark_used(self) -> None:
        """Call after queueing the last kernel that reads the flow: the buffer's next writer waits for it."""
        if self._slot is not None:
            if self._slot.used is None:
                self._slot.used = _Event()
            self._slot.used.record()

    # ---- the array it stands for --------------------------------------------------------------------------------
    @property
    def size(self) -> int:
        return int(np.prod(self.shape))

    @property
    def nbytes(self) -> int:
        return self.size * 4

    def __len__(self) -> int:
        return self.shape[0]

    def host(self) -> np.ndarray:
        """The flow's values on the host: downloaded on first use (page-locked memory), the same array afterwards."""
        if self._host is None:
            from .device import pinned_empty
            out = pinned_empty(self.shape, np.float32)
            self.wait_on_stream()
            check(_lib.load().tf_dev_download(C.c_void_p(out.ctypes.data), C.c_void_p(self._ptr), out.nbytes))
            self._host = out
        return self._host

    def _read(self) -> np.ndarray:
        """What readers get: a READ-ONLY view of the host values, so that looking at a flow (saving it, rendering it,
        numpy.asarray) leaves the device copy the current one.  Writing goes through the flow itself -- `flow[...] = v`,
        `flow *= 2`, `numpy.clip(flow, a, b, out=flow)` -- which marks the device copy stale: the compositor then takes
        the host values, as the reference would have."""
        if self._dirty:
            return self.host()
        v = self.host().view()
        v.flags.writeable = False
        return v

    def _written(self) -> np.ndarray:
        self._dirty = True
        return self.host()

    @property
    def on_host(self) -> bool:
        """True once the host values differ (or may differ) from the device copy."""
        return self._dirty

    def __array__(self, dtype=None, copy=None):
        a = self._read()
        if dtype is not None and np.dtype(dtype) != a.dtype:
            return a.astype(dtype)
        return a.copy() if copy else a

    def __array_ufunc__(self, ufunc, method, *inputs, out=None, **kwargs):
        args = [x._read() if isinstance(x, DeviceFlow) else x for x in inputs]
        if out is None:
            return getattr(ufunc, method)(*args, **kwargs)
        # `flow *= 2`, numpy.clip(flow, a, b, out=flow): the host values change in place and the flow stays a flow
        res = getattr(ufunc, method)(*args, out=tuple(o._written() if isinstance(o, DeviceFlow) else o for o in out), **kwargs)
        if isinstance(res, tuple):
            return tuple(o if isinstance(o, DeviceFlow) else r for o, r in zip(out, res))
        return out[0] if isinstance(out[0], DeviceFlow) else res

    def __array_function__(self, func, types, args, kwargs):
        def down(x):
            if isinstance(x, DeviceFlow):
                return x._read()
            if isinstance(x, (list, tuple)):
                return type(x)(down(v) for v in x)
            return x
        return func(*down(args), **{k: down(v) for k, v in kwargs.items()})

    def __getitem__(self, key):
        return self._read()[key]

    def __setitem__(self, key, value):
        self._written()[key] = value

    def __iter__(self):
        return iter(self._read())

    def __getattr__(self, name):
        # anything else an ndarray has (copy, astype, reshape, T, min, tobytes ...): the host array's
        if name.startswith("_"):
            raise AttributeError(name)
        return getattr(self._read(), name)

    def __repr__(self):
        where = "modified on the host" if self._dirty else ("read on the host" if self._host is not None else "on the device")
        return f"DeviceFlow(shape={self.shape}, float32, {where})"

    # ---- pickling -----------------------------------------------------------------------------------------------
    def __reduce__(self):
        # checkpoints, copy.deepcopy, any ordinary pickler: the host array and nothing else
        return (np.array, (self.host(),))

    def __del__(self):
        try:
            if self._ring is not None and self._slot is not None:
                self._ring.give_back(self._slot)
        except Exception:
            pass


# ---- across a process boundary -----------------------------------------------------------------------------------
_OPENED: dict = {}      # (producer pid, slot index, handle bytes) -> mapped device address in THIS process


def _reduce_for_queue(flow: DeviceFlow):
    """ForkingPickler's reducer (multiprocessing queues and pipes only).  Runs in the queue's feeder thread."""
    if flow._cross == "ipc" and flow._slot is not None and flow._host is None:
        try:
            slot = flow._slot
            if slot.ipc_handle is None:
                h = (C.c_char * 64)()
                check(_lib.load().tf_ipc_export(C.c_void_p(slot.buf.ptr), h))
                slot.ipc_handle = bytes(h.raw)
            if flow._ready is not None:
                flow._ready.synchronize()            # the flow is complete before another process may read it
            if flow._ring is not None:
                flow._ring.exported(slot)            # ... and its buffer stays untouched until that process has had time to
            return (_open_from_queue, (slot.ipc_handle, os.getpid(), slot.index, flow.shape, flow.in_frame))
        except Exception:                            # no IPC on this system: the array crosses instead
            flow._cross = None
    return (np.array, (flow.host(),))


def _open_from_queue(handle: bytes, pid: int, index: int, shape, in_frame: bool = False):
    """In the consumer's process: map the producer's buffer (once per buffer), copy the flow out of it into memory of
    our own and hand that out -- the producer's buffer is free again when queue.get() returns."""
    lib = _lib.load()
    key = (pid, index, handle)
    src = _OPENED.get(key)
    if src is None:
        p = C.c_void_p()
        check(lib.tf_ipc_open(handle, C.byref(p)))
        src = _OPENED[key] = p.value
    ring = _CONSUMER_RINGS.setdefault(tuple(shape), FlowRing(shape, slots=3))
    slot = ring.take()
    check(lib.tf_dev_copy(C.c_void_p(slot.buf.ptr), C.c_void_p(src), ring.nbytes))
    check(lib.tf_sync())                             # done with the producer's memory
    slot.ready.record()
    flow = DeviceFlow(shape, slot.buf.ptr, slot.ready, ring=ring, slot=slot)
    flow.in_frame = bool(in_frame)
    return flow


_CONSUMER_RINGS: dict = {}


def _register():
    from multiprocessing.reduction import ForkingPickler
    ForkingPickler.register(DeviceFlow, _reduce_for_queue)


_register()
