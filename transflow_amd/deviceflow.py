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

What keeps an exported buffer valid until the consumer has copied it (round 6; rounds 1-5 counted exports and hoped):

* every ring buffer starts with a 64-bit GENERATION word (HEADER bytes in front of the flow).  The producer bumps it in
  stream order (tf_dev_store_u64) before the buffer is written again; the token carries the generation the flow was
  written under; the consumer reads the word AFTER its private copy and raises RuntimeError naming the buffer if it has
  moved on -- a flow that was overwritten under the consumer's hands is an error, never a wrong frame;
* the consumer ACKNOWLEDGES every token it has copied in a small file both processes map (_AckBoard; its path travels in
  the token).  A buffer that left as a token comes back into the producer's rotation only once its generation is
  acknowledged (the ring grows meanwhile), and `FlowRing.drain()` -- what `HipFlowSource.close()` calls before the source
  lets go of its buffers -- waits, with a time limit, until nothing is on its way any more: multiprocessing's
  `Queue.get()` frees the queue's slot BEFORE it unpickles, so the producer's last `put()` can return, and its process
  end, while the consumer has not yet opened the last flow's handle;
* a consumer closes its mapping of a producer's buffer (tf_ipc_close) once that producer's process is gone.
"""
from __future__ import annotations

import atexit
import ctypes as C
import mmap
import os
import tempfile
import threading
import time

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


HEADER = 256     # bytes in front of the flow in a ring buffer; the first eight are its generation word


class _AckBoard:
    """Producer side of the acknowledgements: a page of 64-bit words, one per ring buffer, in a file both processes map
    (in /dev/shm where there is one).  Word i = the highest generation of buffer i a consumer has copied."""

    WORDS = 512

    def __init__(self):
        d = "/dev/shm" if os.path.isdir("/dev/shm") and os.access("/dev/shm", os.W_OK) else None
        fd, self.path = tempfile.mkstemp(prefix=f"tfhip-ack-{os.getpid()}-", dir=d)
        try:
            os.ftruncate(fd, 8 * self.WORDS)
            self._mm = mmap.mmap(fd, 8 * self.WORDS)
        finally:
            os.close(fd)
        self._words = np.frombuffer(self._mm, dtype=np.uint64)
        atexit.register(self.close)      # the file does not outlive the process (close() is idempotent)

    def acked(self, index: int) -> int:
        return int(self._words[index]) if self._words is not None else 0

    def close(self):
        self._words = None
        try:
            self._mm.close()
        except (BufferError, ValueError):
            pass
        try:
            os.unlink(self.path)
        except OSError:
            pass
        try:
            atexit.unregister(self.close)
        except Exception:
            pass


_ACK_VIEWS: dict = {}     # consumer side: path -> (mmap, words)


def _acknowledge(path: str, index: int, generation: int) -> None:
    """Consumer side: buffer `index` of the producer whose board lives at `path` has been copied up to `generation`.
    A board that is gone (the producer stopped waiting) is not an error."""
    if not path or index >= _AckBoard.WORDS:
        return
    view = _ACK_VIEWS.get(path)
    if view is None:
        try:
            fd = os.open(path, os.O_RDWR | os.O_NOFOLLOW)
        except OSError:
            return
        try:
            mm = mmap.mmap(fd, 8 * _AckBoard.WORDS)
        except (OSError, ValueError):
            return
        finally:
            os.close(fd)
        view = _ACK_VIEWS[path] = (mm, np.frombuffer(mm, dtype=np.uint64))
    if int(view[1][index]) < generation:
        view[1][index] = generation


class _Slot:
    """One device buffer of a flow ring: the allocation (HEADER bytes, then the flow), the event behind the flow that
    was last written into it (`ready`), the event behind the last kernel that read it (`used`; None until somebody
    did), the generation it is in (the word at its head says the same on the device) and the generation under which a
    flow in it last left this process as an IPC token (None: never)."""

    def __init__(self, nbytes: int, index: int):
        from .device import DevBuffer
        self.buf = DevBuffer(nbytes + HEADER)
        self.flow_ptr = self.buf.ptr + HEADER
        self.index = index
        self.ready = _Event()
        self.used = None
        self.ipc_handle = None          # exported once, on first use
        self.generation = 0
        self.exported_gen = None

    def close(self):
        self.buf.close()
        self.ready.close()
        if self.used is not None:
            self.used.close()


class FlowRing:
    """The device buffers a flow source's DeviceFlows live in.  A buffer goes back into rotation when the DeviceFlow over
    it is garbage; the producer's stream then waits (on the device) for the last kernel that read it.  A flow that left
    as an IPC token is garbage here as soon as it is pickled, long before the other process has read it (it copies the
    flow out inside its queue.get(), see the module text): such a buffer stays out of rotation until the consumer has
    acknowledged that generation of it, the ring growing meanwhile.  `wait_for_acks = False` switches that rule off --
    for the test that makes a producer overrun its consumer and shows the consumer's generation check catching it."""

    DRAIN_TIMEOUT = 10.0     # seconds drain() waits for tokens on their way and for their acknowledgements
    TRANSIT_TIMEOUT = 1.0    # ... of which for flows that were handed out but never became a token (nothing else pending)

    def __init__(self, shape, slots: int = 4, wait_for_acks: bool = True):
        self.shape = tuple(int(v) for v in shape)
        self.nbytes = int(np.prod(self.shape)) * 4
        self.slots = max(2, int(slots))
        self.wait_for_acks = bool(wait_for_acks)
        self._all: list[_Slot] = []
        self._free: list[_Slot] = []
        self._board: _AckBoard | None = None
        self._in_transit = 0             # DeviceFlows of the "ipc" kind alive and not yet exported
        self._count_lock = threading.RLock()    # the queue's feeder thread and the producer's thread both count (re-entrant:
                                                # a flow's __del__ may run while its own thread holds the lock)
        self.exports = 0

    # ---- the exporting side's bookkeeping ------------------------------------------------------------------------
    def exported(self, slot: "_Slot"):
        """A flow in `slot` leaves as a token (called by the queue's pickler, in its feeder thread): returns what the
        token carries besides the handle -- the buffer's generation and the acknowledgement board's path."""
        with self._count_lock:
            if self._board is None:
                self._board = _AckBoard()
            slot.exported_gen = slot.generation
            self.exports += 1
        return slot.generation, (self._board.path if slot.index < _AckBoard.WORDS else "")

    def _acked(self, slot: "_Slot") -> bool:
        if slot.exported_gen is None:
            return True
        if self._board is None or slot.index >= _AckBoard.WORDS:
            return False
        return self._board.acked(slot.index) >= slot.exported_gen

    def unacknowledged(self) -> list:
        """Indices of the buffers whose last token no consumer has acknowledged yet."""
        return [s.index for s in self._all if not self._acked(s)]

    def take(self) -> _Slot:
        slot = next((s for s in self._free if not self.wait_for_acks or self._acked(s)), None)
        if slot is not None:
            self._free.remove(slot)
        else:
            slot = _Slot(self.nbytes, len(self._all))
            self._all.append(slot)           # (more flows held -- or on their way to another process -- than `slots`: the ring grows)
        if slot.used is not None:
            slot.used.stream_wait()          # the producer's writes stay behind the consumer's last read
        # a new generation, on the device BEFORE anything of the new flow is written (same stream): a consumer of another
        # process that still copies the old flow out of this buffer will find the word changed and say so
        slot.generation += 1
        slot.exported_gen = None
        check(_lib.load().tf_dev_store_u64(C.c_void_p(slot.buf.ptr), slot.generation))
        return slot

    def give_back(self, slot: _Slot) -> None:
        if slot in self._all and slot not in self._free:
            self._free.append(slot)

    def drain(self, timeout: float | None = None) -> bool:
        """Before the producer lets go of its buffers (or ends): wait until no flow of the "ipc" kind is still on its way
        into a token (the queue pickles in a feeder thread, after put() has returned) and every token that left has
        been acknowledged by its consumer.  True if that happened within `timeout` seconds (default DRAIN_TIMEOUT); False
        leaves a consumer that comes later with an error from tf_ipc_open or the generation check, not with a wrong flow."""
        limit = self.DRAIN_TIMEOUT if timeout is None else timeout
        start = time.monotonic()
        while self._in_transit > 0 or self.unacknowledged():
            waited = time.monotonic() - start
            # a flow nobody ever sends anywhere (the caller's loop variable still holds the last one) stays "on its way"
            # for as long as it lives: the feeder thread needs milliseconds, so that part of the wait is cut short
            if waited > limit or (self._in_transit > 0 and not self.unacknowledged() and waited > min(limit, self.TRANSIT_TIMEOUT)):
                return False
            time.sleep(0.0005)
        return True

    def close(self):
        """Frees every buffer NOW (callers that know no flow of the ring is in use any more); drain() first where tokens
        may be on their way."""
        for s in self._all:
            s.close()
        self._all, self._free = [], []
        if self._board is not None:
            self._board.close()
            self._board = None

    def __del__(self):
        try:
            if self._board is not None:
                self._board.close()
        except Exception:
            pass


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
        self._transit = bool(cross_process == "ipc" and ring is not None)
        if self._transit:
            with ring._count_lock:
                ring._in_transit += 1      # until it is exported, read on the host, or garbage (FlowRing.drain waits for that)
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

    def _left_transit(self) -> None:
        if self._transit:
            with self._ring._count_lock:
                if self._transit:
                    self._transit = False
                    self._ring._in_transit -= 1

    def __del__(self):
        try:
            self._left_transit()
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
            generation, board = flow._ring.exported(slot) if flow._ring is not None else (slot.generation, "")
            flow._left_transit()
            return (_open_from_queue, (slot.ipc_handle, os.getpid(), slot.index, flow.shape, flow.in_frame, generation, board))
        except Exception:                            # no IPC on this system: the array crosses instead
            flow._cross = None
    out = (np.array, (flow.host(),))
    flow._left_transit()
    return out


def _close_mappings_of_dead_producers(keep_pid: int) -> None:
    lib = _lib.load()
    for key in [k for k in _OPENED if k[0] != keep_pid]:
        try:
            os.kill(key[0], 0)
        except ProcessLookupError:
            lib.tf_ipc_close(C.c_void_p(_OPENED.pop(key)))
        except OSError:
            pass


def _open_from_queue(handle: bytes, pid: int, index: int, shape, in_frame: bool = False, generation: int = 0,
                     board: str = ""):
    """In the consumer's process: map the producer's buffer (once per buffer), copy the flow out of it into memory of
    our own, check that the buffer is still in the generation the token was made under, acknowledge, hand the copy
    out -- the producer's buffer is free again when queue.get() returns."""
    lib = _lib.load()
    key = (pid, index, handle)
    src = _OPENED.get(key)
    if src is None:
        _close_mappings_of_dead_producers(pid)
        p = C.c_void_p()
        check(lib.tf_ipc_open(handle, C.byref(p)))
        src = _OPENED[key] = p.value
    ring = _CONSUMER_RINGS.setdefault(tuple(shape), FlowRing(shape, slots=3))
    slot = ring.take()
    check(lib.tf_dev_copy(C.c_void_p(slot.flow_ptr), C.c_void_p(src + HEADER), ring.nbytes))
    seen = np.zeros(1, np.uint64)
    check(lib.tf_dev_download(C.c_void_p(seen.ctypes.data), C.c_void_p(src), 8))   # same stream: after the copy; synchronises
    _acknowledge(board, index, generation)           # done with the producer's memory, whatever we found there
    if generation and int(seen[0]) != generation:
        ring.give_back(slot)
        raise RuntimeError(f"device flow buffer {index} of process {pid} was written again (generation {int(seen[0])}, the "
                           f"token says {generation}) before this process had copied the flow out of it")
    slot.ready.record()
    flow = DeviceFlow(shape, slot.flow_ptr, slot.ready, ring=ring, slot=slot)
    flow.in_frame = bool(in_frame)
    return flow


_CONSUMER_RINGS: dict = {}


def _register():
    from multiprocessing.reduction import ForkingPickler
    ForkingPickler.register(DeviceFlow, _reduce_for_queue)


_register()
