"""Flow sources backed by libtfhip.so, behind the reference's FlowSource surface.

`FlowSource` mirrors transflow/flow/sources/source.py:17-415 (Direction, LockMode,
Builder with its seek/duration/repeat/lock arithmetic, the iterator protocol and
post_process); `HipFlowSource` mirrors CvFlowSource's Farnebäck branch
(transflow/flow/sources/cv.py:366-524): frames come from a frame provider, the
flow from the GPU.  Items yielded are numpy float32 arrays of shape (H, W, 2),
picklable, exactly what pipeline.py:85-86 puts on its queue.

No HIP call happens before Builder.__enter__/build(): the reference forks the
flow source into a child process (pipeline.py:56-64).
"""
from __future__ import annotations

import dataclasses
import enum
import logging
import os
import warnings
import numpy as np

from .config import FlowConfig

logger = logging.getLogger(__name__)


class FlowFilter:
    """A flow filter of the reference (transflow/flow/filters.py): `name=expr`.  scale / threshold /
    clip: expr is a Python expression of t evaluated on the host for every frame, the arithmetic on
    the flow runs on the GPU.  polar: `polar=expr_radius:expr_theta`, expressions of (t, r, a)
    compiled for the device (transflow_amd/exprs.py)."""

    NAMES = ("scale", "threshold", "clip", "polar")

    def __init__(self, name: str, expr_string: str):
        if name not in self.NAMES:
            raise ValueError(f"Unknown filter name '{name}'")                     # filters.py:33
        self.name = name
        self.expr_string = expr_string
        if name == "polar":
            from .exprs import PolarFilter
            parts = expr_string.split(":")
            if len(parts) != 2:
                raise ValueError(f"Invalid number of arguments: {name} {tuple(parts)}")   # filters.py:29-30
            self.polar = PolarFilter(parts[0], parts[1])
            self.expr = None
            return
        import math
        import random
        import re

        import numpy
        # the names the reference's utils module (where its eval runs, utils.py:407-412) offers
        scope = {"math": math, "numpy": numpy, "random": random, "re": re, "os": os}
        self.expr = eval("lambda t: " + expr_string, scope)

    @classmethod
    def from_string(cls, text: str):
        """'scale=2*t' -> FlowFilter('scale', '2*t')   (source.py:143-149)"""
        i = text.index("=")
        args = tuple(text[i + 1:].strip().split(":"))
        if len(args) != 1 and text[:i].strip() != "polar":
            raise ValueError(f"Invalid number of arguments: {text[:i].strip()} {args}")  # filters.py:21-31
        return cls(text[:i].strip(), args[0] if len(args) == 1 else ":".join(args))


def _enum_from_arg(enum_cls, arg, default, what):
    """The argument forms the reference accepts for its two enums (source.py:24-56): None, a member
    (ours or the reference's own, matched by name), the integer value, or the lower-case name."""
    if arg is None:
        return default
    if isinstance(arg, enum.Enum):
        if arg.name in enum_cls.__members__:
            return enum_cls[arg.name]
    elif isinstance(arg, int):
        return enum_cls(arg)
    elif isinstance(arg, str) and arg.upper() in enum_cls.__members__ and arg == arg.lower():
        return enum_cls[arg.upper()]
    raise ValueError(f"Invalid {what}: {arg}")


@dataclasses.dataclass
class Timeline:
    """What a source's frame range and output length come to (source.py:151-197), as one value."""
    base_length: int | None     # frames the input holds; None = a stream
    repeat: int
    seek_time: float | None
    start_frame: int
    end_frame: int
    length: int | None          # flows the source will yield; None = endless
    ckpt_start_frame: int       # where the FIRST pass starts when resuming from a checkpoint

    @property
    def is_stream(self) -> bool:
        return self.base_length is None


def plan_timeline(base_length, framerate, seek_time, duration_time, repeat, seek_ckpt, stay_pairs) -> Timeline:
    """The seek / duration / repeat / checkpoint arithmetic of FlowSource.Builder.build
    (source.py:151-197).  A non-positive base length means a stream: it cannot repeat or seek (both
    are dropped with the reference's warnings).  Every STAY lock adds its duration to the length."""
    frames = base_length if (base_length is not None and base_length > 0) else None
    if frames is None:
        if repeat > 1:
            warnings.warn("Flow source is a stream, cannot repeat it!")
            repeat = 1
        if seek_time is not None and seek_time > 0:
            warnings.warn("Flow source is a stream, seek time is ignored!")
            seek_time = None
    first = 0
    if frames is not None and seek_time is not None:
        first = int(seek_time * framerate)
    last = 0 if frames is None else frames
    if duration_time is not None:
        # three decimals before truncating: 0.1 s at 29.97 fps must not lose a frame to 2.9969999
        last = first + int(round(duration_time * framerate, 3))
        if frames is not None and last > frames:
            last = frames
    if repeat == 0:
        total = None
    else:
        total = last if frames is None else repeat * (last - first)
        for _, held in (stay_pairs or ()):
            total += int(held * framerate)
    resume = first
    if seek_ckpt is not None:
        resume = first + seek_ckpt % (last - first)
    return Timeline(frames, repeat, seek_time, first, last, total, resume)


class LockSchedule:
    """When the source repeats its previous flow instead of reading a new one (source.py:296-311).
    STAY: (start, duration) pairs in output time; SKIP: a predicate of t (and the input advances)."""

    def __init__(self, mode, stay_pairs, skip_predicate):
        self.mode, self.pairs, self.predicate = mode, stay_pairs, skip_predicate
        self.which = 0              # the STAY pair in force or awaited
        self.since = None           # output time at which the current STAY lock began

    def locked(self, t: float) -> bool:
        if self.mode == FlowSource.LockMode.SKIP:
            return bool(self.predicate(t)) if self.predicate is not None else False
        if self.pairs is None:
            return False
        if self.since is not None:
            if t - self.since < self.pairs[self.which][1]:
                return True
            self.since = None
            self.which += 1
        # past the last pair this indexes out of range: the reference raises the same IndexError
        # (source.py:304-307), and the pipeline's source process stops on it (pipeline.py:90-97)
        if t >= self.pairs[self.which][0]:
            self.since = t
            return True
        return False


class FlowSource:
    """Base class with the reference's constructor, attributes and iteration protocol
    (transflow/flow/sources/source.py:17-415); the bodies are this package's own."""

    @enum.unique
    class Direction(enum.Enum):
        FORWARD = 0   # past to present
        BACKWARD = 1  # present to past

        @classmethod
        def from_arg(cls, arg):
            return _enum_from_arg(cls, arg, cls.FORWARD, "Flow Direction")     # None -> FORWARD, source.py:26-28

    @enum.unique
    class LockMode(enum.Enum):
        STAY = 0
        SKIP = 1

        @classmethod
        def from_arg(cls, arg):
            return _enum_from_arg(cls, arg, cls.STAY, "Lock Mode")

    class Builder:
        """Context manager: `build()` resolves the arguments, `__enter__` makes the source
        (source.py:58-209).  Attribute names are the reference's: its callers and subclasses read them."""

        def __init__(self, direction="backward", mask_path=None, kernel_path=None, flow_filters=None,
                     seek_ckpt=None, seek_time=None, duration_time=None, repeat: int = 1, lock_expr=None,
                     lock_mode="stay"):
            self.direction = FlowSource.Direction.from_arg(direction)
            self.lock_mode = FlowSource.LockMode.from_arg(lock_mode)
            self.mask_path, self.kernel_path = mask_path, kernel_path
            self.flow_filters_string, self.lock_expr_string = flow_filters, lock_expr
            self.seek_ckpt, self.seek_time, self.duration_time, self.repeat = seek_ckpt, seek_time, duration_time, repeat
            # filled by build() (subclasses set width / height / framerate / base_length before calling it)
            self.width = self.height = None
            self.framerate: float = 30
            self.base_length = self.length = None
            self.is_stream = False
            self.start_frame = self.ckpt_start_frame = self.end_frame = 0
            self.mask = self.kernel = None
            self.flow_filters: list = []
            self.lock_expr_stay = self.lock_expr_skip = None
            self.source: FlowSource | None = None

        @property
        def cls(self):
            return FlowSource

        def args(self) -> list:
            return [getattr(self, name) for name in ("direction", "width", "height", "framerate", "length",
                                                     "start_frame", "ckpt_start_frame", "end_frame")]

        def kwargs(self) -> dict:
            return {name: getattr(self, name) for name in ("mask", "kernel", "flow_filters", "lock_mode",
                                                           "lock_expr_stay", "lock_expr_skip")}

        def _load_inputs(self):
            if self.mask_path is not None:                       # a float mask, one trailing axis (source.py:127-129)
                from .masks import load_float_mask
                self.mask = load_float_mask(self.mask_path)[..., np.newaxis]
            if self.kernel_path is not None:                     # source.py:131-132
                self.kernel = np.load(self.kernel_path)
            if self.flow_filters_string is not None:             # "name=expr; name=expr" (source.py:141-149)
                self.flow_filters = [FlowFilter.from_string(part) for part in self.flow_filters_string.strip().split(";")]
            text = self.lock_expr_string
            if text is None:
                return
            if self.lock_mode == FlowSource.LockMode.SKIP:       # a predicate of t (source.py:138-139)
                self.lock_expr_skip = eval("lambda t: " + text)
            else:                                                # "a,b" or "(a,b),(c,d)": (start, duration) pairs
                self.lock_expr_stay = tuple(eval("[" + (text if "(" in text else "(" + text + ")") + ",]"))

        def build(self):
            self._load_inputs()
            plan = plan_timeline(self.base_length, self.framerate, self.seek_time, self.duration_time, self.repeat,
                                 self.seek_ckpt,
                                 self.lock_expr_stay if self.lock_mode == FlowSource.LockMode.STAY else None)
            self.base_length, self.is_stream = plan.base_length, plan.is_stream
            self.repeat, self.seek_time = plan.repeat, plan.seek_time
            self.start_frame, self.end_frame, self.length = plan.start_frame, plan.end_frame, plan.length
            self.ckpt_start_frame = plan.ckpt_start_frame

        def __enter__(self):
            self.build()
            source = self.cls(*self.args(), **self.kwargs())
            source.validate()
            logger.debug("Built '%s'", type(source).__name__)
            self.source = source          # what __exit__ closes
            return source

        def __exit__(self, *exc):
            if self.source is not None:
                self.source.close()

    def __init__(self, direction, width: int, height: int, framerate: float, length, start_frame: int,
                 ckpt_start_frame: int, end_frame: int, mask=None, kernel=None, flow_filters=(),
                 lock_mode=None, lock_expr_stay=None, lock_expr_skip=None):
        self.direction = FlowSource.Direction.from_arg(direction)
        self.width, self.height, self.framerate = width, height, framerate
        self.length, self.end_frame = length, end_frame
        self.mask, self.kernel, self.flow_filters = mask, kernel, list(flow_filters)
        if kernel is not None:
            from .flowops import kernel_result_type
            if not isinstance(kernel, np.ndarray):
                raise ValueError(f"Attribute kernel has incorrect type {type(kernel)}")   # source.py:281
            kernel_result_type(kernel)   # float32 / float64 results only
        if not all(isinstance(f, FlowFilter) for f in self.flow_filters):
            raise ValueError("flow_filters must be transflow_amd.flow.FlowFilter objects")
        self.lock_mode = FlowSource.LockMode.from_arg(lock_mode)
        self.lock_expr_stay, self.lock_expr_skip = lock_expr_stay, lock_expr_skip
        self._locks = LockSchedule(self.lock_mode, lock_expr_stay, lock_expr_skip)
        self.output_frame_index = 0
        self.prev_flow = None
        self._pp = None  # device handle used by post_process, created on first use
        # the first pass starts where a checkpoint left off, later passes at start_frame (source.py:246-248)
        self.input_frame_index = 0
        self.start_frame, later_passes = ckpt_start_frame, start_frame
        self.rewind()                                    # subclasses decode up to start_frame here
        self.start_frame = later_passes
        # source.py:250-263 builds per-pixel clip tables on the host (a 1.2 s Python loop at
        # 1080p); the kernels derive the bounds from the pixel index instead

    # the STAY bookkeeping under the reference's attribute names (its checkpoints and tests read them)
    lock_start = property(lambda self: self._locks.since)
    lock_expr_stay_index = property(lambda self: self._locks.which)

    def __len__(self):
        return self.length

    def validate(self):
        """source.py:268-284: the constructor arguments have the types the pipeline relies on."""
        expected = {"direction": FlowSource.Direction, "width": int, "height": int, "framerate": float,
                    "length": (int, type(None)), "start_frame": int, "end_frame": int, "flow_filters": list,
                    "lock_mode": FlowSource.LockMode, "lock_expr_stay": (tuple, type(None))}
        for name, types in expected.items():
            value = getattr(self, name)
            if not isinstance(value, types):
                raise ValueError(f"Attribute {name} has incorrect type {type(value)}")

    @property
    def t(self) -> float:
        """Output time in seconds."""
        return self.output_frame_index / self.framerate if self.framerate is not None else 0

    def read_next_flow(self):
        """One input flow; the input wraps to start_frame when it reaches end_frame (source.py:286-291)."""
        if self.input_frame_index == self.end_frame:
            self.rewind()
        flow = self.next()
        self.input_frame_index += 1
        return flow

    def __iter__(self):
        return self

    def __next__(self):
        """One output flow (source.py:293-321): a locked source hands out its previous flow again
        (and, in SKIP mode, still consumes an input flow); everything goes through post_process."""
        if self.length is not None and self.output_frame_index >= self.length:
            raise StopIteration
        if self._locks.locked(self.t):
            if self.prev_flow is None:
                raise RuntimeError("Flow is locked but has not been initialized. Maybe lock the flow later?")
            flow = self.prev_flow
            if self.lock_mode == FlowSource.LockMode.SKIP:
                self.read_next_flow()
        else:
            flow = self.prev_flow = self.read_next_flow()
        self.output_frame_index += 1
        return self.post_process(flow)

    def next(self):
        raise NotImplementedError()

    def rewind(self):
        self.input_frame_index = self.start_frame

    def post_process(self, raw):
        """source.py:337-363 on the GPU (tf_fb_post_process_host): FORWARD inverts the push
        field with last-write-wins, both directions clip to the frame.  In place, like the
        reference (so `prev_flow` sees the processed array, source.py:317)."""
        flow = raw
        if (isinstance(raw, np.ndarray) and np.issubdtype(raw.dtype, np.integer) and not self.flow_filters
                and self.mask is None and self.kernel is None):
            # a rounded archive (pipeline.py:506 writes numpy.round(flow).astype(int)): the reference clips
            # and inverts the integer array in place; small integers are exact in float32
            out = self.post_process(raw.astype(np.float32))
            raw[...] = out.astype(raw.dtype)
            return raw
        if not (isinstance(flow, np.ndarray) and flow.dtype == np.float32 and flow.flags.c_contiguous):
            flow = np.ascontiguousarray(raw, dtype=np.float32)
        if self._pp is None:
            from .farneback import Farneback
            self._pp = Farneback(self.width, self.height, levels=0)
        if any(f.name == "polar" for f in self.flow_filters):
            # filters apply in order (source.py:339-341): runs of scale/threshold/clip go to the device as
            # one launch each, every polar filter as its own; the rest of post_process follows unfiltered
            from .flowops import polar_filter
            run = []
            for f in list(self.flow_filters) + [None]:
                if f is not None and f.name != "polar":
                    run.append((f.name, f.expr(self.t)))
                    continue
                if run:
                    self._pp.post_process_host_ex(flow, None, run)
                    run = []
                if f is not None:
                    polar_filter(flow, f.polar, self.t)
            filters, self.flow_filters = self.flow_filters, []
            try:
                out = self.post_process(flow)
            finally:
                self.flow_filters = filters
            return out
        ops = [(f.name, f.expr(self.t)) for f in self.flow_filters]   # filters.py: lambdas of t, host side
        if self.kernel is not None:
            # source.py:339-348: filters in place, mask multiply into a new array, then the convolution
            # of both channels (a NEW array of the convolution's type, float64 unless the kernel is
            # float32) which the direction handling and the clip then work on
            from .flowops import convolve_post_process
            if ops:
                self._pp.post_process_host_ex(flow, None, ops)
            pre = flow
            if self.mask is not None:
                pre = flow.copy()
                self._pp.post_process_host_ex(pre, None, (), self.mask)
            return convolve_post_process(pre, self.kernel, self.direction.value)
        if self.mask is None:
            self._pp.post_process_host_ex(flow, self.direction.value, ops)
            return flow
        # the reference applies the filters IN PLACE on the raw flow (so prev_flow sees them) and
        # then builds a NEW array with the mask multiply (source.py:339-343)
        if ops:
            self._pp.post_process_host_ex(flow, None, ops)
        out = flow.copy()
        self._pp.post_process_host_ex(out, self.direction.value, (), self.mask)
        return out

    def close(self):
        if self._pp is not None:
            self._pp.close()
            self._pp = None


class ArrayFrameProvider:
    """Frames held in memory: a sequence of uint8 arrays, grey (H, W) or BGR (Hs, Ws, 3).  `size` = (width,
    height) the source reports (cv2.CAP_PROP_FRAME_WIDTH / HEIGHT, cv.py:420-427); BGR frames of another size
    are brought to it by the nearest-neighbour resize of cv.py:461, on the device."""

    def __init__(self, frames, framerate: float = 30.0, size=None):
        self.frames = frames
        self.framerate = float(framerate)
        first = np.asarray(frames[0])
        self.height, self.width = first.shape[:2]
        if size is not None:
            self.width, self.height = int(size[0]), int(size[1])
        self.frame_count = len(frames)
        self.pos = 0

    def seek_start(self):
        self.pos = 0

    def read(self):
        if self.pos >= self.frame_count:
            return None
        f = np.asarray(self.frames[self.pos])
        self.pos += 1
        return f

    def release(self):
        pass


class Cv2FrameProvider:
    """Video decode through cv2.VideoCapture, as CvFlowSource does (cv.py:416-429, 447-466).
    Decode stays on the CPU and is out of this backend's scope; it needs opencv-python."""

    def __init__(self, file: str, size=None):
        import re

        import cv2
        self.cv2 = cv2
        self.capture = cv2.VideoCapture(int(file)) if re.match(r"\d+", file) else cv2.VideoCapture(file)
        if size is not None:
            self.capture.set(cv2.CAP_PROP_FRAME_WIDTH, size[0])
            self.capture.set(cv2.CAP_PROP_FRAME_HEIGHT, size[1])
        self.width = int(self.capture.get(cv2.CAP_PROP_FRAME_WIDTH))
        self.height = int(self.capture.get(cv2.CAP_PROP_FRAME_HEIGHT))
        self.framerate = float(self.capture.get(cv2.CAP_PROP_FPS))
        self.frame_count = int(self.capture.get(cv2.CAP_PROP_FRAME_COUNT))

    def seek_start(self):
        self.capture.set(self.cv2.CAP_PROP_POS_MSEC, 0)

    def read(self):
        ok, frame = self.capture.read()
        if not ok or frame is None:
            return None
        return frame     # as decoded: the resize of cv.py:461 and the grey conversion run on the device

    def release(self):
        self.capture.release()


class _Prefetch:
    """A worker thread that iterates a flow source ahead of its consumer (HipFlowSource.__next__)."""

    def __init__(self, source, depth: int):
        import queue
        import threading
        self.source = source
        self.queue = queue.Queue(maxsize=max(1, int(depth)))
        self.halt = threading.Event()
        self.finished = None                       # ("stop", None) or ("error", exception) once the worker has ended
        self.thread = threading.Thread(target=self._run, name="tfhip-flow-prefetch", daemon=True)
        self.thread.start()

    def _put(self, item) -> bool:
        import queue
        while not self.halt.is_set():
            try:
                self.queue.put(item, timeout=0.05)
                return True
            except queue.Full:
                continue
        return False

    def _run(self):
        from . import _lib
        try:
            _lib.check(_lib.load().tf_thread_stream(1))
            # Flow t's download (started by post_process) runs beside frame t + 1's upload and kernels: the worker issues
            # t + 1 first and only then waits for t's transfer and hands the array over.
            held = None                              # (flow, token of its download) not yet handed over

            def hand_over():
                nonlocal held
                if held is None:
                    return True
                flow, token = held
                held = None
                if token is not None:
                    self.source._fb.get_flow_end(token)
                return self._put(("flow", flow))
            while not self.halt.is_set():
                try:
                    self.source._download_token = None
                    flow = FlowSource.__next__(self.source)
                except StopIteration:
                    if hand_over():
                        self._put(("stop", None))
                    return
                except BaseException:
                    hand_over()                      # the flows before the failure still reach the consumer, in order
                    raise
                token = self.source._download_token
                if not hand_over():
                    return
                held = (flow, token)
        except BaseException as err:            # noqa: BLE001 -- re-raised in the consumer's thread
            self._put(("error", err))

    def get(self):
        if self.finished is None:
            kind, value = self.queue.get()
            if kind == "flow":
                return value
            self.finished = (kind, value)
        if self.finished[0] == "error":
            raise self.finished[1]
        raise StopIteration

    def stop(self):
        import queue
        self.halt.set()
        while self.thread.is_alive():
            try:
                self.queue.get_nowait()
            except queue.Empty:
                pass
            self.thread.join(timeout=0.05)


class HipFlowSource(FlowSource):
    """CvFlowSource's Farnebäck branch on the GPU (cv.py:434-521)."""

    class Builder(FlowSource.Builder):

        def __init__(self, provider, config=None, size=None, device: int | None = None, **kwargs):
            super().__init__(**kwargs)
            self.provider_arg, self.size, self.device = provider, size, device
            self.config = FlowConfig.from_reference(config)
            self.provider = None

        @property
        def cls(self):
            return HipFlowSource

        def build(self):
            p = self.provider_arg
            self.provider = Cv2FrameProvider(p, self.size) if isinstance(p, str) else p
            self.width, self.height = int(self.provider.width), int(self.provider.height)
            self.framerate = float(self.provider.framerate)
            self.base_length = int(self.provider.frame_count) - 1                        # cv.py:428
            super().build()

        def args(self):
            return [self.provider, self.config, *FlowSource.Builder.args(self)]

        def kwargs(self):
            kw = super().kwargs()
            kw["device"] = self.device
            return kw

    def __init__(self, provider, config: FlowConfig, *args, device: int | None = None, **kwargs):
        self.config = config
        self.provider = provider
        self.device = device
        self._prev_frame = None  # the decoded frame behind prev_gray
        self._fb = None
        self._prev_slot = None   # frame slot holding prev_gray on the device
        self._batch_left = 0     # FlowConfig.hip_batch: flows of the last call not handed out yet ...
        self._batch_pos = 0      # ... and which pair of the call comes next
        self._pending_pair = 0   # the pair of the call behind the array read_next_flow handed out
        self._pending = None     # array handed out by read_next_flow whose flow is still on the device
        self._mask_dev = None
        self._flow_pool = None
        self._flow_ring = None   # FlowConfig.hip_device_flows: the device buffers the yielded DeviceFlows live in
        self._prefetch = None
        self._download_token = None
        FlowSource.__init__(self, *args, **kwargs)

    def validate(self):
        super().validate()
        if not isinstance(self.config, FlowConfig):
            raise ValueError("Attribute config has incorrect type")

    def _handle(self):
        if self._fb is None:
            from .farneback import Farneback
            # Exactness belongs to the handle (tf_fb_set_exact): this source states what its configuration says, on OR off,
            # and touches nothing process-wide -- sources that disagree may be open together, in any threads.
            batch = self._batch_size()
            self._fb = Farneback(self.width, self.height, device=self.device, frame_slots=batch + 1, max_pairs=batch,
                                 exact=bool(getattr(self.config, "hip_exact_sums", False)), **self.config.fb_kwargs())
            self._fb.keep_expansions(True)  # the frame that was "next" stays expanded for its turn as "prev"
            if getattr(self.config, "hip_prefetch", 0):
                self._fb.async_io(True)     # the next frame up and the previous flow down beside this pair's kernels
            self._pp = self._fb  # one handle serves both calls
        return self._fb

    # The frame the reference keeps as `prev_gray` (cv.py:456, 519) lives in a frame slot of the handle; on the
    # host only the decoded frame it came from is kept (to fill the slot again after a rewind).  Frames are
    # made grey ON THE DEVICE (tf_fb_set_frame_bgr: cv2.resize INTER_NEAREST + COLOR_BGR2GRAY of cv.py:461-466
    # as one kernel into the slot, OpenCV 4's 15-bit weights); a provider of grey frames is taken as it is.
    @property
    def prev_gray(self):
        f = self._prev_frame
        if f is None or np.asarray(f).ndim == 2:
            return f
        from .flowops import bgr_to_grey
        return bgr_to_grey(f, (self.width, self.height))

    @prev_gray.setter
    def prev_gray(self, value):
        self._prev_frame = value
        self._prev_slot = None
        self._batch_left = self._batch_pos = 0     # flows of a hip_batch call made before the seek are not handed out after it

    def _ingest(self, slot: int, frame) -> None:
        a = np.asarray(frame)
        if a.ndim == 2:
            self._handle().set_frame(slot, a)
        else:
            self._handle().set_frame_bgr(slot, a)

    def rewind(self):
        """cv.py:447-458: decode from the start up to the start frame, keep it as `prev`."""
        FlowSource.rewind(self)
        self.provider.seek_start()
        frame = None
        for i in range(self.input_frame_index + 1):
            frame = self.provider.read()
            if frame is None:
                raise RuntimeError(f"An error occurred while reading frame at index {i}")
        self._prev_frame = frame
        self.prev_flow = None
        self._prev_slot = None
        self._batch_left = self._batch_pos = 0     # (an external rewind while a hip_batch call still had flows queued)

    def _batch_size(self) -> int:
        return max(1, int(getattr(self.config, "hip_batch", 1))) if self._resident_ok() else 1

    def _advance(self, want: int = 1) -> int:
        """cv.py:460-490 up to the call: read a frame, make it grey in a slot no pair still needs, order (prev, next) by
        direction, run Farnebäck on the two slots.  The flow stays on the device.  `want` > 1 (FlowConfig.hip_batch): read
        up to that many frames and run their consecutive pairs in ONE call -- frame slots are a ring of hip_batch + 1, the
        last frame of a call is the first of the next; returns how many pairs were computed."""
        frames = []
        for _ in range(max(1, want)):
            frame = self.provider.read()
            if frame is None:
                break
            frames.append(frame)
        if not frames:
            raise StopIteration
        if self._prev_frame is None:
            raise ValueError("Missing reference frames")
        fb = self._handle()                          # (the first GPU call of a source: after the provider has spoken)
        if self._prev_slot is None:                  # first frame, or after a rewind
            self._ingest(0, self._prev_frame)
            self._prev_slot = 0
        ring = fb.frame_slots
        older, newer = [], []
        slot = self._prev_slot
        for frame in frames:
            new_slot = (slot + 1) % ring
            self._ingest(new_slot, frame)
            older.append(slot)
            newer.append(new_slot)
            slot = new_slot
        if self._uses_initial_flow():                # cv.py:478: a copy of the previous flow, zeros before the first
            init = self.prev_flow if self.prev_flow is not None else np.zeros((self.height, self.width, 2), np.float32)
            fb.set_initial_flow(0, init)
        if self.direction == FlowSource.Direction.FORWARD:      # cv.py:467-472
            fb.calc_slots(older, newer)
        elif self.direction == FlowSource.Direction.BACKWARD:
            fb.calc_slots(newer, older)
        else:
            raise ValueError(f"Invalid flow direction '{self.direction}'")
        self._prev_slot, self._prev_frame = slot, frames[-1]
        return len(frames)

    # ---- resident form of one iteration -------------------------------------------------------
    # __next__ (source.py:293-321) calls read_next_flow() and hands its result straight to
    # post_process().  When nothing can look at the raw flow in between (no lock expressions: they are
    # what reads prev_flow) the flow stays on the device from the Farnebäck call through the filters,
    # the mask and the direction handling, only the new frame goes up and only the final flow comes
    # down -- one transfer each instead of two frames up and the flow down, up and down again.  The
    # public next() / post_process() pair keeps working on host arrays for any other caller.
    def _uses_initial_flow(self) -> bool:
        return bool(self.config.fb_flags & 4)      # cv2.OPTFLOW_USE_INITIAL_FLOW

    def _resident_ok(self) -> bool:
        # with OPTFLOW_USE_INITIAL_FLOW every call starts from the previous OUTPUT (cv.py:478 passes a copy of
        # prev_flow, which __next__ has post-processed in place): that array lives on the host
        return (self.lock_expr_stay is None and self.lock_expr_skip is None and self.kernel is None
                and not self._uses_initial_flow() and not any(f.name == "polar" for f in self.flow_filters))

    def read_next_flow(self):
        if not self._resident_ok():
            return FlowSource.read_next_flow(self)
        if self.input_frame_index == self.end_frame:
            self.rewind()
            self._batch_left = 0
        if self._batch_left == 0:
            # never across the wrap of the input (source.py:286-291: the frame after end_frame is start_frame's)
            until_wrap = self.end_frame - self.input_frame_index if self.end_frame > self.input_frame_index else self._batch_size()
            self._batch_left = self._advance(min(self._batch_size(), max(1, until_wrap)))
            self._batch_pos = 0
        self._pending_pair = self._batch_pos
        self._batch_pos += 1
        self._batch_left -= 1
        self.input_frame_index += 1
        if getattr(self.config, "hip_device_flows", False):
            # the flow stays in HBM: a DeviceFlow over a buffer of the source's ring, filled by post_process
            from .deviceflow import DeviceFlow, FlowRing
            if self._flow_ring is None:
                self._flow_ring = FlowRing((self.height, self.width, 2), slots=4 + self.config.hip_prefetch)
            slot = self._flow_ring.take()
            self._pending = DeviceFlow((self.height, self.width, 2), slot.flow_ptr, slot.ready, ring=self._flow_ring, slot=slot,
                                       cross_process="ipc" if self.config.hip_device_flows == "ipc" else None)
            return self._pending
        if self._flow_pool is None:
            from .device import ArrayPool
            self._flow_pool = ArrayPool((self.height, self.width, 2), np.float32, limit=4 + self.config.hip_prefetch, pinned=True)
        self._pending = self._flow_pool.take()       # filled by post_process
        return self._pending

    def post_process(self, raw):
        if self._pending is None or raw is not self._pending:
            return FlowSource.post_process(self, raw)
        self._pending = None
        fb = self._fb
        ops = [(f.name, f.expr(self.t)) for f in self.flow_filters]
        mask_dev = None
        if self.mask is not None:
            if self._mask_dev is None:
                from .device import DevBuffer
                self._mask_dev = DevBuffer.from_array(
                    np.ascontiguousarray(self.mask, dtype=np.float32).reshape(self.height, self.width))
            mask_dev = self._mask_dev.ptr
        pair = self._pending_pair
        fb.post_process_ex(pair, self.direction.value, ops, mask_dev)
        if not isinstance(raw, np.ndarray):
            # a DeviceFlow: out of the handle's result buffer (the next call but one writes it again) into the flow's own,
            # device to device on this thread's stream; the event behind the copy is what consumers wait for
            from . import _lib
            import ctypes as C
            _lib.check(_lib.load().tf_dev_copy(C.c_void_p(raw.dev_ptr), C.c_void_p(fb.flow_ptr(pair)), raw.nbytes))
            raw._ready.record()
            raw.in_frame = True          # both directions of post_process end with the clip (source.py:361-362)
            return raw
        if self._prefetch is not None:
            # the worker thread: the flow starts its way down and the worker goes on to the next frame; it hands this
            # array to the consumer only once the transfer has ended (_Prefetch._run)
            self._download_token = fb.get_flow_begin(pair, raw)
        else:
            fb.get_flow_into(pair, raw)
        return raw

    def next(self):
        """cv.py:460-490: (prev, next) ordered by direction, one Farnebäck call; the flow as a host array."""
        self._advance()
        return self._handle().get_flow(0)

    # ---- prefetch (FlowConfig.hip_prefetch) ----------------------------------------------------
    # The reference runs its flow source in a child process that fills a queue (pipeline.py:56-64, 85-86), so flow
    # t + 1 is computed while the compositor works on flow t.  A source iterated in-process gets the same from a
    # worker thread: it runs FlowSource.__next__ -- the whole recurrence, locks and rewinds included -- up to
    # `hip_prefetch` flows ahead, queueing everything it launches on a library stream of its own (tf_thread_stream),
    # beside the compositor's uploads, kernels and downloads.  ctypes releases the GIL inside every library call.
    def __next__(self):
        if not getattr(self.config, "hip_prefetch", 0):
            return FlowSource.__next__(self)
        if self._prefetch is None:
            self._prefetch = _Prefetch(self, self.config.hip_prefetch)
        return self._prefetch.get()

    def close(self):
        if self._prefetch is not None:
            self._prefetch.stop()
            self._prefetch = None
            if self._fb is not None:
                self._fb.async_io(False)    # waits for a download the worker left on its way
        if self._mask_dev is not None:
            self._mask_dev.close()
            self._mask_dev = None
        self._pending = None
        if self._flow_ring is not None:
            # flows that left this process as IPC tokens: multiprocessing's Queue.get() frees the queue's slot before it
            # unpickles, so the last put() of SourceProcess.run (pipeline.py:85-86) can return -- and this process end --
            # before the consumer has opened the last flow's handle.  Wait (bounded) until every token on its way has
            # been made and acknowledged; our own reference to the last flow goes first.
            self.prev_flow = None
            self._flow_ring.drain()
        self._flow_ring = None      # (buffers live as long as a DeviceFlow the caller still holds)
        if self._fb is not None:
            self._fb.close()
            self._fb = None
            self._pp = None
        else:
            FlowSource.close(self)
        self.provider.release()

    @classmethod
    def from_args(cls, flow_path, use_mvs: bool = False, mask_path=None, kernel_path=None, cv_config=None,
                  flow_filters=None, size=None, direction=None, seek_ckpt=None, seek_time=None,
                  duration_time=None, repeat: int = 1, lock_expr=None, lock_mode="stay"):
        """Same signature as FlowSource.from_args (source.py:365-411); `flow_path` may also be
        a frame provider object.  `.flow.zip` archives go to ArchiveFlowSource (source.py:397-399);
        motion-vector sources are not this backend's."""
        if isinstance(flow_path, str) and flow_path.split("::")[-1].endswith(".flow.zip"):
            from .archive import ArchiveFlowSource
            return ArchiveFlowSource.Builder(flow_path.split("::")[-1], direction=direction, mask_path=mask_path,
                                             kernel_path=kernel_path, flow_filters=flow_filters, seek_ckpt=seek_ckpt,
                                             seek_time=seek_time, duration_time=duration_time, repeat=repeat,
                                             lock_expr=lock_expr, lock_mode=lock_mode)
        if use_mvs:
            raise NotImplementedError("transflow_amd does not read codec motion vectors")
        if isinstance(cv_config, str):
            config = FlowConfig.from_file(cv_config) if os.path.isfile(cv_config) else FlowConfig()
        else:
            config = FlowConfig.from_reference(cv_config)
        if isinstance(flow_path, str) and "::" in flow_path:
            flow_path = flow_path.split("::")[1]
        return cls.Builder(flow_path, config, size, direction=direction, mask_path=mask_path,
                           kernel_path=kernel_path, flow_filters=flow_filters, seek_ckpt=seek_ckpt,
                           seek_time=seek_time, duration_time=duration_time, repeat=repeat, lock_expr=lock_expr,
                           lock_mode=lock_mode)
