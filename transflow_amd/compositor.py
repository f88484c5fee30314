"""Compositor backed by libtfhip.so, behind the reference's Compositor surface.

  HipCompositor            <- transflow/compositor/compositor.py:17-53
  HipMoveReferenceLayer    <- transflow/compositor/layers/move_reference.py:6-14 and its bases
                              (layer.py:11-55, data.py:6-17, movement.py:10-64, reference.py:31-109)
  HipSumLayer, HipStaticLayer, HipIntroductionLayer
                           <- layers/sum.py:7-14, static.py:7-17, introduction.py:8-73

`pipeline.py` only needs: Compositor.from_args(height, width, layer_configs,
background_color), set_sources({layer: [PixmapSourceInterface]}), update(flow),
render() -> uint8 (H, W, 3), `.layers[i].sources`, and a picklable object
(pipeline.py:225-242, 290-306, 440-455, 518, 565).  `extra/control.py:146-162`
reads `layer.data` and `layer.INDEX_I/J/ALPHA`.

Device state is created lazily on the first update/render and never pickled:
__getstate__ downloads `data`/`rgba` into numpy arrays.
"""
from __future__ import annotations

import numpy as np

from .config import LayerConfig
from .masks import load_bool_mask, load_float_mask, parse_color


class _HipLayer:
    """What every layer class shares (layer.py:11-34): config, size, sources, mask_alpha, lazy device
    state, host-only pickling.  Subclasses set LAYER_CLASS / DEPTH / INDEX_* and write `update`."""

    LAYER_CLASS = "moveref"
    DEPTH = 4
    HAS_MOVE_MASKS = False   # mask_src / mask_dst (movement.py:14-15)
    HAS_RESET = False        # reset_mask and the reset modes (reference.py:33-44)

    def __init__(self, config, height: int, width: int, sources, rng: str = "numpy"):
        self.config = LayerConfig.from_reference(config)
        self.height, self.width = int(height), int(width)
        self.sources = list(sources)
        shape = (self.height, self.width)
        self.mask_alpha = load_float_mask(self.config.mask_alpha, shape, 1)            # layer.py:24
        if self.HAS_MOVE_MASKS:
            self.mask_src = load_bool_mask(self.config.mask_src, shape, True)          # movement.py:14
            self.mask_dst = load_bool_mask(self.config.mask_dst, shape, True)          # movement.py:15
        if self.HAS_RESET:
            self.reset_mask = load_float_mask(self.config.reset_mask, shape, 1)        # reference.py:44
            if self.config.reset_mode not in ("off", "random", "constant", "linear"):
                raise ValueError(f"Unknown reset mode {self.config.reset_mode}")       # reference.py:35
        # "numpy": draw the reset field with numpy.random.random on the host, exactly like
        # reference.py:59 (same global stream => same frames for the same numpy seed);
        # "device": counter-based generator on the GPU (no 8 B/px upload)
        self.rng = rng
        self.seed = 0
        self._dev = None
        self._pending_state = None  # (data, rgba) to upload when the device layer is created
        self._sources_dirty = False
        self._deferred = None       # a frame whose kernels wait for render() to run as ONE launch (_defer_step)
        self._alone = False         # set by HipCompositor.update: this is the compositor's only layer

    # ---- device side -------------------------------------------------------------------
    def _layer(self):
        if self._dev is None:
            from .remap import RemapLayer
            c = self.config
            default = lambda a, v: None if np.all(a == v) else a  # noqa: E731  (absent mask == default)
            kw = dict(layer_class=self.LAYER_CLASS, mask_alpha=default(self.mask_alpha, 1))
            if self.HAS_MOVE_MASKS:
                kw.update(transparent_pixels_can_move=c.transparent_pixels_can_move,
                          pixels_can_move_to_empty_spot=c.pixels_can_move_to_empty_spot,
                          pixels_can_move_to_filled_spot=c.pixels_can_move_to_filled_spot,
                          moving_pixels_leave_empty_spot=c.moving_pixels_leave_empty_spot,
                          mask_src=default(self.mask_src, True), mask_dst=default(self.mask_dst, True))
            if self.HAS_RESET:
                kw.update(reset_mode=c.reset_mode, reset_random_factor=c.reset_random_factor,
                          reset_constant_step=c.reset_constant_step, reset_linear_factor=c.reset_linear_factor,
                          reset_source=c.reset_source, reset_mask=default(self.reset_mask, 1))
            if self.LAYER_CLASS == "introduction":
                kw.update({k: getattr(c, k) for k in (
                    "introduce_pixels_on_empty_spots", "introduce_pixels_on_filled_spots", "introduce_moving_pixels",
                    "introduce_unmoving_pixels", "introduce_on_all_filled_spots", "introduce_on_all_empty_spots")})
            self._dev = RemapLayer(self.height, self.width, **kw)
            if self._pending_state is not None:          # restored from a checkpoint
                self._dev.set_state(*self._pending_state)
                self._pending_state = None
            if self.sources:                             # constructor-time sources / masks to (re)install
                self._sources_dirty = True
        if self._sources_dirty:
            self._dev.set_sources([np.asarray(s.introduction_mask, dtype=bool) for s in self.sources])
            self._sources_dirty = False
        return self._dev

    def set_sources(self, sources):
        """Layer.set_sources (layer.py:26-27); for the reference-layer classes also
        reference.py:54-56: write each source's index where its introduction mask is set (also after
        a checkpoint restore, as the reference does: pipeline.py:450-455).  Applied to the device
        state at its next use."""
        self.sources = list(sources)
        self._sources_dirty = True

    def _reset_field(self):
        if self.HAS_RESET and self.config.reset_mode == "random" and self.rng == "numpy":
            return np.random.random(size=(self.height, self.width))                    # reference.py:59
        return None

    def update(self, flow):
        raise NotImplementedError()                                                    # layer.py:29-30

    def _pixmap_beside(self, layer) -> bool:
        """Whether this update's pixmaps go up on the library's upload stream (tf_remap_gather_beside) instead of the
        caller's: when the flow was on the device already -- nothing of this update is on the link yet, and on the
        caller's stream the upload would wait for the update kernel, itself behind the flow source's long kernels -- and
        the frame is large enough for that wait to outweigh a second stream's bookkeeping (one box, alternating runs,
        prefetching source: 4K 618 -> 637 frames/s, 1080p 1945 -> 1694; profiles/r05_host_path_experiments.txt)."""
        return bool(getattr(layer, "flow_was_on_device", False)) and self.height * self.width >= (1 << 22)

    # ---- one launch per frame where a frame allows it ---------------------------------------------------------
    # Compositor.update and .render are two calls (compositor.py:27-40), three launches and a fill here: move + reset,
    # gather, background, paint.  Where the layer is the compositor's only one, a moveref layer with one source, and the
    # flow is on the device, update() only brings the pixmap up and remembers the frame; render() then runs
    # tf_remap_step_dev -- the kernel bench.py times -- as the frame's ONE launch.  Anything that looks at the layer in
    # between (data / rgba, a checkpoint, another update) first runs the remembered frame the ordinary way: same state.
    def _defer_step(self, layer, flow) -> bool:
        if not (self._alone and self.LAYER_CLASS == "moveref" and len(self.sources) == 1
                and getattr(flow, "dev_ptr", None) is not None and not flow.on_host
                and getattr(flow, "in_frame", False)     # (an unclipped flow is checked at update time: RemapLayer.update)
                and tuple(flow.shape) == (self.height, self.width, 2)):
            return False
        if self.HAS_RESET and self.config.reset_mode == "random" and self.rng == "numpy":
            return False                    # the reset field is drawn on the host: it has to go up, the ordinary way
        try:
            pixmap = self.sources[0].next()
        except BaseException:
            # the reference has moved the layer by the time a source runs dry (movement.py:20-60 come before
            # reference.py:99): so has this one when the exception reaches the pipeline (pipeline.py:580-586)
            layer.update(flow, None, self.seed)
            raise
        ptr, channels = layer.stage_pixmap(pixmap, beside=self.height * self.width >= (1 << 22))
        self._deferred = (flow, ptr, channels, self.seed)
        return True

    def _run_deferred(self, comp=None) -> None:
        """comp: render() is asking -- one launch moves, resets, gathers, fills and paints; None: someone looks at the
        layer first -- move + reset and gather as two launches, render() will paint as usual."""
        if self._deferred is None:
            return
        flow, ptr, channels, seed = self._deferred
        self._deferred = None
        layer = self._dev
        flow.wait_on_stream()
        if comp is not None:
            layer.step_dev(comp, flow.dev_ptr, ptr, channels, clip_flow=False, seed=seed)
        else:
            layer.update_dev(flow.dev_ptr, None, seed)
            layer.gather_dev(0, ptr, channels)
        flow.mark_used()
        layer.staged_used()

    def render_into(self, comp):
        self._run_deferred()
        self._layer().render(comp)

    def _state(self, which: int):
        self._run_deferred()
        if self._dev is None:
            if self._pending_state is not None and self._pending_state[which] is not None:
                return self._pending_state[which]
            self._layer()
        return self._dev.get_state()[which]

    @property
    def rgba(self) -> np.ndarray:
        return self._state(1)

    # ---- pickling (checkpoints, pipeline.py:225-242) --------------------------------------
    def __getstate__(self):
        self._run_deferred()
        state = {k: v for k, v in self.__dict__.items() if k not in ("_dev", "_pending_state", "sources", "_deferred")}
        state["sources"] = []   # the pipeline strips sources before pickling (pipeline.py:236-238)
        if self._dev is not None:
            state["_saved_state"] = self._dev.get_state()
        elif self._pending_state is not None:
            state["_saved_state"] = self._pending_state
        return state

    def __setstate__(self, state):
        saved = state.pop("_saved_state", None)
        self.__dict__.update(state)
        self._dev = None
        self._pending_state = saved
        self._sources_dirty = False
        self._deferred = None

    def close(self):
        self._deferred = None
        if self._dev is not None:
            self._dev.close()
            self._dev = None


def _reference_data_layer():
    """transflow's own DataLayer class where that package is importable, else None.  extra/control.py:155-156
    accepts a checkpoint's layer only if `isinstance(layer, DataLayer)`; it then reads nothing but `layer.data`
    and `layer.INDEX_*` (:158-162).  With the reference's class among their bases the layers of this backend
    pass that test in whatever process unpickles them -- the reference's __init__ is never run, none of its
    attributes or methods is used, and where transflow is absent the layers simply do without the base."""
    try:
        from transflow.compositor.layers.data import DataLayer
        return DataLayer
    except Exception:       # not installed, or not importable on this interpreter (cv2, typing.Self ...)
        return None


_REF_DATA_LAYER = _reference_data_layer()


class _HipDataLayer(_HipLayer, *((_REF_DATA_LAYER,) if _REF_DATA_LAYER is not None else ())):
    INDEX_I, INDEX_J, INDEX_ALPHA, INDEX_SOURCE = 0, 1, 2, 3   # data.py:8-12

    @property
    def data(self) -> np.ndarray:
        return self._state(0)


class HipMoveReferenceLayer(_HipDataLayer):
    """The default `moveref` layer (move_reference.py:6-14).  State lives in HBM; `data` / `rgba` are
    downloaded on access (int32 [H,W,4] = (i, j, alpha, source); uint8 [H,W,4])."""

    LAYER_CLASS = "moveref"
    HAS_MOVE_MASKS = True
    HAS_RESET = True

    def update(self, flow):
        """move_reference.py:12-14: MovementLayer.update, then ReferenceLayer.update."""
        layer = self._layer()
        self._run_deferred()
        if self._defer_step(layer, flow):
            return
        layer.update(flow, self._reset_field(), self.seed)
        beside = self._pixmap_beside(layer)
        for i, source in enumerate(self.sources):                                      # reference.py:94-105
            layer.gather(i, source.next(), beside=beside)


class HipSumLayer(_HipDataLayer):
    """`sum` (sum.py:7-14): a reference layer whose (i, j) accumulate floor(flow)."""

    LAYER_CLASS = "sum"
    HAS_RESET = True

    def update(self, flow):
        layer = self._layer()
        layer.update(flow, self._reset_field(), self.seed)                             # sum.py:10 + reference.py:108
        beside = self._pixmap_beside(layer)
        for i, source in enumerate(self.sources):                                      # reference.py:109
            layer.gather(i, source.next(), beside=beside)


class HipStaticLayer(_HipLayer):
    """`static` (static.py:7-17): no data; every update copies each source where it is introduced."""

    LAYER_CLASS = "static"
    DEPTH = 0

    def update(self, flow):
        layer = self._layer()
        for i, source in enumerate(self.sources):                                      # static.py:14-17
            layer.gather(i, source.next())


class HipIntroductionLayer(_HipDataLayer):
    """`introduction` (introduction.py:8-73): int32 [H,W,8] canvas (r, g, b, alpha, source, i, j,
    frame); `rgba` is its first four channels (:65-66)."""

    LAYER_CLASS = "introduction"
    HAS_MOVE_MASKS = True
    DEPTH = 8
    INDEX_I, INDEX_J, INDEX_ALPHA, INDEX_SOURCE = 5, 6, 3, 4                            # introduction.py:11-14

    def __init__(self, *args, **kwargs):
        _HipLayer.__init__(self, *args, **kwargs)
        self.introduced_once = False                                                    # :18

    def update(self, flow):
        layer = self._layer()
        layer.update(flow)                                                              # MovementLayer.update, :69
        if self.config.introduce_once and self.introduced_once:                         # :21-22
            return
        self.introduced_once = True
        for i, source in enumerate(self.sources):                                       # :46-63
            pixmap = source.next()
            layer.introduce(i, pixmap, source.frame_number)


def bind_reference_data_layer() -> bool:
    """For a process that imported this module before transflow was importable (dropin.install() calls it): give
    the data layers the reference's DataLayer base now.  True if they have it afterwards."""
    global _REF_DATA_LAYER
    if _REF_DATA_LAYER is None:
        ref = _reference_data_layer()
        if ref is not None:
            _HipDataLayer.__bases__ = (_HipLayer, ref)
            _REF_DATA_LAYER = ref
    return _REF_DATA_LAYER is not None


LAYER_CLASSES = {"moveref": HipMoveReferenceLayer, "sum": HipSumLayer, "static": HipStaticLayer,
                 "introduction": HipIntroductionLayer}                                  # layer.py:44-56


class HipCompositor:
    """Same constructor, methods and attributes as transflow.compositor.Compositor."""

    def __init__(self, height: int, width: int, layers, background_color: str = "#ffffff", lazy_frames: bool = False):
        """lazy_frames: render() returns a DeviceFrame (transflow_amd/deviceframe.py) -- the uint8 (H, W, 3) array to
        everything numpy, its download started but not waited for, so that frame t comes down (pipeline.py:518,
        output/ffmpeg.py:32-54) beside frame t + 1's uploads and kernels (pipeline.py:565).  Default: a plain ndarray."""
        self.height, self.width = int(height), int(width)
        self.background_color = parse_color(background_color)
        self.background = np.zeros((self.height, self.width, 3), dtype=np.uint8)
        self.background[:, :] = self.background_color
        self.layers = list(layers)
        self.lazy_frames = bool(lazy_frames)
        self._comp = None
        self._comp2 = None          # lazy frames: the second image (frame t downloads from one while t + 1 is rendered into the other)
        self._flip = False
        self._frame_pool = None

    def _image(self):
        from .remap import CompImage
        if self._comp is None:
            self._comp = CompImage(self.height, self.width, self.background_color)
        if not self.lazy_frames:
            return self._comp
        if self._comp2 is None:
            self._comp2 = CompImage(self.height, self.width, self.background_color)
        self._flip = not self._flip
        comp = self._comp2 if self._flip else self._comp
        comp.download_end()         # the frame before last, long in host memory: its image may be written again
        return comp

    def update(self, flow):
        for layer in self.layers:
            layer._alone = len(self.layers) == 1
            layer.update(flow)

    def render(self) -> np.ndarray:
        """compositor.py:31-40: background, then every layer's opaque pixels in order."""
        comp = self._image()
        only = self.layers[0] if len(self.layers) == 1 else None
        if only is not None and getattr(only, "_deferred", None) is not None:
            only._run_deferred(comp)      # the frame's one launch: move, reset, gather, background, paint
        else:
            comp.begin()
            for layer in self.layers:
                layer.render_into(comp)
        if self._frame_pool is None:
            from .device import ArrayPool
            self._frame_pool = ArrayPool((self.height, self.width, 3), np.uint8, pinned=True)
        if self.lazy_frames:
            from .deviceframe import DeviceFrame
            frame = DeviceFrame(comp.download_begin(self._frame_pool.take()), comp)
        else:
            frame = comp.download(self._frame_pool.take())
        return frame

    @classmethod
    def from_args(cls, height: int, width: int, layer_configs, background_color: str = "#ffffff", rng: str = "numpy",
                  lazy_frames: bool = False):
        layers = []
        for config in layer_configs:
            classname = getattr(config, "classname", "moveref")
            if classname not in LAYER_CLASSES:
                raise ValueError(f"Unknown layer classname {classname}")                # layer.py:56
            layers.append(LAYER_CLASSES[classname](config, height, width, [], rng=rng))
        return cls(height, width, layers, background_color=background_color, lazy_frames=lazy_frames)

    def set_sources(self, pixmap_interfaces: dict):
        for i, layer in enumerate(self.layers):
            layer.set_sources(pixmap_interfaces.get(i, []))

    def __getstate__(self):
        return {k: v for k, v in self.__dict__.items() if k not in ("_comp", "_comp2", "_frame_pool")}

    def __setstate__(self, state):
        self.__dict__.update(state)
        self.lazy_frames = bool(state.get("lazy_frames", False))
        self._comp = None
        self._comp2 = None
        self._flip = False
        self._frame_pool = None

    def close(self):
        for layer in self.layers:
            layer.close()
        for name in ("_comp", "_comp2"):
            comp = getattr(self, name, None)
            if comp is not None:
                comp.close()
                setattr(self, name, None)
