"""Parameter objects of the hot path, mirroring the reference's own
(same names, defaults and parsing), so a config built for transflow drives this
backend unchanged.

  LayerConfig   <- transflow/config.py:57-104   (fields the compositor reads)
  FlowConfig    <- CvFlowConfig, transflow/flow/sources/cv.py:271-363 (the fb_* fields)
"""
from __future__ import annotations

import json

_TRUE_WORDS = ("1", "on", "o", "oui", "yes", "y")  # config.py:49-54


def parse_bool_arg(arg, default: bool) -> bool:
    if arg is None:
        return default
    if isinstance(arg, str):
        return arg.lower().strip() in _TRUE_WORDS
    return arg


class LayerConfig:
    """Same constructor and attributes as the reference's LayerConfig."""

    _BOOLS = (
        ("transparent_pixels_can_move", False), ("pixels_can_move_to_empty_spot", True),
        ("pixels_can_move_to_filled_spot", True), ("moving_pixels_leave_empty_spot", False),
        ("reset_source", False), ("introduce_pixels_on_empty_spots", True),
        ("introduce_pixels_on_filled_spots", True), ("introduce_moving_pixels", True),
        ("introduce_unmoving_pixels", True), ("introduce_once", False),
        ("introduce_on_all_filled_spots", False), ("introduce_on_all_empty_spots", False),
    )

    def __init__(self, index: int, classname: str | None = None, mask_alpha: str | None = None,
                 mask_src: str | None = None, mask_dst: str | None = None,
                 transparent_pixels_can_move=None, pixels_can_move_to_empty_spot=None,
                 pixels_can_move_to_filled_spot=None, moving_pixels_leave_empty_spot=None,
                 reset_mode: str | None = None, reset_mask: str | None = None,
                 reset_random_factor: float | None = None, reset_constant_step: float | None = None,
                 reset_linear_factor: float | None = None, reset_source=None,
                 introduce_pixels_on_empty_spots=None, introduce_pixels_on_filled_spots=None,
                 introduce_moving_pixels=None, introduce_unmoving_pixels=None, introduce_once=None,
                 introduce_on_all_filled_spots=None, introduce_on_all_empty_spots=None):
        given = locals()
        self.index = index
        self.classname = "moveref" if classname is None else classname
        self.mask_alpha, self.mask_src, self.mask_dst = mask_alpha, mask_src, mask_dst
        for name, default in self._BOOLS:
            setattr(self, name, parse_bool_arg(given[name], default))
        self.reset_mode = "off" if reset_mode is None else reset_mode
        self.reset_mask = reset_mask
        self.reset_random_factor = 1 if reset_random_factor is None else reset_random_factor
        self.reset_constant_step = 1 if reset_constant_step is None else reset_constant_step
        self.reset_linear_factor = 0.1 if reset_linear_factor is None else reset_linear_factor

    _KEYS = ("index", "classname", "mask_src", "mask_dst", "mask_alpha", "transparent_pixels_can_move",
             "pixels_can_move_to_empty_spot", "pixels_can_move_to_filled_spot", "moving_pixels_leave_empty_spot",
             "reset_mode", "reset_mask", "reset_random_factor", "reset_constant_step", "reset_linear_factor",
             "reset_source", "introduce_pixels_on_empty_spots", "introduce_pixels_on_filled_spots",
             "introduce_moving_pixels", "introduce_unmoving_pixels", "introduce_once",
             "introduce_on_all_filled_spots", "introduce_on_all_empty_spots")

    def todict(self) -> dict:
        return {k: getattr(self, k) for k in self._KEYS}

    @classmethod
    def fromdict(cls, d: dict):
        kw = {k: d[k] for k in cls._KEYS if k in d and k != "index"}
        kw.setdefault("classname", "reference")  # config.py:110
        return cls(d["index"], **kw)

    @classmethod
    def from_reference(cls, cfg):
        """Accepts a transflow.config.LayerConfig (or anything with the same attributes)."""
        if isinstance(cfg, cls):
            return cfg
        return cls(cfg.index, **{k: getattr(cfg, k) for k in cls._KEYS if k != "index" and hasattr(cfg, k)})


class FlowConfig:
    """The Farnebäck fields of CvFlowConfig (cv.py:273-281) and their defaults; other
    methods' fields (hs_*, lk_*) are accepted and carried so a CvFlowConfig JSON loads."""

    FB_DEFAULTS = dict(fb_pyr_scale=0.5, fb_levels=3, fb_winsize=15, fb_iterations=3, fb_poly_n=5,
                       fb_poly_sigma=1.2, fb_flags=0)

    def __init__(self, method: str = "farneback", **kwargs):
        if method != "farneback":
            raise ValueError(f"transflow_amd implements the 'farneback' method only, got {method!r}")
        self.method = method
        for k, v in self.FB_DEFAULTS.items():
            setattr(self, k, kwargs.pop(k, v))
        # This backend's own key in a CvFlowConfig JSON: "hip_exact_sums": true asks for the box window summed in
        # OpenCV's own order (library option fb_exact_sums: flows bit-identical to the CPU path's, about 1.5 times
        # the Farnebäck time for a single 4K pair).  The reference ignores keys it does not know only if they are not there: leave it out
        # of files the reference itself must read.
        self.hip_exact_sums = parse_bool_arg(kwargs.pop("hip_exact_sums", None), False)
        # "hip_prefetch": n > 0 lets the flow source run up to n flows ahead of its consumer in a worker thread with a
        # library stream of its own (what the reference gets from running the source in a child process behind a
        # queue, pipeline.py:56-64, for a source used in-process).  Its position attributes then run ahead by as much.
        self.hip_prefetch = int(kwargs.pop("hip_prefetch", 0) or 0)
        # "hip_device_flows": true -- the source yields DeviceFlow objects (transflow_amd/deviceflow.py): flows that stay in
        # HBM until something reads them on the host, and that HipCompositor.update takes by device address (no 66 MB per
        # 4K frame down the link and up again across pipeline.py:562-567).  "ipc": the same, and through a multiprocessing
        # queue (pipeline.py:85-86) such a flow travels as a 64-byte HIP IPC handle instead of the pickled array.
        # "hip_batch": n > 1 -- where nothing can look at a raw flow in between (no lock expressions, no convolution kernel,
        # no initial flow), the source reads n frames ahead and computes their n pairs in ONE Farneback call: a single 4K
        # pair leaves most of the chip idle at the coarse levels, a batch of four costs 0.6 of four single calls.  Flows
        # still come out one at a time, in order, each post-processed with its own t.
        self.hip_batch = max(1, int(kwargs.pop("hip_batch", 1) or 1))
        v = kwargs.pop("hip_device_flows", None)
        self.hip_device_flows = "ipc" if isinstance(v, str) and v.lower() == "ipc" else parse_bool_arg(v, False)
        self.extra = dict(kwargs)  # hs_*, lk_*, show_window ...: not used by this backend

    def fb_kwargs(self) -> dict:
        return dict(pyr_scale=self.fb_pyr_scale, levels=self.fb_levels, winsize=self.fb_winsize,
                    iterations=self.fb_iterations, poly_n=self.fb_poly_n, poly_sigma=self.fb_poly_sigma,
                    flags=self.fb_flags)

    def to_dict(self) -> dict:
        d = {"method": self.method}
        d.update({k: getattr(self, k) for k in self.FB_DEFAULTS})
        d.update(self.extra)
        if self.hip_exact_sums:
            d["hip_exact_sums"] = True
        if self.hip_prefetch:
            d["hip_prefetch"] = self.hip_prefetch
        if self.hip_device_flows:
            d["hip_device_flows"] = self.hip_device_flows
        if self.hip_batch > 1:
            d["hip_batch"] = self.hip_batch
        return d

    def to_file(self, path: str):
        with open(path, "w", encoding="utf8") as f:
            json.dump(self.to_dict(), f, indent=4)

    @classmethod
    def from_file(cls, path: str):
        with open(path, "r", encoding="utf8") as f:
            return cls(**json.load(f))

    @classmethod
    def from_reference(cls, cfg):
        if cfg is None:
            return cls()
        if isinstance(cfg, cls):
            return cfg
        return cls(**{k: getattr(cfg, k) for k in cls.FB_DEFAULTS if hasattr(cfg, k)})
