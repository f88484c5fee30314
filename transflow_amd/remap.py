"""Remap handles: thin objects over tf_remap_* / tf_comp_* of libtfhip.so."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import _lib
from ._lib import TfLayerCfg, check

RESET_MODES = {"off": 0, "random": 1, "constant": 2, "linear": 3}  # reference.py:16-21
LAYER_CLASSES = {"moveref": 0, "sum": 1, "static": 2, "introduction": 3}  # layer.py:44-56


def _ptr(a):
    return None if a is None else C.c_void_p(a.ctypes.data)


class CompImage:
    """Background + output frame of Compositor (compositor/compositor.py:17-40)."""

    def __init__(self, height: int, width: int, background_rgb=(255, 255, 255), image_dev: int | None = None):
        """image_dev: device address of H*W*3 bytes the caller owns (and keeps alive) for the image --
        the frames of a batch side by side in one buffer leave in one gather."""
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.height, self.width = int(height), int(width)
        bg = (C.c_uint8 * 3)(*[int(v) & 255 for v in background_rgb])
        if image_dev is None:
            check(self._lib.tf_comp_create(C.byref(self._h), self.height, self.width, bg))
        else:
            check(self._lib.tf_comp_create_on(C.byref(self._h), self.height, self.width, bg, C.c_void_p(image_dev)))

    def begin(self):
        check(self._lib.tf_comp_begin(self._h))

    def download(self, out: np.ndarray | None = None) -> np.ndarray:
        if out is None:
            out = np.empty((self.height, self.width, 3), np.uint8)
        check(self._lib.tf_comp_download(self._h, _ptr(out)))
        return out

    def download_begin(self, out: np.ndarray) -> np.ndarray:
        """The download started on the library's download stream behind what the caller's stream has been given;
        `out` (page-locked, C-contiguous uint8 (H, W, 3)) is filled once download_end() has returned.  The image
        must not be rendered into again before that."""
        if out.dtype != np.uint8 or not out.flags.c_contiguous or out.shape != (self.height, self.width, 3):
            raise ValueError("download_begin needs a C-contiguous uint8 array of shape (H, W, 3)")
        check(self._lib.tf_comp_download_begin(self._h, _ptr(out)))
        return out

    def download_end(self) -> None:
        if getattr(self, "_h", None) is not None and self._h.value:
            check(self._lib.tf_comp_download_end(self._h))

    def image_ptr(self) -> int:
        p = C.c_void_p()
        check(self._lib.tf_comp_image_ptr(self._h, C.byref(p)))
        return p.value

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.tf_comp_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class RemapLayer:
    """Device state and kernels of one compositor layer (`layer_class`: moveref, sum, static or
    introduction -- layer.py:44-56)."""

    def __init__(self, height: int, width: int, *, layer_class="moveref", transparent_pixels_can_move=False,
                 pixels_can_move_to_empty_spot=True, pixels_can_move_to_filled_spot=True,
                 moving_pixels_leave_empty_spot=False, reset_mode="off", reset_random_factor=1.0,
                 reset_constant_step=1.0, reset_linear_factor=0.1, reset_source=False,
                 introduce_pixels_on_empty_spots=True, introduce_pixels_on_filled_spots=True,
                 introduce_moving_pixels=True, introduce_unmoving_pixels=True,
                 introduce_on_all_filled_spots=False, introduce_on_all_empty_spots=False,
                 mask_src=None, mask_dst=None, mask_alpha=None, reset_mask=None):
        self._lib = _lib.load()
        self._h = C.c_void_p()
        self.height, self.width = int(height), int(width)
        if reset_mode not in RESET_MODES:
            raise ValueError(f"Unknown reset mode {reset_mode}")  # reference.py:35
        if layer_class not in LAYER_CLASSES:
            raise ValueError(f"Unknown layer classname {layer_class}")  # layer.py:56
        self.layer_class = layer_class
        self.depth = {"static": 0, "introduction": 8}.get(layer_class, 4)
        cfg = TfLayerCfg(int(bool(transparent_pixels_can_move)), int(bool(pixels_can_move_to_empty_spot)),
                         int(bool(pixels_can_move_to_filled_spot)), int(bool(moving_pixels_leave_empty_spot)),
                         RESET_MODES[reset_mode], float(reset_random_factor), float(reset_constant_step),
                         float(reset_linear_factor), int(bool(reset_source)), LAYER_CLASSES[layer_class],
                         int(bool(introduce_pixels_on_empty_spots)), int(bool(introduce_pixels_on_filled_spots)),
                         int(bool(introduce_moving_pixels)), int(bool(introduce_unmoving_pixels)),
                         int(bool(introduce_on_all_filled_spots)), int(bool(introduce_on_all_empty_spots)))
        shape = (self.height, self.width)

        def mask(a, dtype):
            if a is None:
                return None
            a = np.ascontiguousarray(a, dtype=dtype)
            if a.shape != shape:
                raise ValueError(f"mask shape {a.shape} != {shape}")
            return a

        ms, md = mask(mask_src, np.uint8), mask(mask_dst, np.uint8)
        ma, rm = mask(mask_alpha, np.float32), mask(reset_mask, np.float32)
        check(self._lib.tf_remap_create(C.byref(self._h), self.height, self.width, C.byref(cfg), _ptr(ms), _ptr(md),
                                        _ptr(ma), _ptr(rm)))

    def set_sources(self, introduction_masks) -> None:
        masks = [np.ascontiguousarray(m, dtype=np.uint8) for m in introduction_masks]
        for m in masks:
            if m.shape != (self.height, self.width):
                raise ValueError("introduction mask has the wrong shape")
        arr = (C.c_void_p * max(1, len(masks)))(*[m.ctypes.data for m in masks])
        check(self._lib.tf_remap_set_sources(self._h, len(masks), arr))

    def update(self, flow: np.ndarray, uniform: np.ndarray | None = None, seed: int = 0) -> None:
        if getattr(flow, "dev_ptr", None) is not None and not flow.on_host:
            # a DeviceFlow nobody brought down (transflow_amd/deviceflow.py): the kernels read it where it is.  A flow a
            # source's post_process clipped (`in_frame`) cannot leave the frame and is only queued; any other is checked
            # here, at the price of one synchronisation, so that the reference's IndexError (movement.py:33, 39) comes
            # out of update() as it does for a host array (round 6; before, it came at the next render()).
            if tuple(flow.shape) != (self.height, self.width, 2):
                raise ValueError(f"flow shape {tuple(flow.shape)} != {(self.height, self.width, 2)}")
            u_dev = None
            if uniform is not None:
                from .device import DevBuffer
                u = np.ascontiguousarray(uniform, dtype=np.float64)
                if u.shape != (self.height, self.width):
                    raise ValueError("uniform field has the wrong shape")
                if getattr(self, "_u_dev", None) is None:
                    self._u_dev = DevBuffer(u.nbytes)
                self._u_dev.upload(u)
                u_dev = self._u_dev.ptr
            flow.wait_on_stream()
            self.update_dev(flow.dev_ptr, u_dev, seed)
            flow.mark_used()
            self.flow_was_on_device = True
            if not getattr(flow, "in_frame", False) and self.out_of_frame():
                raise IndexError("a rounded flow vector leaves the frame (run post_process first)")
            return
        self.flow_was_on_device = False
        flow = np.asarray(flow)
        if flow.dtype != np.float32 and np.issubdtype(flow.dtype, np.floating):
            # a float64 flow (post_process after a float64 convolution kernel returns one, source.py:344-348):
            # the reference rounds THAT array (numpy.round, movement.py:21; numpy.floor, sum.py:10), and a
            # value like 1.4999999999 rounds differently once cast to float32.  Round in the flow's own
            # type; the integers that come out are exact in float32 and the kernels' rint / floor keep them.
            flow = np.floor(flow) if self.layer_class == "sum" else np.rint(flow)
        flow = np.ascontiguousarray(flow, dtype=np.float32)
        if flow.shape != (self.height, self.width, 2):
            raise ValueError(f"flow shape {flow.shape} != {(self.height, self.width, 2)}")
        u = None
        if uniform is not None:
            u = np.ascontiguousarray(uniform, dtype=np.float64)
            if u.shape != (self.height, self.width):
                raise ValueError("uniform field has the wrong shape")
        check(self._lib.tf_remap_update(self._h, _ptr(flow), _ptr(u), C.c_uint64(seed & (2**64 - 1))))

    def update_dev(self, flow_dev: int, uniform_dev: int | None = None, seed: int = 0) -> None:
        check(self._lib.tf_remap_update_dev(self._h, C.c_void_p(flow_dev),
                                            C.c_void_p(uniform_dev) if uniform_dev else None,
                                            C.c_uint64(seed & (2**64 - 1))))

    def uniform_dev(self, seed: int, out_dev: int) -> None:
        """The float64 (H, W) field the next update / step_dev with no `uniform` and this seed will draw."""
        check(self._lib.tf_remap_uniform_dev(self._h, C.c_uint64(seed & (2**64 - 1)), C.c_void_p(out_dev)))

    def out_of_frame(self) -> bool:
        v = C.c_int()
        check(self._lib.tf_remap_check(self._h, C.byref(v)))
        return bool(v.value)

    def gather(self, source_index: int, pixmap: np.ndarray, beside: bool = False) -> None:
        """beside: the pixmap goes up on the library's upload stream, beside the update queued before it
        (tf_remap_gather_beside: for updates whose flow was on the device already)."""
        pm = np.ascontiguousarray(pixmap, dtype=np.uint8)
        if pm.ndim != 3 or pm.shape[:2] != (self.height, self.width):
            raise ValueError(f"pixmap shape {pm.shape} does not match the layer")
        fn = self._lib.tf_remap_gather_beside if beside else self._lib.tf_remap_gather
        check(fn(self._h, int(source_index), _ptr(pm), int(pm.shape[2])))

    def stage_pixmap(self, pixmap: np.ndarray, beside: bool = False):
        """The upload of gather() alone (tf_remap_stage_pixmap): (device address, channels) of the staged pixmap, for
        gather_dev / step_dev later on this thread's stream; staged_used() after the last kernel that reads it."""
        pm = np.ascontiguousarray(pixmap, dtype=np.uint8)
        if pm.ndim != 3 or pm.shape[:2] != (self.height, self.width):
            raise ValueError(f"pixmap shape {pm.shape} does not match the layer")
        p = C.c_void_p()
        check(self._lib.tf_remap_stage_pixmap(self._h, _ptr(pm), int(pm.shape[2]), int(bool(beside)), C.byref(p)))
        return p.value, int(pm.shape[2])

    def staged_used(self) -> None:
        check(self._lib.tf_remap_staged_used(self._h))

    def introduce(self, source_index: int, pixmap: np.ndarray, frame_number: int) -> None:
        """Introduction layer: one iteration of introduction.py:46-63."""
        pm = np.ascontiguousarray(pixmap, dtype=np.uint8)
        if pm.ndim != 3 or pm.shape[:2] != (self.height, self.width):
            raise ValueError(f"pixmap shape {pm.shape} does not match the layer")
        check(self._lib.tf_remap_introduce(self._h, int(source_index), _ptr(pm), int(pm.shape[2]), int(frame_number)))

    def gather_dev(self, source_index: int, pixmap_dev: int, channels: int) -> None:
        check(self._lib.tf_remap_gather_dev(self._h, int(source_index), C.c_void_p(pixmap_dev), int(channels)))

    def step_dev(self, comp: "CompImage", flow_dev: int, pixmap_dev: int, channels: int = 3, clip_flow=False,
                 uniform_dev: int | None = None, seed: int = 0) -> None:
        """update + gather(source 0) + begin + render in one call (one kernel when possible).
        clip_flow: False / True (BACKWARD post_process, the clip alone, folded in) / 2 (flow_dev is the
        winner map of Farneback.post_process_scatter: the rest of FORWARD post_process folded in)."""
        check(self._lib.tf_remap_step_dev(self._h, comp._h, C.c_void_p(flow_dev), int(clip_flow),
                                          C.c_void_p(uniform_dev) if uniform_dev else None,
                                          C.c_uint64(seed & (2**64 - 1)), C.c_void_p(pixmap_dev), int(channels)))

    def steps_dev(self, comps, flows_dev, pixmaps_dev, channels: int = 3, clip_flow=False, uniforms_dev=None, seed: int = 0) -> None:
        """len(comps) consecutive step_dev calls as one (tf_remap_steps_dev): step i takes flows_dev[i] and paints comps[i]
        from pixmaps_dev[i] (one pointer for all steps may be given as an int).  Same state and frames as the single
        calls; the library leaves out the rgba stores nothing reads (tfhip.h)."""
        n = len(comps)
        if isinstance(pixmaps_dev, int):
            pixmaps_dev = [pixmaps_dev] * n
        if len(flows_dev) != n or len(pixmaps_dev) != n or (uniforms_dev is not None and len(uniforms_dev) != n):
            raise ValueError("steps_dev: one flow, one pixmap (and one uniform field) per compositor image")
        arr = lambda ptrs: (C.c_void_p * n)(*[C.c_void_p(int(p)) for p in ptrs])
        check(self._lib.tf_remap_steps_dev(self._h, n, arr([c._h.value for c in comps]),
                                           arr(flows_dev), int(clip_flow), None if uniforms_dev is None else arr(uniforms_dev),
                                           C.c_uint64(seed & (2**64 - 1)), arr(pixmaps_dev), int(channels)))

    def render(self, comp: CompImage) -> None:
        check(self._lib.tf_remap_render(self._h, comp._h))

    def get_state(self):
        """(data, rgba): data int32 [H,W,depth] (None for a static layer); rgba uint8 [H,W,4], or the
        int32 view data[..., :4] for an introduction layer (introduction.py:65-66)."""
        data = np.empty((self.height, self.width, self.depth), np.int32) if self.depth else None
        if self.layer_class == "introduction":
            check(self._lib.tf_remap_get_state(self._h, _ptr(data), None))
            return data, data[:, :, :4]
        rgba = np.empty((self.height, self.width, 4), np.uint8)
        check(self._lib.tf_remap_get_state(self._h, _ptr(data), _ptr(rgba)))
        return data, rgba

    def set_state(self, data=None, rgba=None) -> None:
        if self.layer_class == "introduction":
            rgba = None  # a view of data
        if self.depth == 0:
            data = None
        d = None if data is None else np.ascontiguousarray(data, dtype=np.int32)
        r = None if rgba is None else np.ascontiguousarray(rgba, dtype=np.uint8)
        if d is not None and d.shape != (self.height, self.width, self.depth):
            raise ValueError("data has the wrong shape")
        if r is not None and r.shape != (self.height, self.width, 4):
            raise ValueError("rgba has the wrong shape")
        check(self._lib.tf_remap_set_state(self._h, _ptr(d), _ptr(r)))

    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._lib.tf_remap_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
