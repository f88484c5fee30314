"""ctypes binding of libtfhip.so (include/tfhip.h).

The HIP library is the product: there is no CPU fallback.  Importing this
module does not touch the GPU (the reference forks its flow source into a child
process, pipeline.py:56-64); the first call does.  A missing or unloadable
library raises ImportError loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
# TFHIP_LIBRARY: developer override to load an alternate build of the same library (kernel experiments)
LIB_PATH = os.environ.get("TFHIP_LIBRARY") or os.path.join(_HERE, "libtfhip.so")

TF_OK = 0
TF_ERR_ARG = -1
TF_ERR_HIP = -2
TF_ERR_INDEX = -3
TF_ERR_STATE = -4
TF_ERR_UNSUPPORTED = -5


class TfFbParams(C.Structure):
    _fields_ = [("pyr_scale", C.c_double), ("levels", C.c_int), ("winsize", C.c_int),
                ("iterations", C.c_int), ("poly_n", C.c_int), ("poly_sigma", C.c_double),
                ("flags", C.c_int)]


class TfPolarStep(C.Structure):
    _fields_ = [("op", C.c_int), ("wide", C.c_int), ("imm", C.c_double)]


class TfFlowOp(C.Structure):
    _fields_ = [("kind", C.c_int), ("wide", C.c_int), ("value", C.c_double)]


class TfLayerCfg(C.Structure):
    _fields_ = [("transparent_pixels_can_move", C.c_int), ("pixels_can_move_to_empty_spot", C.c_int),
                ("pixels_can_move_to_filled_spot", C.c_int), ("moving_pixels_leave_empty_spot", C.c_int),
                ("reset_mode", C.c_int), ("reset_random_factor", C.c_double),
                ("reset_constant_step", C.c_double), ("reset_linear_factor", C.c_double),
                ("reset_source", C.c_int), ("layer_class", C.c_int),
                ("introduce_pixels_on_empty_spots", C.c_int), ("introduce_pixels_on_filled_spots", C.c_int),
                ("introduce_moving_pixels", C.c_int), ("introduce_unmoving_pixels", C.c_int),
                ("introduce_on_all_filled_spots", C.c_int), ("introduce_on_all_empty_spots", C.c_int)]


_P = C.c_void_p
_PP = C.POINTER(C.c_void_p)
_I = C.c_int
_PI = C.POINTER(C.c_int)

# name -> (restype, argtypes); every symbol include/tfhip.h declares
PROTOTYPES = {
    "tf_abi_version": (_I, []),
    "tf_init": (_I, [_I]),
    "tf_is_initialized": (_I, []),
    "tf_device_count": (_I, [_PI]),
    "tf_set_option": (_I, [C.c_char_p, C.c_long]),
    "tf_get_option": (_I, [C.c_char_p, C.POINTER(C.c_long)]),
    "tf_last_error": (C.c_char_p, []),
    "tf_sync": (_I, []),
    "tf_stream": (_I, [_PP]),
    "tf_event_create": (_I, [_PP]),
    "tf_event_record": (_I, [_P]),
    "tf_event_elapsed_ms": (_I, [_P, _P, C.POINTER(C.c_float)]),
    "tf_event_destroy": (None, [_P]),
    "tf_stream_wait_event": (_I, [_P]),
    "tf_event_synchronize": (_I, [_P]),
    "tf_ipc_export": (_I, [_P, _P]),
    "tf_ipc_open": (_I, [_P, _PP]),
    "tf_ipc_close": (_I, [_P]),
    "tf_prof_enable": (_I, [_I]),
    "tf_prof_set_filter": (_I, [C.c_char_p]),
    "tf_prof_reset": (_I, []),
    "tf_prof_report": (_I, [C.c_char_p, C.c_size_t]),
    "tf_dev_alloc": (_I, [_PP, C.c_size_t]),
    "tf_dev_free": (_I, [_P]),
    "tf_dev_upload": (_I, [_P, _P, C.c_size_t]),
    "tf_dev_download": (_I, [_P, _P, C.c_size_t]),
    "tf_dev_copy": (_I, [_P, _P, C.c_size_t]),
    "tf_dev_store_u64": (_I, [_P, C.c_uint64]),
    "tf_dev_stream_copy": (_I, [_P, _P, C.c_size_t]),
    "tf_fb_create": (_I, [_PP, _I, _I, C.POINTER(TfFbParams), _I, _I]),
    "tf_host_alloc": (_I, [_PP, C.c_size_t]),
    "tf_host_free": (_I, [_P]),
    "tf_thread_stream": (_I, [_I]),
    "tf_fb_create_lane": (_I, [_PP, _P]),
    "tf_fb_set_exact": (_I, [_P, _I]),
    "tf_fb_async_io": (_I, [_P, _I]),
    "tf_fb_get_flow_begin": (_I, [_P, _I, _P, _PI]),
    "tf_fb_get_flow_end": (_I, [_P, _I]),
    "tf_fb_destroy": (None, [_P]),
    "tf_fb_calc": (_I, [_P, _P, C.c_ssize_t, _P, C.c_ssize_t, _P]),
    "tf_fb_set_frame": (_I, [_P, _I, _P, C.c_ssize_t]),
    "tf_fb_set_frame_bgr": (_I, [_P, _I, _P, _I, _I, C.c_ssize_t]),
    "tf_fb_frame_ptr": (_I, [_P, _I, _PP]),
    "tf_fb_set_initial_flow": (_I, [_P, _I, _P]),
    "tf_fb_initial_flow_ptr": (_I, [_P, _I, _PP]),
    "tf_fb_stage_initial_flow": (_I, [_P, _P, _P]),
    "tf_fb_calc_slots": (_I, [_P, _I, _PI, _PI]),
    "tf_fb_get_flow": (_I, [_P, _I, _P]),
    "tf_fb_flow_ptr": (_I, [_P, _I, _PP]),
    "tf_fb_keep_expansions": (_I, [_P, _I]),
    "tf_fb_post_process": (_I, [_P, _I, _I]),
    "tf_fb_post_process_scatter": (_I, [_P, _I, _PP]),
    "tf_fb_post_process_host": (_I, [_P, _P, _I]),
    "tf_fb_post_process_ex": (_I, [_P, _I, _I, _I, C.POINTER(TfFlowOp), _P]),
    "tf_fb_post_process_host_ex": (_I, [_P, _P, _I, _I, C.POINTER(TfFlowOp), _P]),
    "tf_fb_stage_level_image": (_I, [_P, _P, C.c_ssize_t, _I, _P]),
    "tf_fb_stage_polyexp": (_I, [_P, _P, _I, _I, _P]),
    "tf_fb_stage_level_polyexp": (_I, [_P, _P, C.c_ssize_t, _I, _P]),
    "tf_fb_stage_update_matrices": (_I, [_P, _P, _P, _P, _I, _I, _P]),
    "tf_fb_stage_upsampled_matrices": (_I, [_P, _I, _P, _P, _P, _P]),
    "tf_fb_stage_blur_solve": (_I, [_P, _P, _I, _I, _P]),
    "tf_fb_level_count": (_I, [_P, _PI]),
    "tf_fb_level_size": (_I, [_P, _I, _PI, _PI]),
    "tf_remap_create": (_I, [_PP, _I, _I, C.POINTER(TfLayerCfg), _P, _P, _P, _P]),
    "tf_remap_destroy": (None, [_P]),
    "tf_remap_set_sources": (_I, [_P, _I, C.POINTER(C.c_void_p)]),
    "tf_remap_update": (_I, [_P, _P, _P, C.c_uint64]),
    "tf_remap_update_dev": (_I, [_P, _P, _P, C.c_uint64]),
    "tf_remap_uniform_dev": (_I, [_P, C.c_uint64, _P]),
    "tf_remap_check": (_I, [_P, _PI]),
    "tf_remap_gather": (_I, [_P, _I, _P, _I]),
    "tf_remap_gather_dev": (_I, [_P, _I, _P, _I]),
    "tf_remap_gather_beside": (_I, [_P, _I, _P, _I]),
    "tf_remap_stage_pixmap": (_I, [_P, _P, _I, _I, _PP]),
    "tf_remap_staged_used": (_I, [_P]),
    "tf_flow_merge_dev": (_I, [_I, _I, _PP, _P, C.c_size_t]),
    "tf_flow_upscale_dev": (_I, [_P, _P, _I, _I, _I, _I]),
    "tf_flow_convolve_dev": (_I, [_P, _P, _I, _I, _I, _P, _I, _I]),
    "tf_flow_post_process_dev": (_I, [_P, _I, _I, _I, _I, _P]),
    "tf_flow_polar_dev": (_I, [_P, C.c_size_t, _I, _P, _I, _P, _I, _I]),
    "tf_flow_render1d_dev": (_I, [_P, _P, C.c_size_t, C.c_float, C.POINTER(C.c_float), _I]),
    "tf_flow_render2d_dev": (_I, [_P, _P, C.c_size_t, C.c_float, C.POINTER(C.c_float)]),
    "tf_frame_grey_dev": (_I, [_P, _I, _I, _P, _I, _I]),
    "tf_remap_introduce": (_I, [_P, _I, _P, _I, _I]),
    "tf_remap_introduce_dev": (_I, [_P, _I, _P, _I, _I]),
    "tf_remap_data_depth": (_I, [_P, _PI]),
    "tf_remap_render": (_I, [_P, _P]),
    "tf_remap_step_dev": (_I, [_P, _P, _P, _I, _P, C.c_uint64, _P, _I]),
    "tf_remap_steps_dev": (_I, [_P, _I, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p), _I, C.POINTER(C.c_void_p), C.c_uint64,
                                C.POINTER(C.c_void_p), _I]),
    "tf_remap_get_state": (_I, [_P, _P, _P]),
    "tf_remap_set_state": (_I, [_P, _P, _P]),
    "tf_comp_create": (_I, [_PP, _I, _I, C.POINTER(C.c_uint8)]),
    "tf_comp_create_on": (_I, [_PP, _I, _I, C.POINTER(C.c_uint8), _P]),
    "tf_comp_destroy": (None, [_P]),
    "tf_comp_begin": (_I, [_P]),
    "tf_comp_download": (_I, [_P, _P]),
    "tf_comp_download_begin": (_I, [_P, _P]),
    "tf_comp_download_end": (_I, [_P]),
    "tf_comp_image_ptr": (_I, [_P, _PP]),
    "tf_batch_unique_id": (_I, [_P]),
    "tf_batch_init": (_I, [_PP, _I, _I, _P]),
    "tf_batch_destroy": (None, [_P]),
    "tf_batch_info": (_I, [_P, _PI, _PI, _PI]),
    "tf_batch_broadcast": (_I, [_P, _P, C.c_size_t, _I]),
    "tf_batch_gather": (_I, [_P, _P, C.c_size_t, _P, C.POINTER(C.c_size_t), _I]),
    "tf_batch_gather_at": (_I, [_P, _P, C.c_size_t, _P, C.POINTER(C.c_size_t), C.POINTER(C.c_size_t), C.c_size_t, _I]),
    "tf_batch_gather_begin": (_I, [_P, _P, C.c_size_t, _P, C.POINTER(C.c_size_t), _I]),
    "tf_batch_gather_end": (_I, [_P]),
    "tf_batch_reduce": (_I, [_P, C.POINTER(C.c_double), _I, _I]),
}

_lib = None


def load() -> C.CDLL:
    """Loads libtfhip.so and binds every prototype.  No GPU call is made."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "or `make -C transflow_amd/csrc` (hipcc, gfx950). transflow_amd has no CPU fallback.")
    try:
        lib = C.CDLL(LIB_PATH)
    except OSError as err:
        raise ImportError(f"cannot load {LIB_PATH}: {err}") from err
    for name, (restype, argtypes) in PROTOTYPES.items():
        try:
            fn = getattr(lib, name)
        except AttributeError as err:
            raise ImportError(f"{LIB_PATH} does not export {name}; rebuild it") from err
        fn.restype = restype
        fn.argtypes = argtypes
    _lib = lib
    return lib


def set_option(name: str, value: int) -> None:
    """tf_set_option: a documented run-time option of the library (include/tfhip.h)."""
    check(load().tf_set_option(name.encode(), int(value)))


def get_option(name: str) -> int:
    v = C.c_long()
    check(load().tf_get_option(name.encode(), C.byref(v)))
    return v.value


def profile(on: bool, name_filter: str | None = None, reset: bool = True) -> None:
    """tf_prof_*: HIP events around every launch whose label contains `name_filter` (all if None)."""
    lib = load()
    check(lib.tf_prof_set_filter(name_filter.encode() if name_filter else None))
    if reset:
        check(lib.tf_prof_reset())
    check(lib.tf_prof_enable(1 if on else 0))


def profile_report() -> dict:
    """label -> (launches, milliseconds) since the last reset."""
    buf = C.create_string_buffer(1 << 16)
    check(load().tf_prof_report(buf, len(buf)))
    out = {}
    for line in buf.value.decode().splitlines():
        name, cnt, ms = line.split()
        out[name] = (int(cnt), float(ms))
    return out


class TfError(RuntimeError):
    """A HIP/runtime failure inside libtfhip.so."""


def check(rc: int) -> None:
    """Maps tf_status to the exception the reference's code would have raised
    at that point, so pipeline.py's handlers fire (Errors)."""
    if rc == TF_OK:
        return
    msg = (load().tf_last_error() or b"").decode("utf8", "replace")
    if rc == TF_ERR_ARG:
        raise ValueError(msg)
    if rc == TF_ERR_INDEX:
        raise IndexError(msg)
    if rc == TF_ERR_UNSUPPORTED:
        raise NotImplementedError(msg)
    raise TfError(f"tfhip error {rc}: {msg}")
