"""Switch an importable transflow installation over to this backend without editing it.

    import transflow_amd.dropin as dropin
    dropin.install()          # before transflow.pipeline builds its sources
    ...                       # transflow runs as usual; pipeline.py is unchanged
    dropin.uninstall()

`install()` replaces the two factories pipeline.py calls
(transflow/pipeline.py:325 FlowSource.from_args, :445 Compositor.from_args) with
dispatchers that build HipFlowSource / HipCompositor when the request is one this
backend serves -- a video path (or webcam index) with the Farnebäck method (flow mask and the
scale/threshold/clip filters, the convolution kernel and per-pixel polar expressions included), layers of
any of the reference's classes (`moveref`, `sum`, `static`, `introduction`) -- and fall through to
the reference's own factory otherwise (motion vectors, other flow methods, polar expressions that are
not per-pixel formulas).
`.flow.zip` archives are served too (their flows are post-processed on the GPU).
INTEGRATION.md shows the three-line patch a maintainer would add instead.
"""
from __future__ import annotations

import os

_saved = {}


def _flow_from_args(original):
    from .config import FlowConfig
    from .flow import HipFlowSource

    def from_args(cls, flow_path, use_mvs=False, mask_path=None, kernel_path=None, cv_config=None,
                  flow_filters=None, size=None, direction=None, seek_ckpt=None, seek_time=None,
                  duration_time=None, repeat=1, lock_expr=None, lock_mode="stay"):
        served = isinstance(flow_path, str) and not use_mvs and cv_config != "window"
        if served and flow_filters is not None and "polar" in flow_filters:
            try:                                   # polar expressions the device cannot run stay the reference's
                from .flow import FlowFilter
                for part in flow_filters.strip().split(";"):
                    FlowFilter.from_string(part)
            except NotImplementedError:
                served = False
        if served and cv_config is not None and os.path.isfile(cv_config):
            try:
                FlowConfig.from_file(cv_config)
            except ValueError:      # another flow method: the reference's own source
                served = False
        if not served:
            return original(flow_path, use_mvs=use_mvs, mask_path=mask_path, kernel_path=kernel_path,
                            cv_config=cv_config, flow_filters=flow_filters, size=size, direction=direction,
                            seek_ckpt=seek_ckpt, seek_time=seek_time, duration_time=duration_time, repeat=repeat,
                            lock_expr=lock_expr, lock_mode=lock_mode)
        return HipFlowSource.from_args(flow_path, use_mvs, mask_path, kernel_path, cv_config, flow_filters, size,
                                       direction, seek_ckpt, seek_time, duration_time, repeat, lock_expr, lock_mode)

    return classmethod(from_args)


def _compositor_from_args(original, lazy_frames=False):
    from .compositor import LAYER_CLASSES, HipCompositor

    def from_args(cls, height, width, layer_configs, background_color="#ffffff"):
        if all(getattr(c, "classname", None) in LAYER_CLASSES for c in layer_configs):
            return HipCompositor.from_args(height, width, layer_configs, background_color=background_color,
                                           lazy_frames=lazy_frames)
        return original(height, width, layer_configs, background_color=background_color)

    return classmethod(from_args)


def install(flow: bool = True, compositor: bool = True, lazy_frames: bool = False) -> None:
    """Needs `transflow` importable.  Idempotent.  lazy_frames: the compositors built for the pipeline return
    DeviceFrames from render() (transflow_amd/deviceframe.py): the pipeline's `oq.put(frame)` (pipeline.py:518-522) then
    pickles the frame -- and waits for its download -- in the queue's feeder thread, beside the next update."""
    if flow and "flow" not in _saved:
        from transflow.flow.sources.source import FlowSource as RefFlowSource
        _saved["flow"] = (RefFlowSource, RefFlowSource.__dict__["from_args"])
        RefFlowSource.from_args = _flow_from_args(RefFlowSource.from_args)
    if compositor and "compositor" not in _saved:
        from transflow.compositor.compositor import Compositor as RefCompositor

        from .compositor import bind_reference_data_layer
        bind_reference_data_layer()    # extra/control.py:155 asks isinstance(layer, DataLayer) of checkpointed layers
        _saved["compositor"] = (RefCompositor, RefCompositor.__dict__["from_args"])
        RefCompositor.from_args = _compositor_from_args(RefCompositor.from_args, lazy_frames)


def uninstall() -> None:
    for key in list(_saved):
        cls, original = _saved.pop(key)
        cls.from_args = original
