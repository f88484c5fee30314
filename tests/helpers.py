"""Shared helpers for the parity tests (oracle side + golden fixtures)."""
import glob
import os

import numpy as np

from oracle import remap_ref

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def layer_case_files():
    return sorted(glob.glob(os.path.join(GOLDEN, "remap_layer_*.npz")))


def _parse(v):
    if v in ("True", "False"):
        return v == "True"
    try:
        return float(v)
    except ValueError:
        return v


def case_cfg(z):
    """cfg kwargs stored by tools/capture_golden.py as two string arrays."""
    return {str(k): _parse(str(v)) for k, v in zip(z["cfg_keys"], z["cfg_vals"])}


PRM_KEYS = ("transparent_pixels_can_move", "pixels_can_move_to_empty_spot",
            "pixels_can_move_to_filled_spot", "moving_pixels_leave_empty_spot",
            "reset_mode", "reset_random_factor", "reset_constant_step",
            "reset_linear_factor", "reset_source")


def oracle_params(cfg):
    return remap_ref.LayerParams(**{k: v for k, v in cfg.items() if k in PRM_KEYS})


def synth_pair(h, w, seed=1234, shift=(3.0, 2.0), noise=6.0):
    """SURVEY.md §8(d) synthetic frame pair: multi-scale sine texture + noise;
    frame B is A's texture evaluated at smoothly displaced coordinates."""
    rng = np.random.default_rng(seed)
    a = rng.uniform(0.4, 1.0, 6)
    fx = rng.uniform(0.004, 0.06, 6)
    fy = rng.uniform(0.004, 0.06, 6)
    ph = rng.uniform(0, 2 * np.pi, 6)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)

    def tex(x, y):
        v = np.zeros_like(x)
        for m in range(6):
            v += a[m] * np.sin(2 * np.pi * (fx[m] * x + fy[m] * y) + ph[m])
        return 128 + 40 * v / 2.0

    u = shift[0] * np.sin(2 * np.pi * yy / h * 2)
    v = shift[1] * np.cos(2 * np.pi * xx / w * 3)
    n0 = np.random.default_rng(seed + 1).normal(0, 1, (h, w)) * noise
    n1 = np.random.default_rng(seed + 2).normal(0, 1, (h, w)) * noise
    fa = np.clip(np.rint(tex(xx, yy) + n0), 0, 255).astype(np.uint8)
    fb = np.clip(np.rint(tex(xx - u, yy - v) + n1), 0, 255).astype(np.uint8)
    return fa, fb


def layer2_case_files():
    """sum / static / introduction layer recurrences (tools/capture_golden.py --layers2-only)."""
    return sorted(glob.glob(os.path.join(GOLDEN, "layer2_*.npz")))


INTRO_KEYS = ("introduce_pixels_on_empty_spots", "introduce_pixels_on_filled_spots", "introduce_moving_pixels",
              "introduce_unmoving_pixels", "introduce_once", "introduce_on_all_filled_spots",
              "introduce_on_all_empty_spots")


def oracle_layer2(z):
    """The oracle's layer object for one layer2_* fixture."""
    cfg = case_cfg(z)
    h, w, ns = int(z["h"]), int(z["w"]), int(z["nsources"])
    intro = [z[f"intro_{s}"] for s in range(ns)]
    cls = str(z["classname"])
    if cls == "sum":
        return remap_ref.SumLayer(h, w, oracle_params(cfg), mask_alpha=z["mask_alpha"], reset_mask=z["reset_mask"],
                                  introduction_masks=intro)
    if cls == "static":
        return remap_ref.StaticLayer(h, w, mask_alpha=z["mask_alpha"], introduction_masks=intro)
    if cls == "introduction":
        prm = remap_ref.IntroParams(**{k: v for k, v in cfg.items() if k in PRM_KEYS + INTRO_KEYS})
        return remap_ref.IntroductionLayer(h, w, prm, mask_src=z["mask_src"], mask_dst=z["mask_dst"],
                                           mask_alpha=z["mask_alpha"], introduction_masks=intro)
    raise ValueError(cls)


def capture_frame_numbers(prm, t, ns):
    """Frame numbers the capture's sources reported at frame t (FakeSource.frame_number = number of
    next() calls - 1): one call per frame in which the layer introduced, i.e. every frame, or only
    the first with introduce_once."""
    return [0 if prm.introduce_once else t] * ns
