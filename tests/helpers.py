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


# Farneback shapes / parameter sets shared by the GPU parity tests and the oracle's envelope test
FB_CASES = [
    ((270, 480), dict()),                                   # transflow defaults (cv.py:273-281)
    ((480, 854), dict()),                                   # configs[0] geometry (River.mp4 854x480)
    ((135, 241), dict(levels=2)),                           # odd sizes: non-integer resize ratios
    ((200, 260), dict(levels=0)),                           # one scale
    ((96, 128), dict(levels=5, winsize=9, iterations=2, poly_n=7, poly_sigma=1.5)),
    ((40, 50), dict(levels=3)),                             # below min_size: K = 0
    ((64, 300), dict(levels=1, pyr_scale=0.8)),
]


FB_SWEEP = [
    ((360, 642), dict(levels=4, pyr_scale=0.7)),                 # non-dyadic pyramid: every level resized with fractions
    ((358, 639), dict(levels=3, pyr_scale=0.5, winsize=21)),     # odd sizes, window half-width 10
    ((240, 320), dict(levels=2, winsize=25, iterations=5)),      # half-width 12, more iterations
    ((300, 400), dict(levels=3, winsize=5, iterations=1)),       # half-width 2, a single iteration
    ((270, 482), dict(levels=5, poly_n=7, poly_sigma=1.5)),      # width % 4 != 0: no split level images
    ((540, 960), dict(levels=5, poly_n=5, poly_sigma=1.1)),      # quarter-4K: the bench's level structure
    ((128, 4096), dict(levels=2)),                               # wide and flat
    ((2048, 64), dict(levels=1)),                                # tall and narrow
    ((90, 130), dict(levels=1, pyr_scale=0.3)),                  # a big step between two scales
]


# ---- fixtures made by tools/pin_with_cv2.py on a machine with a real OpenCV (none is committed until one was) ----------

def cv2_fixture_files(directory=None):
    return sorted(glob.glob(os.path.join(directory or GOLDEN, "farneback_cv2_*.npz")))


def cv2_fixture_cases(path):
    """(meta, [(case dict, prev, next, initial flow or None, cv2's flow)], skipped) of one fixture.  The inputs are
    regenerated from the stored seeds; a case whose regenerated frames do not have the stored CRCs is listed in
    `skipped` (another numpy drawing other numbers), never compared."""
    import json
    import zlib
    z = np.load(path)
    meta = json.loads(str(z["meta_json"]))
    out, skipped = [], []
    for c in meta["cases"]:
        a, b = synth_pair(c["h"], c["w"], seed=c["seed"])
        if (zlib.crc32(a.tobytes()) & 0xFFFFFFFF, zlib.crc32(b.tobytes()) & 0xFFFFFFFF) != (c["crc_prev"], c["crc_next"]):
            skipped.append(c["key"])
            continue
        init = z[c["initial_flow"]] if c.get("initial_flow") else None
        out.append((c, a, b, init, z[c["key"]]))
    return meta, out, skipped


def cv2_fixture_grey(path):
    """(bgr frame, [(width, height, cv2's grey frame)]) of one fixture, or None when the frame does not regenerate."""
    import json
    import zlib
    z = np.load(path)
    g = json.loads(str(z["meta_json"]))["grey"]
    bgr = np.random.default_rng(g["seed"]).integers(0, 256, tuple(g["shape"]) + (3,), dtype=np.uint8)
    if zlib.crc32(bgr.tobytes()) & 0xFFFFFFFF != g["crc_bgr"]:
        return None
    return bgr, [(o["width"], o["height"], z[o["key"]]) for o in g["outputs"]]
