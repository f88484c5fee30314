"""Byte models of the hot path -- the single source of the byte counts bench.py's `roofline` object uses.

Two accountings, both per launch / per step, both stated in DESIGN.md §5:

* **model** ("stage-once" on the reference's own array types): every logical
  stage of the reference reads each input once and writes each output once; no credit for fusion,
  both frames of every pair expanded.  Bytes of the *reference's work*; dividing them by a kernel's
  time gives a work rate, not HBM utilisation (`roofline.model_work_rate`).
* **built**: the bytes the kernels of this library must move as built -- the one-kernel iteration never
  stores M (R0 20 + R1 20 + flow in 8 + flow out 8 B/px), a frame two pairs share is expanded once,
  the level image of levels 0 and 1 never leaves the CU.  `roofline.achieved` / `frac` use these.

N0 = full-resolution pixels, Nk = pixels of pyramid scale k (k = 0..K), I = iterations, P = pairs
per pass, F = P + 1 frames per pass.
"""
from __future__ import annotations

import json
import os

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip table)
HBM_COPY_CEILING_GBS = 6290.0  # measured float4 copy on MI355X (same table)

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def level_sizes(width: int, height: int, pyr_scale: float, levels: int):
    """OpenCV's scale schedule: [(Wk, Hk)] for k = 0..K."""
    k, scale = 0, 1.0
    while k < levels:
        scale *= pyr_scale
        if width * scale < 32 or height * scale < 32:
            break
        k += 1
    out = []
    for lvl in range(k + 1):
        s = 1.0
        for _ in range(lvl):
            s *= pyr_scale
        out.append((int(round_half_even(width * s)), int(round_half_even(height * s))))
    return out


def round_half_even(v: float) -> int:
    return int(round(v))  # python's round() is half-to-even, like cvRound


def farneback_bytes(width, height, pyr_scale=0.5, levels=3, iterations=3) -> int:
    """B_fb = 2(K+1)N0 + (64+96I) sum(Nk) + 8 sum_{k>=1} Nk   (Appendix C)."""
    sizes = level_sizes(width, height, pyr_scale, levels)
    n = [w * h for w, h in sizes]
    k1 = len(n)
    return 2 * k1 * n[0] + (64 + 96 * iterations) * sum(n) + 8 * sum(n[1:])


def remap_bytes(width, height, reset_mask=False, external_uniform=False, forward=False) -> int:
    """46 B/px moveref with one RGB source (+4 reset mask, +8 supplied u, +8 FORWARD)."""
    per_px = 46 + (4 if reset_mask else 0) + (8 if external_uniform else 0) + (8 if forward else 0)
    return per_px * width * height


FUSE_MIN_PX = 4_000_000   # the library's default (tf_set_option "fb_fuse_min_px"): pixels of a level over the batch


def level_is_fused(nk: int, pairs: int, fused: int = -1, fuse_min_px: int = FUSE_MIN_PX) -> bool:
    """Whether a level's iterations run as k_flow_iter_pc (one kernel) or as update_matrices + blur_solve.
    `fused` / `fuse_min_px` = the library options "fb_fused" (-1 auto, 0 never, 1 always) and "fb_fuse_min_px"."""
    if fused >= 0:
        return fused > 0
    return nk * pairs >= fuse_min_px


# ---- model: per-launch stage-once bytes of each kernel -------------------------------------------
# (n0 = full-res pixels, nk = level pixels, nc = pixels of the next coarser level)
def kernel_bytes(name: str, n0: int, nk: int, nc: int, pairs: int) -> int:
    if name == "fb_level_image":                               # S1: u8 frame in, f32 level out, x2 images
        return pairs * 2 * (n0 + 4 * nk)
    if name == "fb_level_rowpass":                             # S1 of a long-kernel level, first half: the frame read
        return pairs * 2 * n0
    if name == "fb_level_colpass":                             # ... second half: the level image written
        return pairs * 2 * 4 * nk
    if name == "fb_polyexp":                                   # S2 x2 images
        return pairs * 2 * 24 * nk
    if name == "fb_update_matrices":                           # S4 (+S3: upsample read + level flow write)
        return pairs * (68 * nk + (8 * nc + 8 * nk if nc else 8 * nk))
    if name == "fb_blur_solve":                                # S5
        return pairs * 28 * nk
    if name == "fb_level_polyexp":                             # S1+S2 in one kernel: stage-once sum of both
        return pairs * 2 * (n0 + 4 * nk + 24 * nk)
    if name == "fb_flow_iter":                                 # S4+S5 in one kernel
        return kernel_bytes("fb_update_matrices", n0, nk, nc, pairs) + pairs * 28 * nk
    raise KeyError(name)


def _levels(width, height, levels):
    n = [a * b for a, b in level_sizes(width, height, 0.5, levels)]
    return n, level_sizes(width, height, 0.5, levels)


def model_kernel_bytes(name, width, height, levels, pairs, iterations=3, fused=-1) -> int:
    """Model bytes one step (one pass of `pairs` pairs) moves through kernel `name`, all its launches."""
    n, _ = _levels(width, height, levels)
    total = 0
    for k in range(len(n)):
        nc = n[k + 1] if k + 1 < len(n) else 0
        mult = 1
        if name in ("fb_level_rowpass", "fb_level_colpass") and k < 2:   # long blur kernels start at level 2 (pyr_scale 0.5)
            continue
        if name == "fb_level_polyexp" and k >= 2:
            continue
        if name == "fb_polyexp" and k < 2:
            continue
        if name == "fb_flow_iter" and not level_is_fused(n[k], pairs, fused):
            continue
        if name in ("fb_update_matrices", "fb_blur_solve") and level_is_fused(n[k], pairs, fused):
            continue
        if name == "fb_blur_solve":
            mult = iterations
        elif name in ("fb_update_matrices", "fb_flow_iter"):
            # first launch of a level carries S3 (flow init); the I-1 rebuilds are plain S4 (+S5 when fused)
            total += (iterations - 1) * pairs * (68 if name == "fb_update_matrices" else 96) * n[k]
        total += mult * kernel_bytes(name, n[0], n[k], nc, pairs)
    return total


# ---- built: what the kernels of this library must move ---------------------------------------------
def built_kernel_bytes(name, width, height, levels, pairs, iterations=3, fused=-1) -> int:
    """Bytes one step must move through kernel `name` as built (all its launches)."""
    n, sizes = _levels(width, height, levels)
    frames = pairs + 1                                         # consecutive pairs share a frame: expanded once
    total = 0
    if name.startswith("remap_step"):
        return 0
    for k in range(len(n)):
        nc = n[k + 1] if k + 1 < len(n) else 0
        is_fused = level_is_fused(n[k], pairs, fused)
        if name == "fb_flow_iter" and is_fused:
            # R0 20 + R1 20 (bilinear gather, counted once) + flow in 8 + flow out 8; the first iteration of
            # a level reads the coarser level's flow (8 nc) instead of its own (there is none at the coarsest)
            first = 48 * n[k] + (8 * nc if nc else 0)
            total += pairs * (first + (iterations - 1) * 56 * n[k])
        elif name == "fb_update_matrices" and not is_fused:
            # first: R0 20 + R1 20 + M 20 (+ coarse flow 8 nc); rebuilds: + flow 8.  After the last iteration
            # no rebuild: `iterations` launches in all
            total += pairs * (60 * n[k] + (8 * nc if nc else 0) + (iterations - 1) * 68 * n[k])
        elif name == "fb_blur_solve" and not is_fused:
            total += pairs * iterations * 28 * n[k]
        elif name == "fb_level_polyexp" and k < 2:             # A1+A2 in one kernel: frame bytes in, R out
            total += frames * (n[0] + 20 * n[k])
        elif name == "fb_level_rowpass" and k >= 2:            # frame rows read once for all these levels,
            if k == 2:                                         # one float plane [H][2 Wk] written per level
                total += frames * n[0]
            total += frames * 8 * sizes[0][1] * sizes[k][0]
        elif name == "fb_level_colpass" and k >= 2:
            total += frames * (8 * sizes[0][1] * sizes[k][0] + 4 * n[k])
        elif name == "fb_polyexp" and k >= 2:
            total += frames * 24 * n[k]
    return total


def built_flow_iter_level_bytes(width, height, levels, pairs, iterations=3, fused=-1) -> dict:
    """built_kernel_bytes("fb_flow_iter") level by level: {level: bytes of its `iterations` launches} (fused levels only)."""
    n, _ = _levels(width, height, levels)
    out = {}
    for k in range(len(n)):
        nc = n[k + 1] if k + 1 < len(n) else 0
        if level_is_fused(n[k], pairs, fused):
            out[k] = pairs * (48 * n[k] + (8 * nc if nc else 0) + (iterations - 1) * 56 * n[k])
    return out


BUILT_FB_KERNELS = ("fb_flow_iter", "fb_update_matrices", "fb_blur_solve", "fb_level_polyexp", "fb_level_rowpass",
                    "fb_level_colpass", "fb_polyexp")


def built_remap_bytes(width, height, reset_mask=False, forward=False) -> int:
    """The one-kernel remap step per frame: flow 8 (FORWARD: winner map 4, plus the scatter pass that makes
    it: flow 8 read + map 4 initialised + 4 claimed), layer state read at the source and written -- one 32-bit word
    per pixel for frames up to 8192 x 8192 (4 + 4; round 5), int16 x 4 beyond (8 + 8) --, pixmap 3, rgba 4, RGB frame 3,
    reset mask 4."""
    state = 8 if width <= 8192 and height <= 8192 else 16
    per_px = (4 + 16 if forward else 8) + state + 3 + 4 + 3 + (4 if reset_mask else 0)
    return per_px * width * height


def built_step_bytes(width, height, levels, pairs, iterations=3, reset_mask=False, forward=False, fused=-1) -> int:
    """A pass of `pairs` pairs.  BACKWARD: its remap steps are ONE tf_remap_steps_dev call, which stores the layer's rgba
    (4 B/px) in the last step alone while every pixel is selected (the bench's layers: one source, random reset)."""
    fb = sum(built_kernel_bytes(k, width, height, levels, pairs, iterations, fused) for k in BUILT_FB_KERNELS)
    rm = pairs * built_remap_bytes(width, height, reset_mask, forward)
    if not forward and pairs >= 2:
        rm -= (pairs - 1) * 4 * width * height
    return fb + rm


# ---- counters ------------------------------------------------------------------------------------
def profile_traffic(name, width, height, levels, pairs, launches_per_step, iterations=3, fused=-1):
    """HBM bytes per launch of kernel `name` from the PMC passes recorded under profiles/ (separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE runs, corrected as profiles/README.md says): bytes per level
    pixel measured at 4K level 0, scaled to the pixels this workload's launches cover.  A PROFILE
    CONSTANT, not a measurement of the run that quotes it.  None when the kernel was never measured."""
    table = None
    for fname in ("r05_traffic.json", "r04_traffic.json", "r03_traffic.json", "r02_traffic.json", "r01_traffic.json"):
        path = os.path.join(_ROOT, "profiles", fname)
        if os.path.exists(path):
            with open(path) as f:
                t = json.load(f)
            if name in t:
                table = t
                break
    if table is None:
        return None
    n, _ = _levels(width, height, levels)
    per_level_launches = {"fb_update_matrices": iterations, "fb_blur_solve": iterations,
                          "fb_flow_iter": iterations}.get(name, 1)
    if name == "fb_flow_iter":
        n = [v for v in n if level_is_fused(v, pairs, fused)]
    images = 2 if name in ("fb_polyexp", "fb_level_image") else 1
    entry = table[name]
    # a batch of consecutive pairs reads the frame two pairs share once per XCD (the iteration kernel's work order):
    # measured at the bench's own batch where the table has it
    batched = table.get(name + "_batched")
    if batched and pairs >= batched.get("pairs", 1 << 30) // 2:
        entry = batched
    total = entry["bytes_per_px"] * sum(n) * pairs * images * per_level_launches
    return total / max(1.0, launches_per_step)


def profile_step_traffic(workload, width, height, levels, pairs):
    """HBM bytes of one whole step by the counters (profiles/r06_traffic_step.json: tools/traffic_step.py over separate
    rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes at the bench's own pass size), with the per-kernel table's path -- or
    None when the table was made for another workload or another number of pairs per pass (no scaling: a step's
    traffic depends on which levels fill the chip and on what consecutive pairs share in L2).  A PROFILE CONSTANT."""
    for fname in ("r06_traffic_step.json",):
        path = os.path.join(_ROOT, "profiles", fname)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            t = json.load(f)
        if (t.get("width"), t.get("height"), t.get("levels"), t.get("pairs")) == (width, height, levels, pairs) \
                and t.get("workload") == workload:
            return {"bytes_per_step": t["bytes_per_step"], "table": "profiles/" + fname,
                    "kernels_GB_per_step": {k: round(v["bytes_per_step"] / 1e9, 4) for k, v in t["kernels"].items()},
                    "kernels_bytes_per_step": {k: float(v["bytes_per_step"]) for k, v in t["kernels"].items()},
                    "kernels_launches_per_step": {k: int(v["launches_per_step"]) for k, v in t["kernels"].items()}}
    return None
