"""Algorithmic-byte model of the hot path -- the single
source of the byte counts bench.py's `roofline` object uses.

"Stage-once" accounting on the reference's own array types: every logical stage
reads each input once and writes each output once; gathers count once; no halo
re-reads and no credit for fusion.  N0 = full-resolution pixels, Nk = pixels of
pyramid scale k (k = 0..K), I = iterations.
"""
from __future__ import annotations

HBM_PEAK_GBS = 8000.0          # MI355X HBM3E spec peak (MI355X_MICROARCH.md, chip table)
HBM_COPY_CEILING_GBS = 6290.0  # measured float4 copy on MI355X (same table)


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


FUSE_MIN_PX = 4_000_000   # the library's default threshold (TF_FB_FUSE_MIN_PX): pixels of a level over the batch


def level_is_fused(nk: int, pairs: int) -> bool:
    """Whether a level's iterations run as k_flow_iter_pc (one kernel) or as update_matrices + blur_solve."""
    import os
    forced = os.environ.get("TF_FB_FUSED")
    if forced is not None and int(forced) >= 0:
        return int(forced) > 0
    return nk * pairs >= int(os.environ.get("TF_FB_FUSE_MIN_PX", FUSE_MIN_PX))


# per-launch algorithmic bytes of each kernel, per pixel of the level it runs on
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
    if name == "fb_flow_iter":                                 # S4+S5 in one kernel (TF_FB_FUSED=1)
        return kernel_bytes("fb_update_matrices", n0, nk, nc, pairs) + pairs * 28 * nk
    raise KeyError(name)
