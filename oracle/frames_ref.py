"""CPU oracle for the frame ingest step: cv2.resize(INTER_NEAREST) then
cv2.cvtColor(COLOR_BGR2GRAY), transflow/flow/sources/cv.py:461-466.

TEST INFRASTRUCTURE ONLY.  PARITY UNPINNED: cv2 is not installable in this environment and the
reference's tests hold no pixel values for this step, so this restates OpenCV 4.x's published 8-bit
arithmetic from memory: Y = (B*3735 + G*19235 + R*9798 + 2^14) >> 15 (the BT.601 weights in 15-bit
fixed point); nearest-neighbour source index min(floor(dst_index * (1 / (dst_size / src_size))),
src_size - 1) per axis.
"""
from __future__ import annotations

import numpy as np


# The two fixed-point forms OpenCV has shipped for 8-bit BGR -> grey: (weights of B, G, R; shift)
WEIGHTS = {
    "15bit": (3735, 19235, 9798, 15),     # OpenCV 4.x (the parity target here, recalled)
    "14bit": (1868, 9617, 4899, 14),      # OpenCV 2 / 3 (a documented variant: what "unpinned" can cost, see below)
}


def bgr_to_grey(frame: np.ndarray, size=None, weights: str = "15bit") -> np.ndarray:
    """`weights`: "15bit" is the statement the HIP kernel is compared with; "14bit" is the other rounding a build of
    OpenCV could apply ((B*1868 + G*9617 + R*4899 + 2^13) >> 14).  On uniformly random pixels the two differ in
    about 0.27 % of them, always by one grey level (tests/test_oracle_frames.py counts it): that bounds what this
    unpinned step can cost downstream -- a one-level change of a few pixels in a thousand of the Farneback input."""
    wb, wg, wr, shift = WEIGHTS[weights]
    f = np.asarray(frame, np.uint8)
    sh, sw, _ = f.shape
    w, h = (sw, sh) if size is None else size
    ix = np.minimum(np.floor(np.arange(w) * (1.0 / (w / sw))).astype(np.int64), sw - 1)
    iy = np.minimum(np.floor(np.arange(h) * (1.0 / (h / sh))).astype(np.int64), sh - 1)
    r = f[iy][:, ix].astype(np.int64)
    return ((r[:, :, 0] * wb + r[:, :, 1] * wg + r[:, :, 2] * wr + (1 << (shift - 1))) >> shift).astype(np.uint8)
