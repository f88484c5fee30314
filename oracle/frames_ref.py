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


def bgr_to_grey(frame: np.ndarray, size=None) -> np.ndarray:
    f = np.asarray(frame, np.uint8)
    sh, sw, _ = f.shape
    w, h = (sw, sh) if size is None else size
    ix = np.minimum(np.floor(np.arange(w) * (1.0 / (w / sw))).astype(np.int64), sw - 1)
    iy = np.minimum(np.floor(np.arange(h) * (1.0 / (h / sh))).astype(np.int64), sh - 1)
    r = f[iy][:, ix].astype(np.int64)
    return ((r[:, :, 0] * 3735 + r[:, :, 1] * 19235 + r[:, :, 2] * 9798 + 16384) >> 15).astype(np.uint8)
