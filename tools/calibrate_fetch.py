#!/usr/bin/env python3
"""Known-traffic launches for calibrating rocprofv3's FETCH_SIZE / WRITE_SIZE on the access
widths this library uses (MI355X_MICROARCH.md: only 16 B/lane streams are calibrated).

    rocprofv3 --pmc FETCH_SIZE --kernel-trace --output-format csv -d out -- python tools/calibrate_fetch.py

  k_pp_clip      : reads N*8 B (float2 per lane) and writes N*8 B           -> 8 B/lane loads
  k_remap_init   : writes N*16 B (int4 per lane), reads nothing              -> 16 B/lane stores
  comp_fill      : writes N*3 B with byte stores
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from transflow_amd._lib import set_option  # noqa: E402
from transflow_amd.device import sync  # noqa: E402
from transflow_amd.farneback import BACKWARD, Farneback  # noqa: E402
from transflow_amd.remap import CompImage, RemapLayer  # noqa: E402

h, w = 2160, 3840
fb = Farneback(w, h, levels=0, frame_slots=2, max_pairs=1)
a = np.zeros((h, w), np.uint8)
fb.set_frame(0, a)
fb.set_frame(1, a)
fb.calc_slots([0], [1])                   # the fused iteration kernel (8.3M pixels >= its threshold)
set_option("fb_fused", 0)
fb2 = Farneback(w, h, levels=0, frame_slots=2, max_pairs=1)   # the same as two kernels per iteration
fb2.set_frame(0, a)
fb2.set_frame(1, a)
fb2.calc_slots([0], [1])
set_option("fb_fused", -1)
# the exact mode's kernels: column sums straight from the expansions + row walker, then through M in memory
set_option("fb_exact_sums", 1)
for fused in (1, 0):
    set_option("fb_fused", fused)
    fbx = Farneback(w, h, levels=0, frame_slots=2, max_pairs=1)
    fbx.set_frame(0, a)
    fbx.set_frame(1, a)
    fbx.calc_slots([0], [1])
    fbx.close()
set_option("fb_fused", -1)
set_option("fb_exact_sums", 0)
for _ in range(5):
    fb.post_process(0, BACKWARD)          # k_pp_clip: N*8 read, N*8 written
layer = RemapLayer(h, w)                   # k_remap_init: N*16 written
comp = CompImage(h, w)                     # k_comp_fill: N*3 written
for _ in range(4):
    comp.begin()
sync()
print("pixels", h * w, "pp_clip bytes r/w", h * w * 8, "remap_init bytes w", h * w * 16, "comp_fill bytes w", h * w * 3)
