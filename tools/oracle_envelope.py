#!/usr/bin/env python3
"""How far do builds a real OpenCV could be sit from the scalar statement in oracle/farneback_ref.c?  (CPU only.)

The parity target of the Farneback half is the default build of the oracle: the scalar statements of optflowgf.cpp
with no FMA contraction -- and it is unpinned (no cv2 here, no flow values in the reference: DESIGN.md section 4).  This
tool measures the sensitivity variants of oracle/Makefile (`make -C oracle variants`) against it:

  fma      the same source with -ffp-contract=fast -mfma (what an AVX2/FMA wheel's bodies contract)
  area     [VERIFY] 4: INTER_AREA's scalar-tail order at every column of an exactly half-size level
  polyf32  FarnebackPolyExp's horizontal part accumulated in float with FMA (the cheaper GPU form round 5's review
           asked to be evaluated; this is its CPU model)

on the test suites' shapes, on 1080p levels = 5 and on pairs of bench.py's 4K clip, and prints per case the largest
deviation in units of the tolerance 1e-4 * max(1, max|ref|), the pixels beyond a quarter of it and beyond it, and where
those lie.   usage: tools/oracle_envelope.py [--no-4k] > profiles/r06_oracle_envelope.txt
`--clip-sample N`: only N pairs of the 4K clip, evenly spaced, and how many of them each variant moves beyond the tolerance
(how often the border discontinuity of DESIGN.md section 4 is met): > profiles/r06_oracle_envelope_clip_sample.txt"""
import os
import sys
import time
from concurrent.futures import ThreadPoolExecutor

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench  # noqa: E402
from oracle import farneback as F  # noqa: E402
from tests.helpers import FB_CASES, FB_SWEEP, synth_pair  # noqa: E402


BEYOND = {}      # variant -> cases with a pixel beyond the tolerance


def report(name, prev, nxt, kw):
    with ThreadPoolExecutor(4) as ex:                      # ctypes releases the GIL: the four builds side by side
        futs = {v: ex.submit(F.calc, prev, nxt, variant=v, **kw) for v in (None,) + F.VARIANTS}
        res = {v: f.result() for v, f in futs.items()}
    ref = res[None]
    tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
    cells = []
    for v in F.VARIANTS:
        d = np.abs(res[v] - ref).max(axis=2)
        far = d > tol
        cell = f"{v}: max {d.max() / tol:6.3f} tol, > tol/4: {int((d > tol / 4).sum())}, > tol: {int(far.sum())}"
        if far.any():
            BEYOND[v] = BEYOND.get(v, 0) + 1
            ys, xs = np.nonzero(far)
            cell += f" (rows {ys.min()}-{ys.max()}, columns {xs.min()}-{xs.max()} of {ref.shape[0]} x {ref.shape[1]})"
        cells.append(cell)
    print(f"{name:58s} tol {tol:.3g} | " + " | ".join(cells), flush=True)


def main():
    t0 = time.time()
    if "--clip-sample" in sys.argv:
        n = int(sys.argv[sys.argv.index("--clip-sample") + 1])
        clip = bench.ClipSynth(2160, 3840, 256, 2000)
        pairs = sorted({round(j * 254 / (n - 1)) for j in range(n)})
        for t in pairs:
            report(f"bench 4k clip, pair {t} (BACKWARD order), levels=5", clip.frame(t + 1), clip.frame(t), dict(levels=5))
        print(f"# {len(pairs)} pairs of the 4K clip: pairs with a pixel beyond the tolerance -- " +
              ", ".join(f"{v}: {BEYOND.get(v, 0)}" for v in F.VARIANTS) + f"; {time.time() - t0:.0f} s")
        return
    for (h, w), kw in FB_CASES + FB_SWEEP:
        a, b = synth_pair(h, w, seed=70)
        report(f"{w}x{h} {kw}", a, b, kw)
    a, b = synth_pair(1080, 1920, seed=70)
    report("1920x1080 levels=5", a, b, dict(levels=5))
    if "--no-4k" not in sys.argv:
        clip = bench.ClipSynth(2160, 3840, 256, 2000)      # the 4k workload's clip (bench.Job: seed 2000)
        for t in (0, 63, 127):
            report(f"bench 4k clip, pair {t} (BACKWARD order), levels=5", clip.frame(t + 1), clip.frame(t), dict(levels=5))
    print(f"# {time.time() - t0:.0f} s")


if __name__ == "__main__":
    main()
