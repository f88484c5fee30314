#!/usr/bin/env python3
"""Pin the unpinned half of the parity story to a real OpenCV -- on any machine that has one.

STAND-ALONE: needs numpy and cv2 only and imports nothing from this repository, so it can be copied to wherever
`import cv2` works (neither the build container nor the GPU box of this project has it).  It regenerates the test
suites' synthetic frame pairs from their seeds, runs the one call the reference makes on the path,

    cv2.calcOpticalFlowFarneback(prev, next, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags)
    (reference: transflow/flow/sources/cv.py:479-490; parameter defaults cv.py:273-281)

on them with flags 0 / 4 / 256 / 260, the ingest step cv2.resize(INTER_NEAREST) + cv2.cvtColor(COLOR_BGR2GRAY)
(cv.py:461-466) on a seeded BGR frame, and writes

    farneback_cv2_<cv2 version>.npz

to commit under tests/golden/.  tests/test_oracle_farneback.py (the CPU oracle), tests/test_gpu_farneback.py (the HIP
library, default and exact mode) and tests/test_oracle_flow_ops.py (BGR -> grey) consume every such file they find and
skip with a reason when there is none.  The file holds DATA only: seeds, parameters, CRCs of the regenerated inputs (a
consumer whose numpy draws other frames from the same seeds skips instead of failing), the flows, the grey frames and
the lines of cv2.getBuildInformation() that say which SIMD / FMA code paths the build dispatches to.

    usage:  python pin_with_cv2.py [output directory]        (default: the current directory)
"""
import json
import os
import sys
import zlib

import numpy as np

FORMAT = 1

# (height, width), keyword parameters: the shapes of tests/helpers.py FB_CASES (the GPU suites' whole-call cases)
CASES = [
    ((270, 480), dict()),
    ((480, 854), dict()),
    ((135, 241), dict(levels=2)),
    ((200, 260), dict(levels=0)),
    ((96, 128), dict(levels=5, winsize=9, iterations=2, poly_n=7, poly_sigma=1.5)),
    ((40, 50), dict(levels=3)),
    ((64, 300), dict(levels=1, pyr_scale=0.8)),
]
DEFAULTS = dict(pyr_scale=0.5, levels=3, winsize=15, iterations=3, poly_n=5, poly_sigma=1.2)   # cv.py:273-281
FLAG_CASE = 0            # the case that is also run with flags 4, 256 and 260
SEED = 70
GREY_SEED, GREY_SHAPE = 4242, (123, 217)
GREY_SIZES = [(217, 123), (160, 90), (301, 77), (64, 200)]     # (width, height) targets of the nearest resize


def synth_pair(h, w, seed=1234, shift=(3.0, 2.0), noise=6.0):
    """tests/helpers.py synth_pair, restated so that this file stands alone (same draws from the same seed)."""
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


def crc(a):
    return zlib.crc32(np.ascontiguousarray(a).tobytes()) & 0xFFFFFFFF


def build_lines(cv2):
    """The lines of the build information a reader needs to judge rounding: CPU baseline / dispatch, FMA, parallel
    framework, version control revision."""
    keep = ("Version control", "Baseline", "Dispatched", "requested", "required", "FMA", "AVX", "NEON", "SSE",
            "Parallel framework", "C++ flags (Release)", "Platform", "Host:")
    out = []
    for line in cv2.getBuildInformation().splitlines():
        if any(k in line for k in keep):
            out.append(line.strip())
    return out


def main(argv=None):
    import cv2

    argv = sys.argv[1:] if argv is None else argv
    out_dir = argv[0] if argv else "."
    arrays, cases = {}, []
    for i, ((h, w), kw) in enumerate(CASES):
        prm = dict(DEFAULTS, **kw)
        a, b = synth_pair(h, w, seed=SEED)
        args = (prm["pyr_scale"], prm["levels"], prm["winsize"], prm["iterations"], prm["poly_n"], prm["poly_sigma"])
        first = cv2.calcOpticalFlowFarneback(a, b, None, *args, 0)
        runs = [(0, first)]
        if i == FLAG_CASE:
            for flags in (cv2.OPTFLOW_USE_INITIAL_FLOW, cv2.OPTFLOW_FARNEBACK_GAUSSIAN,
                          cv2.OPTFLOW_USE_INITIAL_FLOW | cv2.OPTFLOW_FARNEBACK_GAUSSIAN):
                init = first.copy() if flags & cv2.OPTFLOW_USE_INITIAL_FLOW else None   # cv.py:478: the previous output
                runs.append((int(flags), cv2.calcOpticalFlowFarneback(a, b, init, *args, flags)))
        for flags, flow in runs:
            key = f"flow_{i}_{flags}"
            arrays[key] = np.asarray(flow, np.float32)
            cases.append(dict(key=key, h=h, w=w, seed=SEED, params=prm, flags=flags, crc_prev=crc(a), crc_next=crc(b),
                              initial_flow=("flow_%d_0" % i) if flags & 4 else None))
    rng = np.random.default_rng(GREY_SEED)
    bgr = rng.integers(0, 256, GREY_SHAPE + (3,), dtype=np.uint8)
    greys = []
    for (gw, gh) in GREY_SIZES:
        frame = cv2.resize(bgr, dsize=(gw, gh), interpolation=cv2.INTER_NEAREST)        # cv.py:461-464
        key = f"grey_{gw}x{gh}"
        arrays[key] = cv2.cvtColor(frame, cv2.COLOR_BGR2GRAY)                           # cv.py:466
        greys.append(dict(key=key, width=gw, height=gh))
    meta = dict(format=FORMAT, cv2_version=cv2.__version__, numpy_version=np.__version__, build=build_lines(cv2),
                threads=int(cv2.getNumThreads()), cases=cases,
                grey=dict(seed=GREY_SEED, shape=list(GREY_SHAPE), crc_bgr=crc(bgr), outputs=greys))
    arrays["meta_json"] = np.array(json.dumps(meta))
    path = os.path.join(out_dir, f"farneback_cv2_{cv2.__version__}.npz")
    np.savez_compressed(path, **arrays)
    print(f"wrote {path}: {len(cases)} flows, {len(greys)} grey frames, cv2 {cv2.__version__}")
    return path


if __name__ == "__main__":
    main()
