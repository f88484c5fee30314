#!/usr/bin/env python3
"""A longer run of tests/test_gpu_farneback.py::test_fused_iteration_on_random_shapes_and_batches: N random frame
shapes, window widths, level counts, iteration counts, polynomial radii and batch sizes, every pair six ways --
  * the one-kernel iteration (option fb_fused = 1) with its columns cut into 1 / 3 row segments, the segments' column sums
    handed down inside the launch (fb_chain = 1) or computed by a pre-pass (fb_chain = 0),
  * the two-kernel iteration (fb_fused = 0) whole and in 3 segments (pre-pass),
  * the exact mode (fb_exact_sums = 1: the window summed in OpenCV's own order, along the rows too) three ways: the
    two-kernel iteration with its column / row walkers, and the one-kernel iteration with the rows' sums handed from strip
    to strip, whole columns and 3 segments,
and the oracle.  Since round 4 every form keeps FarnebackUpdateFlow_Blur's column sums (one running sum per column from row
0, float-differenced), so what is left between a default form and the oracle is the association of double additions
(~1e-16 in a sum).  What must hold:
  * exact mode: the oracle's flow BIT FOR BIT, every pair;
  * every other form: NO pixel beyond 1e-4 * max(1, max|ref|) of the oracle, in any pair (round 3 allowed 0.5 %);
  * the forms among themselves: no pixel further apart than that either.
Reported besides: in how many pairs each form is bit-identical to the oracle, and its largest deviation.
usage (GPU box, repo root): python3 tools/fuzz_fused.py [N] [seed] [only-case]
tests/test_gpu_farneback.py::test_plan_forms_fuzz runs fuzz(150, seed) inside the GPU suite."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
from tests.helpers import synth_pair  # noqa: E402  (the tests' frame generator: SURVEY.md section 8(d))
from oracle import farneback as O  # noqa: E402
from transflow_amd import _lib  # noqa: E402
from transflow_amd.farneback import Farneback  # noqa: E402


FORMS = [("one-kernel whole", dict(fb_fused=1, fb_segs=1, fb_chain=-1)),
         ("one-kernel 3 segments handed down", dict(fb_fused=1, fb_segs=3, fb_chain=1)),
         ("one-kernel 3 segments pre-pass", dict(fb_fused=1, fb_segs=3, fb_chain=0)),
         ("two-kernel whole", dict(fb_fused=0, fb_segs=1, fb_chain=-1)),
         ("two-kernel 3 segments", dict(fb_fused=0, fb_segs=3, fb_chain=-1)),
         ("default choice", dict(fb_fused=-1, fb_segs=0, fb_chain=-1))]
EXACT_FORMS = [("M in memory + column / row walkers", dict(fb_fused=0)),
               ("column sums straight from the expansions + row walker", dict(fb_fused=1))]
DEFAULTS = dict(fb_fused=-1, fb_segs=0, fb_chain=-1, fb_exact_sums=0)


def run(opts, w, h, n, frames, kw):
    for k, v in {**DEFAULTS, **opts}.items():
        _lib.set_option(k, v)
    fb = Farneback(w, h, max_pairs=n, frame_slots=n + 1, **kw)
    for i, f in enumerate(frames):
        fb.set_frame(i, f)
    fb.calc_slots(list(range(1, n + 1)), list(range(n)))
    out = [fb.get_flow(i) for i in range(n)]
    fb.close()
    return out


def fuzz(n_cases=100, seed=1, only=None):
    rng = np.random.default_rng(seed)
    bad = 0
    pairs = exact_identical = 0
    st = {name: dict(identical=0, worst=0.0, outlier_pairs=0, px_differ=0) for name, _ in FORMS}
    worst_between = 0.0
    for case in range(n_cases):
        if case and case % 250 == 0:
            print(f"... {case} cases, {pairs} pairs, {bad} failures so far", file=sys.stderr, flush=True)   # (a long run stays audible)
        h, w = int(rng.integers(10, 300)), int(rng.integers(10, 460))
        kw = dict(levels=int(rng.integers(0, 4)), winsize=int(rng.choice([7, 11, 15])), iterations=int(rng.integers(1, 4)),
                  poly_n=int(rng.choice([5, 7])))
        n = int(rng.integers(1, 5))
        if only is not None and case != only:
            continue
        frames = [synth_pair(h, w, seed=h * 1000 + w, shift=(0.7 * i, -0.4 * i))[1] for i in range(n + 1)]
        got = {name: run(opts, w, h, n, frames, kw) for name, opts in FORMS}
        exact = {name: run(dict(opts, fb_exact_sums=1), w, h, n, frames, kw) for name, opts in EXACT_FORMS}
        for i in range(n):
            ref = O.calc(frames[i + 1], frames[i], **kw)
            scale = max(1.0, float(np.abs(ref).max()))
            pairs += 1
            same = True
            for name, _ in EXACT_FORMS:
                if not np.array_equal(exact[name][i], ref):
                    same = False
                    bad += 1
                    de = np.abs(exact[name][i] - ref).max(axis=2)
                    print(f"FAIL case {case}: {h}x{w} {kw} pair {i} of {n}: exact mode ({name}) differs from the oracle in "
                          f"{int((de > 0).sum())} pixels, max|d| {de.max():.3g}")
            exact_identical += same
            for name, _ in FORMS:
                d = np.abs(got[name][i] - ref).max(axis=2)
                n_out = int((d > 1e-4 * scale).sum())
                st[name]["identical"] += bool(np.array_equal(got[name][i], ref))
                st[name]["px_differ"] += int((d > 0).sum())
                st[name]["worst"] = max(st[name]["worst"], float(d.max()) / (1e-4 * scale))
                st[name]["outlier_pairs"] += n_out > 0
                if n_out or only is not None:
                    ys, xs = np.nonzero(d > 1e-4 * scale)
                    where = f"rows {ys.min()}..{ys.max()}, cols {xs.min()}..{xs.max()}" if len(ys) else "none"
                    print(f"{'FAIL ' if n_out else ''}case {case} {h}x{w} {kw} pair {i} of {n}: {name}: {n_out} pixels beyond tolerance ({where}), "
                          f"{int((d > 0).sum())} differ at all, max|d| {d.max():.3g}, max|ref| {scale:.3g}")
                    bad += n_out > 0
            for name, _ in FORMS[1:]:
                db = float(np.abs(got[name][i] - got[FORMS[0][0]][i]).max()) / (1e-4 * scale)
                worst_between = max(worst_between, db)
                if db > 1.0:
                    bad += 1
                    print(f"FAIL case {case} {h}x{w} {kw} pair {i}: {name} is {db:.2f} tolerances from the one-kernel form")
    for k, v in DEFAULTS.items():
        _lib.set_option(k, v)
    print(f"{n_cases} cases, {pairs} pairs: exact mode ({len(EXACT_FORMS)} forms) bit-identical to the oracle in {exact_identical}")
    for name, _ in FORMS:
        t = st[name]
        print(f"  {name:52s}: bit-identical to the oracle in {t['identical']} pairs ({t['px_differ']} pixels differ in all), "
              f"largest deviation {t['worst']:.2e} of the tolerance, {t['outlier_pairs']} pairs with a pixel beyond it")
    print(f"  the forms among themselves: at most {worst_between:.2e} of the tolerance apart; {bad} failures")
    return dict(cases=n_cases, pairs=pairs, failures=bad, exact_identical=exact_identical, forms=st, worst_between=worst_between)


if __name__ == "__main__":
    res = fuzz(int(sys.argv[1]) if len(sys.argv) > 1 else 100, int(sys.argv[2]) if len(sys.argv) > 2 else 1,
               int(sys.argv[3]) if len(sys.argv) > 3 else None)
    sys.exit(1 if res["failures"] else 0)
