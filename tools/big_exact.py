#!/usr/bin/env python3
"""The exact mode (option fb_exact_sums) and the default mode against the oracle at sizes the suite does not reach:
7680x4320, the odd 7683x4321 and a 4K call with a 21-pixel window and poly_n = 7.
usage (GPU box, repo root): python3 tools/big_exact.py"""
import sys, os, time
sys.path.insert(0, os.getcwd()); sys.path.insert(0, os.path.join(os.getcwd(), "tests"))
import numpy as np
from helpers import synth_pair
from oracle import farneback as O
from transflow_amd import _lib
from transflow_amd.farneback import Farneback
for (h, w, kw) in [(4320, 7680, dict(levels=5)), (4321, 7683, dict(levels=5)), (2160, 3840, dict(levels=3, winsize=21, poly_n=7, poly_sigma=1.5))]:
    a, b = synth_pair(h, w, seed=41)
    t0 = time.time(); ref = O.calc(a, b, **kw); t_o = time.time() - t0
    out = {}
    # default; exact (M in memory, column and row walkers); exact with the column sums straight from the expansions
    for name, opts in (("default", dict(fb_exact_sums=0)), ("exact", dict(fb_exact_sums=1)),
                       ("exact from R", dict(fb_exact_sums=1, fb_fused=1))):
        for k, v in {**dict(fb_exact_sums=0, fb_fused=-1, fb_segs=0, fb_chain=-1), **opts}.items():
            _lib.set_option(k, v)
        fb = Farneback(w, h, **kw)
        got = fb.calc(a, b)
        fb.close()
        d = np.abs(got - ref).max(axis=2)
        tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
        out[name] = (int((d > tol).sum()), float(d.max()), bool(np.array_equal(got, ref)), int((d > 0).sum()))
    print(f"{w}x{h} {kw}: oracle {t_o:.1f} s; default mode: {out['default'][0]} pixels beyond {tol:.3g}, {out['default'][3]} differ at all "
          f"(max|d| {out['default'][1]:.3g}); exact mode: bit-identical={out['exact'][2]}; exact, column sums straight from the expansions: bit-identical={out['exact from R'][2]}", flush=True)
