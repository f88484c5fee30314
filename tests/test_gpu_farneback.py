"""GPU parity, Farnebäck half: libtfhip.so (through the C ABI) against the C oracle
(oracle/farneback_ref.c).  Stages whose arithmetic order is fixed (pre-blur +
resize, polynomial expansion, update-matrices) must be bit-identical; the box-blur
solve sums its window in a different order (double), so it and the whole call are
held to the tolerance north_star states: max|d| <= 1e-4 * max(1, max|ref|)."""
import numpy as np
import pytest

from oracle import farneback as O
from tests.helpers import FB_CASES as CASES, FB_SWEEP as SWEEP, synth_pair

pytestmark = pytest.mark.gpu


TOL = 1e-4


def flow_tol(ref):
    return TOL * max(1.0, float(np.abs(ref).max()))


@pytest.fixture(scope="module")
def FB():
    from transflow_amd.farneback import Farneback
    return Farneback


@pytest.mark.parametrize("shape,levels", [((270, 480), 3), ((135, 241), 2), ((480, 854), 3), ((1080, 1920), 5),
                                          # frames that are exactly 4 x their scale-1/4 level (k_level_quarter_image): one tile
                                          # that is all border, partial tiles, interior tiles between border ones
                                          ((128, 256), 2), ((132, 260), 2), ((520, 1028), 2)])
def test_level_images_bit_exact(FB, shape, levels):
    h, w = shape
    a, _ = synth_pair(h, w, seed=11)
    fb = FB(w, h, levels=levels)
    sizes = fb.level_sizes()
    assert len(sizes) == O.num_levels(w, h, 0.5, levels) + 1
    for k, (wk, hk) in enumerate(sizes):
        assert (wk, hk) == O.level_geometry(w, h, 0.5, k)[:2]
        np.testing.assert_array_equal(fb.stage_level_image(a, k), O.level_image(a, 0.5, k), err_msg=f"level {k}")
    fb.close()


@pytest.mark.parametrize("shape", [(270, 480), (135, 241), (33, 70), (1080, 1920)])
@pytest.mark.parametrize("poly", [(5, 1.2), (7, 1.5), (3, 1.0)])
def test_level_then_polyexp_bit_exact(FB, shape, poly):
    """A1 followed by A2 through the kernels the full path uses (fused at level 0 for poly_n 5/7)."""
    h, w = shape
    a, _ = synth_pair(h, w, seed=13)
    fb = FB(w, h, levels=2, poly_n=poly[0], poly_sigma=poly[1])
    for k in range(len(fb.level_sizes())):
        exp = O.polyexp(O.level_image(a, 0.5, k), *poly)
        np.testing.assert_array_equal(fb.stage_level_polyexp(a, k), exp, err_msg=f"level {k}")
    fb.close()


@pytest.mark.parametrize("shape", [(64, 64), (67, 131), (16, 200), (270, 480), (5, 7)])
@pytest.mark.parametrize("poly", [(5, 1.2), (7, 1.5), (3, 0.0)])
def test_polyexp_bit_exact(FB, shape, poly):
    h, w = shape
    rng = np.random.default_rng(3)
    img = rng.uniform(0, 255, (h, w)).astype(np.float32)
    fb = FB(max(w, 32), max(h, 32), levels=0, poly_n=poly[0], poly_sigma=poly[1])
    np.testing.assert_array_equal(fb.stage_polyexp(img), O.polyexp(img, *poly))
    fb.close()


@pytest.mark.parametrize("shape", [(64, 64), (67, 131), (270, 480), (11, 13), (9, 40), (40, 9), (7, 8), (3, 12), (1, 1)])
def test_update_matrices_bit_exact(FB, shape):
    h, w = shape
    rng = np.random.default_rng(4)
    r0 = rng.normal(0, 3, (h, w, 5)).astype(np.float32)
    r1 = rng.normal(0, 3, (h, w, 5)).astype(np.float32)
    flow = rng.normal(0, 4, (h, w, 2)).astype(np.float32)
    flow[0, 0] = (-3.5, 2.25)          # leaves the frame: the "else" branch
    flow[h // 2, w // 2] = (1e4, -1e4)
    fb = FB(max(w, 32), max(h, 32), levels=0)
    np.testing.assert_array_equal(fb.stage_update_matrices(r0, r1, flow), O.update_matrices(r0, r1, flow))
    fb.close()


@pytest.mark.parametrize("shape,pyr_scale,levels", [((270, 480), 0.5, 2), ((271, 483), 0.5, 2), ((480, 854), 0.7, 3),
                                                    ((333, 517), 0.8, 4), ((200, 300), 0.37, 1), ((135, 241), 0.9, 5)])
def test_flow_upsample_then_matrices_bit_exact(FB, shape, pyr_scale, levels):
    """Stage test of A5 (resize(prevFlow -> level size, INTER_LINEAR), flow *= 1/pyr_scale), which
    only exists fused into the matrix kernel: at every level below the coarsest, including non-dyadic
    pyramids and odd sizes, M from (R0, R1, upsampled coarse flow) equals the oracle's resize + scale +
    FarnebackUpdateMatrices bit for bit -- so the upsampled flow itself is bit-identical."""
    h, w = shape
    fb = FB(w, h, pyr_scale=pyr_scale, levels=levels)
    sizes = fb.level_sizes()
    assert len(sizes) >= 2
    rng = np.random.default_rng(55)
    for k in range(len(sizes) - 1):
        (wk, hk), (wc, hc) = sizes[k], sizes[k + 1]
        r0 = rng.normal(0, 3, (hk, wk, 5)).astype(np.float32)
        r1 = rng.normal(0, 3, (hk, wk, 5)).astype(np.float32)
        coarse = rng.normal(0, 2, (hc, wc, 2)).astype(np.float32)
        flow = O.resize_linear(coarse, wk, hk) * np.float32(1.0 / pyr_scale)
        np.testing.assert_array_equal(fb.stage_upsampled_matrices(k, r0, r1, coarse), O.update_matrices(r0, r1, flow),
                                      err_msg=f"level {k}: {wc}x{hc} -> {wk}x{hk}")
    fb.close()


@pytest.mark.parametrize("shape,winsize", [((64, 64), 15), ((67, 531), 15), ((270, 480), 15), ((40, 50), 9),
                                           ((33, 300), 4), ((9, 11), 15)])
def test_blur_solve_close(FB, shape, winsize):
    h, w = shape
    rng = np.random.default_rng(5)
    r = rng.normal(0, 3, (h, w, 5)).astype(np.float32)
    # a plausible positive-definite system: M from update_matrices of random R's
    m = O.update_matrices(r, rng.normal(0, 3, (h, w, 5)).astype(np.float32), np.zeros((h, w, 2), np.float32))
    ref, _ = O.update_flow_blur(r, r, np.zeros((h, w, 2), np.float32), m, winsize, False)
    fb = FB(max(w, 32), max(h, 32), levels=0, winsize=winsize)
    got = fb.stage_blur_solve(m)
    fb.close()
    assert np.abs(got - ref).max() <= flow_tol(ref)


@pytest.mark.parametrize("shape,kw", SWEEP)
def test_parameter_sweep_close(FB, shape, kw):
    """More of cv.py:273-281's parameter space against the oracle, default kernel choice per level."""
    h, w = shape
    a, b = synth_pair(h, w, seed=31, shift=(2.0, -1.5))
    ref = O.calc(a, b, **kw)
    fb = FB(w, h, **kw)
    got = fb.calc(a, b)
    err = np.abs(got - ref).max()
    assert err <= flow_tol(ref), f"max|d|={err} tol={flow_tol(ref)}"
    fb.close()


def test_small_and_degenerate_shapes_close(FB):
    """Every shape down to one pixel, one row and one column, odd and prime sizes, with random pyramid
    depth and window: the path never reads outside a frame and stays within tolerance of the oracle."""
    rng = np.random.default_rng(2024)
    shapes = [(1, 1), (1, 2), (2, 1), (2, 2), (3, 3), (1, 64), (64, 1), (1, 300), (300, 1), (3, 130), (130, 3),
              (31, 33), (32, 32), (33, 31), (63, 65), (64, 64), (65, 63)]
    shapes += [(int(rng.integers(1, 90)), int(rng.integers(1, 150))) for _ in range(14)]
    for h, w in shapes:
        kw = dict(levels=int(rng.integers(0, 4)), winsize=int(rng.choice([5, 7, 9, 11, 15, 21])),
                  iterations=int(rng.integers(1, 4)), poly_n=int(rng.choice([5, 7])))
        kw["poly_sigma"] = 1.2 if kw["poly_n"] == 5 else 1.5
        a, b = synth_pair(h, w, seed=h * 1000 + w, shift=(1.2, -0.7), noise=2.0)
        ref = O.calc(a, b, **kw)
        fb = FB(w, h, **kw)
        got = fb.calc(a, b)
        fb.close()
        assert got.shape == (h, w, 2) and np.isfinite(got).all(), f"{h}x{w} {kw}"
        err = np.abs(got - ref).max()
        assert err <= flow_tol(ref), f"{h}x{w} {kw}: max|d|={err} tol={flow_tol(ref)}"


def test_random_pyramids_close(FB):
    """Random mid-size frames, pyramid scales and depths: every level is resized with different fractions."""
    rng = np.random.default_rng(77)
    for _ in range(14):
        h, w = int(rng.integers(70, 420)), int(rng.integers(70, 520))
        kw = dict(levels=int(rng.integers(1, 6)), pyr_scale=float(rng.choice([0.5, 0.6, 0.75, 0.8, 0.9])),
                  winsize=int(rng.choice([7, 11, 15, 19])), iterations=int(rng.integers(1, 4)))
        a, b = synth_pair(h, w, seed=h * 1000 + w, shift=(1.7, 0.9))
        ref = O.calc(a, b, **kw)
        fb = FB(w, h, **kw)
        got = fb.calc(a, b)
        fb.close()
        err = np.abs(got - ref).max()
        assert err <= flow_tol(ref), f"{h}x{w} {kw}: max|d|={err} tol={flow_tol(ref)}"


@pytest.mark.parametrize("shape,kw", CASES)
def test_full_calc_close(FB, shape, kw):
    h, w = shape
    a, b = synth_pair(h, w, seed=21)
    ref = O.calc(a, b, **kw)
    fb = FB(w, h, **kw)
    got = fb.calc(a, b)
    assert got.dtype == np.float32 and got.shape == (h, w, 2)
    err = np.abs(got - ref).max()
    assert err <= flow_tol(ref), f"max|d|={err} tol={flow_tol(ref)}"
    # the call is a pure function of its inputs: replay is bit-identical
    np.testing.assert_array_equal(fb.calc(a, b), got)
    fb.close()


def test_1080p_default_and_levels5(FB):
    h, w = 1080, 1920
    a, b = synth_pair(h, w, seed=31)
    for kw in (dict(levels=3), dict(levels=5)):
        ref = O.calc(a, b, **kw)
        fb = FB(w, h, **kw)
        got = fb.calc(a, b)
        fb.close()
        err = np.abs(got - ref).max()
        assert err <= flow_tol(ref), f"{kw}: max|d|={err} tol={flow_tol(ref)}"


def test_4k_properties(FB):
    """3840x2160 (BASELINE configs[3]): the oracle takes ~20 s here, so check it once at
    levels=5, plus size-independent properties: identical frames give zero flow away from
    the bottom/right edge; swapping the frames negates the field."""
    h, w = 2160, 3840
    a, b = synth_pair(h, w, seed=41)
    fb = FB(w, h, levels=5)
    got = fb.calc(a, b)
    ref = O.calc(a, b, levels=5)
    err = np.abs(got - ref).max()
    assert err <= flow_tol(ref), f"max|d|={err}"
    back = fb.calc(b, a)
    inner = (slice(200, -200), slice(200, -200))
    assert np.abs(got[inner] + back[inner]).mean() < 0.2 * np.abs(got[inner]).mean() + 0.1
    fb.close()
    # the edge residual of identical frames spreads 3*7 px per scale: use two coarse scales
    fb = FB(w, h, levels=2)
    same = fb.calc(a, a)
    fb.close()
    assert np.abs(same[:1000, :2000]).max() == 0.0
    assert np.abs(same).max() < 0.5


def test_batch_equals_single(FB):
    h, w = 270, 480
    frames = [synth_pair(h, w, seed=50 + i)[i % 2] for i in range(5)]
    single = FB(w, h)
    exp = [single.calc(frames[i], frames[i + 1]) for i in range(4)]
    single.close()
    fb = FB(w, h, frame_slots=5, max_pairs=4)
    for i, f in enumerate(frames):
        fb.set_frame(i, f)
    fb.calc_slots([0, 1, 2, 3], [1, 2, 3, 4])
    for i in range(4):
        np.testing.assert_array_equal(fb.get_flow(i), exp[i])
    # BACKWARD ordering (cv.py:470-472): (prev, next) = (current, previous)
    fb.calc_slots([1], [0])
    single = FB(w, h)
    np.testing.assert_array_equal(fb.get_flow(0), single.calc(frames[1], frames[0]))
    single.close()
    fb.close()


def test_shared_frames_are_expanded_once_with_identical_results(FB, lib_option):
    """Pairs of a batch that name the same frame slot share its expansion (A1+A2 depend on the frame
    alone): bit-identical to expanding per pair and side (option "fb_no_share" = 1), whatever the order, with
    repeated pairs and a pair of a frame with itself."""
    h, w = 270, 480
    frames = [synth_pair(h, w, seed=90, shift=(0.8 * i, 0.5 * i))[1] for i in range(5)]   # one texture, five displacements
    prev = [0, 1, 2, 3, 3, 1, 4, 2]
    nxt = [1, 2, 3, 4, 3, 0, 0, 3]
    out = {}
    for mode in ("0", "1"):
        lib_option("fb_no_share", int(mode))
        fb = FB(w, h, levels=3, max_pairs=len(prev), frame_slots=len(frames))
        for i, f in enumerate(frames):
            fb.set_frame(i, f)
        fb.calc_slots(prev, nxt)
        out[mode] = [fb.get_flow(i) for i in range(len(prev))]
        fb.calc_slots(nxt, prev)                      # a second call on the same handle: the map is per call
        back = [fb.get_flow(i) for i in range(len(prev))]
        fb.close()
        for i in range(len(prev)):
            if prev[i] == nxt[i]:
                continue    # a frame against itself is ill-conditioned on the first row and column (see below)
            ref = O.calc(frames[prev[i]], frames[nxt[i]], levels=3)
            assert np.abs(out[mode][i] - ref).max() <= flow_tol(ref)
        ref = O.calc(frames[nxt[0]], frames[prev[0]], levels=3)
        assert np.abs(back[0] - ref).max() <= flow_tol(ref)
    for a, b in zip(out["0"], out["1"]):
        np.testing.assert_array_equal(a, b)
    np.testing.assert_array_equal(out["0"][2], out["0"][7])       # the same pair twice in one batch
    # A frame against itself: zero flow away from the borders.  On the first row / column the sign of a
    # 1e-8 flow decides whether x + dx floors to -1 (R1 "absent", cv2's bounds test) or to 0, so any two
    # implementations that differ in the last bit of the window sums part ways there: checked loosely.
    assert np.abs(out["0"][4][40:-40, 40:-40]).max() < 1e-3
    assert np.abs(out["0"][4]).max() < 0.5


def test_kept_expansions_stream_like_fresh_calls(FB):
    """tf_fb_keep_expansions: a stream of frames through two alternating slots, one new frame per call
    (what HipFlowSource does), equals a fresh handle per pair bit for bit -- also after a slot is
    rewritten out of turn, when nothing at all needs expanding, and for a slot written on the device."""
    import ctypes as C
    from transflow_amd import _lib
    h, w = 270, 480
    frames = [synth_pair(h, w, seed=95, shift=(0.9 * i, 0.6 * i))[1] for i in range(6)]

    def fresh(a, b):
        fb = FB(w, h, levels=3)
        out = fb.calc(a, b)
        fb.close()
        return out

    fb = FB(w, h, levels=3, frame_slots=2, max_pairs=1)
    fb.keep_expansions(True)
    fb.set_frame(0, frames[0])
    prev = 0
    for t in range(1, 5):
        new = prev ^ 1
        fb.set_frame(new, frames[t])
        fb.calc_slots([new], [prev])                                   # BACKWARD ordering (cv.py:470-472)
        np.testing.assert_array_equal(fb.get_flow(0), fresh(frames[t], frames[t - 1]), err_msg=f"frame {t}")
        prev = new
    fb.calc_slots([prev], [prev ^ 1])                                  # nothing was written: no expansion at all
    np.testing.assert_array_equal(fb.get_flow(0), fresh(frames[4], frames[3]))
    fb.set_frame(prev, frames[5])                                      # the slot that would have been kept
    fb.calc_slots([prev ^ 1], [prev])
    np.testing.assert_array_equal(fb.get_flow(0), fresh(frames[3], frames[5]))
    # a slot the caller writes on the device is expanded every time
    lib = _lib.load()
    ptr = fb.frame_ptr(0)
    for f in (frames[1], frames[2]):
        _lib.check(lib.tf_dev_upload(C.c_void_p(ptr), C.c_void_p(f.ctypes.data), f.nbytes))
        fb.calc_slots([0], [1])                                        # tf_dev_upload returns with the bytes in place
        other = frames[5] if prev == 1 else frames[3]
        np.testing.assert_array_equal(fb.get_flow(0), fresh(f, other))
    fb.close()
    big = FB(w, h, levels=3, frame_slots=5, max_pairs=2)
    with pytest.raises(ValueError):
        big.keep_expansions(True)                                       # 5 slots, room for 4 expansions
    big.close()


def test_pipelined_calls_equal_single_calls(FB):
    """tf_fb_calc_slots only enqueues: six calls issued back to back without a host synchronisation
    (the next call's frame expansion runs on the second stream beside the previous call's
    iterations, R double-buffered) give bit for bit the flows of the same pairs run one at a time."""
    import ctypes as C
    from transflow_amd import _lib
    from transflow_amd._lib import check
    from transflow_amd.device import DevBuffer
    h, w = 360, 640
    frames = [synth_pair(h, w, seed=70 + i)[i % 2] for i in range(8)]
    single = FB(w, h, levels=4)
    calls = [([0, 1], [1, 2]), ([2, 3], [3, 4]), ([5, 4], [4, 3]), ([6, 7], [7, 6]), ([0, 7], [7, 0]), ([3, 1], [2, 5])]
    exp = [[single.calc(frames[a], frames[b]) for a, b in zip(*c)] for c in calls]
    single.close()
    fb = FB(w, h, levels=4, frame_slots=8, max_pairs=2)
    for i, f in enumerate(frames):
        fb.set_frame(i, f)
    lib = _lib.load()
    keep = [[DevBuffer(h * w * 8) for _ in range(2)] for _ in calls]
    for c, bufs in zip(calls, keep):
        fb.calc_slots(*c)                       # no synchronisation between the calls
        for i, buf in enumerate(bufs):          # the flow buffers are reused by the next call: copy, in stream order
            check(lib.tf_dev_copy(C.c_void_p(buf.ptr), C.c_void_p(fb.flow_ptr(i)), h * w * 8))
    for e, bufs in zip(exp, keep):
        for i, buf in enumerate(bufs):
            np.testing.assert_array_equal(buf.download((h, w, 2), np.float32), e[i])
            buf.close()
    fb.close()


def test_fused_and_two_kernel_iterations_agree(FB, lib_option):
    """The same pyramid with the iteration as one kernel on every level (option "fb_fused" = 1), as two kernels on
    every level (=0) and with the default per-level choice: identical algorithm, different summation
    order in the window sums and a different reciprocal -- far inside the path's 1e-4 tolerance, and
    each within tolerance of the oracle."""
    h, w = 540, 960
    a, b = synth_pair(h, w, seed=80, shift=(2.5, 1.5))
    ref = O.calc(a, b, levels=3)
    out = {}
    for mode in ("1", "0", None):
        lib_option("fb_fused", -1 if mode is None else int(mode))      # read when the handle is created
        fb = FB(w, h, levels=3, max_pairs=12, frame_slots=2)   # 12 pairs: level 0 crosses the 4M-pixel threshold
        fb.set_frame(0, a)
        fb.set_frame(1, b)
        fb.calc_slots([0] * 12, [1] * 12)
        out[mode] = fb.get_flow(5)
        np.testing.assert_array_equal(out[mode], fb.get_flow(11))      # every pair of the batch alike
        assert np.abs(out[mode] - ref).max() <= flow_tol(ref)
        fb.close()
    scale = max(1.0, float(np.abs(ref).max()))
    assert np.abs(out["1"] - out["0"]).max() <= 2e-5 * scale
    assert np.abs(out[None] - out["0"]).max() <= 2e-5 * scale


@pytest.mark.parametrize("shape,kw", CASES + [
    ((135, 241), dict(levels=2, winsize=7)),                # window half-widths 3 and 5 of the fused kernel
    ((97, 113), dict(levels=1, winsize=11)),
    ((33, 2), dict(levels=0)),                              # narrower than one lane pair, shorter than the window
    ((2, 300), dict(levels=0)),
])
def test_full_calc_close_fused_on_every_level(FB, lib_option, shape, kw):
    """The one-kernel iteration forced on every level it supports (option "fb_fused" = 1; by default only levels of
    >= 4M pixels use it): ragged widths and heights, strips narrower than a workgroup, every window width."""
    lib_option("fb_fused", 1)
    h, w = shape
    a, b = synth_pair(h, w, seed=23)
    ref = O.calc(a, b, **kw)
    fb = FB(w, h, max_pairs=2, frame_slots=2, **kw)
    fb.set_frame(0, a)
    fb.set_frame(1, b)
    fb.calc_slots([0, 1], [1, 0])
    got = fb.get_flow(0)
    err = np.abs(got - ref).max()
    assert err <= flow_tol(ref), f"max|d|={err} tol={flow_tol(ref)}"
    back = O.calc(b, a, **kw)
    assert np.abs(fb.get_flow(1) - back).max() <= flow_tol(back)
    fb.close()


def test_fused_iteration_on_random_shapes_and_batches(FB, lib_option):
    """The one-kernel iteration forced everywhere (option "fb_fused" = 1) over random frame shapes, window widths
    and batch sizes: strips narrower and wider than a workgroup, segments shorter than the window, the last
    strip and the last segment ragged; every pair of the batch against the oracle."""
    lib_option("fb_fused", 1)
    rng = np.random.default_rng(606)
    for _ in range(16):
        h, w = int(rng.integers(10, 260)), int(rng.integers(10, 420))
        kw = dict(levels=int(rng.integers(0, 4)), winsize=int(rng.choice([7, 11, 15])), iterations=int(rng.integers(1, 4)))
        n = int(rng.integers(1, 6))
        frames = [synth_pair(h, w, seed=h * 1000 + w, shift=(0.7 * i, -0.4 * i))[1] for i in range(n + 1)]
        fb = FB(w, h, max_pairs=n, frame_slots=n + 1, **kw)
        for i, f in enumerate(frames):
            fb.set_frame(i, f)
        fb.calc_slots(list(range(1, n + 1)), list(range(n)))
        for i in range(n):
            ref = O.calc(frames[i + 1], frames[i], **kw)
            err = np.abs(fb.get_flow(i) - ref).max()
            assert err <= flow_tol(ref), f"{h}x{w} {kw} pair {i} of {n}: max|d|={err} tol={flow_tol(ref)}"
        fb.close()


def test_plan_forms_fuzz(lib_option, capsys):
    """tools/fuzz_fused.py inside the suite (rounds 3-4 kept its runs under profiles/ only): 150 random configurations,
    fixed seed, every pair through the six forms of the marching kernels' plan -- one-kernel iteration whole / 3 segments
    handed down / 3 segments from the pre-pass, two-kernel iteration whole / 3 segments, the planner's own choice -- and
    the exact mode two ways, against the oracle.  Exact mode: bit-identical in every pair.  Every other form: no pixel
    beyond 1e-4 * max(1, max|ref|), and no two forms further apart than that."""
    for name in ("fb_fused", "fb_segs", "fb_chain", "fb_exact_sums"):
        lib_option(name, _lib_get(name))             # the tool sets them per form: restored whatever happens
    from tools.fuzz_fused import EXACT_FORMS, FORMS, fuzz
    res = fuzz(150, 20261004)
    report = capsys.readouterr().out
    assert res["failures"] == 0, report
    assert res["pairs"] >= 150 and res["exact_identical"] == res["pairs"], report
    assert all(res["forms"][name]["outlier_pairs"] == 0 for name, _ in FORMS), report
    assert res["worst_between"] <= 1.0 and len(EXACT_FORMS) == 2


def _lib_get(name):
    from transflow_amd import _lib
    return _lib.get_option(name)


def test_unusual_parameter_values_close(FB):
    """Values cv2 accepts but nobody picks: even window sizes (the window is 2*(winsize/2)+1 wide, the
    scale 1/winsize^2: optflowgf.cpp as written), windows of 1-3 pixels, every polynomial radius from 1 to
    9 (the generic expansion kernel below 5 and between), poly_sigma <= 0 (0.3*n instead), one scale
    after the other of a 0.95 pyramid."""
    h, w = 120, 164
    a, b = synth_pair(h, w, seed=41, shift=(1.4, 0.8))
    rng = np.random.default_rng(12)
    cases = [dict(winsize=ws) for ws in (1, 2, 3, 4, 6, 8, 10, 12, 16, 20, 24)]
    cases += [dict(poly_n=n, poly_sigma=float(s)) for n, s in ((1, 0.5), (2, 0.8), (3, 1.0), (4, 1.1), (6, 1.4), (8, 1.7), (9, 0.0))]
    cases += [dict(levels=4, pyr_scale=0.95), dict(levels=30, pyr_scale=0.5), dict(iterations=7, winsize=9)]
    cases += [dict(winsize=int(rng.integers(1, 30)), poly_n=int(rng.integers(1, 10)), poly_sigma=float(rng.uniform(0.4, 2)),
                   iterations=int(rng.integers(1, 4)), levels=int(rng.integers(0, 3))) for _ in range(8)]
    for kw in cases:
        ref = O.calc(a, b, **kw)
        fb = FB(w, h, **kw)
        got = fb.calc(a, b)
        fb.close()
        d = np.abs(got - ref)
        assert np.isfinite(got).all(), kw
        # (Windows of 1-3 pixels -- 1-9 samples per 2x2 system, G close to singular wherever the image is locally
        # one-dimensional -- amplified the last-bit difference between OpenCV's running sums and the kernels' direct sums
        # through iterations and scales; rounds 1-3 allowed 1 % of their pixels beyond tolerance.  With OpenCV's column
        # sums in the kernels they meet the bar like every other window.)
        assert d.max() <= flow_tol(ref), f"{kw}: max|d|={d.max()} tol={flow_tol(ref)}"


def test_frame_content_extremes_close(FB):
    """Flat frames, saturated frames, white noise, a one-pixel checkerboard, a displacement larger than
    the window (most gathers leave the frame at the fine scales), a pure ramp: one iteration at one scale
    and the default pyramid within tolerance everywhere (rounds 1-3 allowed the structureless ones -- noise, the
    checkerboard -- 1 % of their pixels beyond it)."""
    h, w = 150, 200
    rng = np.random.default_rng(8)
    yy, xx = np.mgrid[0:h, 0:w]
    tex, moved = synth_pair(h, w, seed=55, shift=(31.0, -17.0))
    noise = [rng.integers(0, 256, (h, w), dtype=np.uint8) for _ in range(2)]
    check = ((yy + xx) & 1).astype(np.uint8) * 255
    ramp = np.clip(xx * 255 // (w - 1), 0, 255).astype(np.uint8)
    pairs = {
        "zeros": (np.zeros((h, w), np.uint8), np.zeros((h, w), np.uint8), True),
        "saturated": (np.full((h, w), 255, np.uint8), np.full((h, w), 255, np.uint8), True),
        "flat vs texture": (np.full((h, w), 128, np.uint8), tex, True),
        "large shift": (tex, moved, True),
        "ramp shifted": (ramp, np.roll(ramp, 3, axis=1), True),
        "noise": (noise[0], noise[1], False),
        "checkerboard": (check, np.roll(check, 1, axis=1), False),
    }
    for name, (a, b, smooth) in pairs.items():
        for kw in (dict(levels=0, iterations=1), dict()):
            ref = O.calc(a, b, **kw)
            fb = FB(w, h, **kw)
            got = fb.calc(a, b)
            fb.close()
            d = np.abs(got - ref)
            assert np.isfinite(got).all(), name
            assert d.max() <= flow_tol(ref), f"{name} {kw}: max|d|={d.max()} tol={flow_tol(ref)}"


def test_strided_input_and_errors(FB):
    h, w = 64, 96
    a, b = synth_pair(h, w + 8, seed=60)
    fb = FB(w, h)
    ref = O.calc(np.ascontiguousarray(a[:, :w]), np.ascontiguousarray(b[:, :w]))
    got = fb.calc(a[:, :w], b[:, :w])        # row stride > width
    assert np.abs(got - ref).max() <= flow_tol(ref)
    with pytest.raises(ValueError):
        fb.calc(a, b)                         # wrong shape
    with pytest.raises(ValueError):
        fb.calc(a[:, :w].astype(np.float32), b[:, :w])
    fb.close()
    with pytest.raises(NotImplementedError):
        FB(w, h, flags=8)                     # no such flag: only 4 and 256 exist (cv.py:281, 489)
    with pytest.raises(NotImplementedError):
        FB(w, h, flags=256, winsize=129)      # the Gaussian window's tile serves winsize <= 63
    plain = FB(w, h)
    with pytest.raises(ValueError):
        plain.set_initial_flow(0, np.zeros((h, w, 2), np.float32))    # handle without OPTFLOW_USE_INITIAL_FLOW
    plain.close()
    with pytest.raises(ValueError):
        FB(w, h, pyr_scale=1.0)
    with pytest.raises(ValueError):
        FB(0, 0)


def test_against_cv2_when_available(FB, lib_option):
    """Wherever `import cv2` works (neither this build container nor the GPU box): OpenCV itself as the reference --
    the box window in the default mode (tolerance) and in the exact mode (bit for bit, if the build of OpenCV does not
    contract multiply-adds), and the Gaussian window (flags = 256: tolerance; this is where the float-or-double question
    of FarnebackUpdateFlow_GaussianBlur's solve would be decided)."""
    cv2 = pytest.importorskip("cv2")
    h, w = 480, 854
    a, b = synth_pair(h, w, seed=70)
    for flags in (0, 256):
        ref = cv2.calcOpticalFlowFarneback(a, b, None, 0.5, 3, 15, 3, 5, 1.2, flags)
        fb = FB(w, h, flags=flags)
        got = fb.calc(a, b)
        fb.close()
        assert np.abs(got - ref).max() <= flow_tol(ref), f"flags={flags}"
    ref = cv2.calcOpticalFlowFarneback(a, b, None, 0.5, 3, 15, 3, 5, 1.2, 0)
    lib_option("fb_exact_sums", 1)
    fb = FB(w, h)
    got = fb.calc(a, b)
    fb.close()
    assert np.abs(got - ref).max() <= 1e-6 * max(1.0, float(np.abs(ref).max())), "exact mode vs cv2"


# ---- fb_flags: OPTFLOW_USE_INITIAL_FLOW (4) and OPTFLOW_FARNEBACK_GAUSSIAN (256), cv.py:281, 489 ----------------

@pytest.mark.parametrize("shape,winsize", [((64, 64), 15), ((67, 531), 15), ((270, 480), 15), ((40, 50), 9), ((33, 300), 4),
                                           ((9, 11), 15), ((130, 70), 31), ((1, 1), 3)])
def test_gaussian_window_solve_bit_exact(FB, shape, winsize):
    """FarnebackUpdateFlow_GaussianBlur: float Gaussian of M (vertical, then horizontal, replicated borders) and
    the 2x2 solve -- the kernel runs optflowgf.cpp's scalar statements, so it equals the oracle bit for bit."""
    h, w = shape
    rng = np.random.default_rng(75)
    r = rng.normal(0, 3, (h, w, 5)).astype(np.float32)
    m = O.update_matrices(r, rng.normal(0, 3, (h, w, 5)).astype(np.float32), np.zeros((h, w, 2), np.float32))
    ref, _ = O.update_flow_gaussian(r, r, np.zeros((h, w, 2), np.float32), m, winsize, False)
    fb = FB(max(w, 32), max(h, 32), levels=0, winsize=winsize, flags=256)
    np.testing.assert_array_equal(fb.stage_blur_solve(m), ref)
    fb.close()


@pytest.mark.parametrize("shape,pyr_scale,levels", [((256, 384), 0.5, 3), ((270, 480), 0.5, 3), ((480, 854), 0.5, 3),
                                                    ((1080, 1920), 0.5, 5), ((333, 517), 0.7, 4), ((64, 96), 0.5, 0)])
def test_initial_flow_shrunk_to_the_coarsest_scale_bit_exact(FB, shape, pyr_scale, levels):
    """OPTFLOW_USE_INITIAL_FLOW's first step: resize(flow, INTER_AREA) to the coarsest scale, times pyr_scale^K --
    both of resize's paths (integer factors: 256x384 / 8; fractional cell coverage otherwise; same size: a copy)."""
    h, w = shape
    fb = FB(w, h, pyr_scale=pyr_scale, levels=levels, flags=4)
    wc, hc = fb.level_sizes()[-1]
    k = len(fb.level_sizes()) - 1
    rng = np.random.default_rng(76)
    flow = rng.normal(0, 5, (h, w, 2)).astype(np.float32)
    scale = 1.0
    for _ in range(k):
        scale *= pyr_scale
    ref = O.resize_area(flow, wc, hc) * np.float32(scale)
    np.testing.assert_array_equal(fb.stage_initial_flow(flow), ref)
    fb.close()


@pytest.mark.parametrize("shape,kw", [((270, 480), dict()), ((480, 854), dict()), ((135, 241), dict(levels=2, winsize=7)),
                                      ((200, 300), dict(levels=0)), ((97, 113), dict(levels=1, winsize=11, iterations=2))])
@pytest.mark.parametrize("flags", [4, 256, 260])
def test_full_calc_with_flags_close(FB, shape, kw, flags):
    """The whole call with OPTFLOW_USE_INITIAL_FLOW and / or OPTFLOW_FARNEBACK_GAUSSIAN against the oracle, the
    initial flow being the previous frame pair's result as cv.py:478 passes it; and the caller's array stays as it is."""
    h, w = shape
    a, b = synth_pair(h, w, seed=77)
    _, c = synth_pair(h, w, seed=77, shift=(4.0, 2.5))
    first = O.calc(a, b, **kw)
    ref = O.calc(b, c, flags=flags, flow=first, **kw)
    fb = FB(w, h, flags=flags, **kw)
    init = first.copy()
    got = fb.calc(b, c, flow=init)
    np.testing.assert_array_equal(init, first)
    err = np.abs(got - ref).max()
    assert err <= flow_tol(ref), f"flags={flags}: max|d|={err} tol={flow_tol(ref)}"
    if flags & 4:       # no flow given = zeros (cv.py:478 before the first frame) = the plain call
        zero = fb.calc(b, c)
        plain = O.calc(b, c, flags=flags & 256, **kw)
        assert np.abs(zero - plain).max() <= flow_tol(plain)
    fb.close()


def test_initial_flows_of_a_batch(FB, lib_option):
    """Resident path: every pair of a batch starts from its own initial flow (tf_fb_set_initial_flow / the device
    address), on the two-kernel levels and -- forced -- through the one-kernel iteration at the coarsest scale."""
    h, w = 135, 241
    frames = [synth_pair(h, w, seed=78, shift=(0.9 * i, -0.6 * i))[1] for i in range(4)]
    kw = dict(levels=2, winsize=7)
    inits = [np.random.default_rng(80 + i).normal(0, 1.5, (h, w, 2)).astype(np.float32) for i in range(3)]
    refs = [O.calc(frames[i], frames[i + 1], flags=4, flow=inits[i], **kw) for i in range(3)]
    for fused in (-1, 1):
        lib_option("fb_fused", fused)
        fb = FB(w, h, flags=4, max_pairs=3, frame_slots=4, **kw)
        for i, f in enumerate(frames):
            fb.set_frame(i, f)
        for i in range(3):
            fb.set_initial_flow(i, inits[i])
        assert fb.initial_flow_ptr(1) - fb.initial_flow_ptr(0) == h * w * 8
        fb.calc_slots([0, 1, 2], [1, 2, 3])
        for i in range(3):
            got = fb.get_flow(i)
            assert np.abs(got - refs[i]).max() <= flow_tol(refs[i]), (fused, i)
        fb.close()


# ---- option fb_exact_sums: the box window summed in OpenCV's own order -------------------------------------------
@pytest.mark.parametrize("shape,winsize", [((64, 64), 15), ((67, 531), 15), ((270, 480), 15), ((40, 50), 9),
                                           ((33, 300), 4), ((9, 11), 15), ((5, 7), 15), ((1, 1), 15), ((300, 200), 1),
                                           ((130, 70), 3), ((31, 33), 21)])
def test_exact_sums_blur_solve_bit_identical(FB, lib_option, shape, winsize):
    """FarnebackUpdateFlow_Blur keeps one set of running sums per image (float-differenced down the columns from
    row 0, double-differenced along the rows from column 0); with fb_exact_sums the kernels repeat that order and the
    flow is the oracle's, bit for bit -- windows wider than the frame, one-pixel frames and winsize 1 included."""
    lib_option("fb_exact_sums", 1)
    h, w = shape
    rng = np.random.default_rng(5)
    r = rng.normal(0, 3, (h, w, 5)).astype(np.float32)
    m = O.update_matrices(r, rng.normal(0, 3, (h, w, 5)).astype(np.float32), np.zeros((h, w, 2), np.float32))
    ref, _ = O.update_flow_blur(r, r, np.zeros((h, w, 2), np.float32), m, winsize, False)
    fb = FB(max(w, 32), max(h, 32), levels=0, winsize=winsize)
    got = fb.stage_blur_solve(m)
    fb.close()
    np.testing.assert_array_equal(got, ref)


@pytest.mark.parametrize("shape,kw", CASES + SWEEP)
def test_exact_sums_whole_call_bit_identical(FB, lib_option, shape, kw):
    """A1, A2, A3 and A5 are bit-identical to the oracle stage by stage; with the window sums in OpenCV's order A4 is
    too, and so is the whole pyramid: every difference the default mode shows against the oracle is summation order."""
    lib_option("fb_exact_sums", 1)
    h, w = shape
    a, b = synth_pair(h, w, seed=21)
    ref = O.calc(a, b, **kw)
    fb = FB(w, h, **kw)
    got = fb.calc(a, b)
    fb.close()
    np.testing.assert_array_equal(got, ref)


def test_exact_sums_random_configurations_bit_identical(FB, lib_option):
    lib_option("fb_exact_sums", 1)
    rng = np.random.default_rng(404)
    shapes = [(1, 1), (2, 3), (1, 64), (64, 1), (9, 9), (10, 10), (31, 33)]
    shapes += [(int(rng.integers(1, 200)), int(rng.integers(1, 300))) for _ in range(25)]
    for h, w in shapes:
        kw = dict(levels=int(rng.integers(0, 5)), pyr_scale=float(rng.choice([0.5, 0.6, 0.75, 0.8])),
                  winsize=int(rng.choice([1, 3, 5, 7, 9, 11, 15, 21, 25])), iterations=int(rng.integers(1, 4)),
                  poly_n=int(rng.choice([5, 7])))
        kw["poly_sigma"] = 1.2 if kw["poly_n"] == 5 else 1.5
        a, b = synth_pair(h, w, seed=h * 1000 + w, shift=(1.7, -0.9), noise=3.0)
        ref = O.calc(a, b, **kw)
        fb = FB(w, h, **kw)
        got = fb.calc(a, b)
        fb.close()
        np.testing.assert_array_equal(got, ref, err_msg=f"{h}x{w} {kw}")


def test_exact_sums_batch_with_initial_flow_bit_identical(FB, lib_option):
    """OPTFLOW_USE_INITIAL_FLOW keeps the box window: exact there too (the Gaussian window has no running sums)."""
    lib_option("fb_exact_sums", 1)
    h, w = 270, 480
    a, b = synth_pair(h, w, seed=9)
    init = np.random.default_rng(1).normal(0, 1.5, (h, w, 2)).astype(np.float32)
    ref = O.calc(a, b, flags=4, flow=init)
    fb = FB(w, h, flags=4)
    got = fb.calc(a, b, flow=init)
    fb.close()
    np.testing.assert_array_equal(got, ref)


# ---- what bench.py times, against the oracle at the bench's own shapes (strict tolerance) --------------------------
def clip_frames(h, w, n, seed, step=(0.6, 0.4)):
    """n consecutive frames of one scene: the same texture seen through a displacement that grows with the frame index."""
    return [synth_pair(h, w, seed=seed, shift=(step[0] * i, step[1] * i))[1] for i in range(n)]


def outliers(got, ref):
    d = np.abs(got - ref).max(axis=2)
    return int((d > flow_tol(ref)).sum()), float(d.max())


def assert_within_tolerance(got, ref, what):
    """The default mode's bar since round 4 (bench.py's parity gate applies the same): EVERY pixel within flow_tol(ref) of
    the oracle.  Rounds 2 and 3 excused a few pixels near the frame edges: FarnebackUpdateMatrices' in-frame test is
    discontinuous in the flow, and the ~1e-7 by which per-segment window sums differed from OpenCV's image-long running
    sums decided it where a sample point sat within float resolution of the frame's last row / column.  The marching
    kernels now keep OpenCV's column sums (ColumnCarry in fb_iterate.hip) and compute M without FMA contraction; what is
    left between them and the oracle is the association of double additions.  Returns the number of pixels that differ
    from the oracle at all."""
    d = np.abs(got - ref).max(axis=2)
    bad = d > flow_tol(ref)
    ys, xs = np.nonzero(bad)
    where = f"rows {ys.min()}..{ys.max()}, cols {xs.min()}..{xs.max()}" if bad.any() else "none"
    assert not bad.any(), f"{what}: {int(bad.sum())} pixels beyond tolerance ({where}), max|d|={d.max()}, tol={flow_tol(ref)}"
    return int((d > 0).sum())


@pytest.mark.parametrize("shape,kw", [((480, 854), dict()), ((1080, 1920), dict(levels=5))])
def test_default_mode_within_tolerance_of_every_variant_build_of_the_oracle(FB, shape, kw):
    """The envelope: the oracle is the scalar statement built without FMA contraction and nothing
    pins it to a real OpenCV build.  Builds a real OpenCV could be -- FMA-contracted bodies, INTER_AREA's scalar-tail
    order on a half-size level -- and the float expansion (oracle/Makefile `variants`) sit a few percent of the
    tolerance from it (tests/test_oracle_farneback.py); so the HIP library's default mode is within tolerance of every
    one of them as well, not only of the build it is bit-compared with."""
    h, w = shape
    a, b = synth_pair(h, w, seed=70)
    fb = FB(w, h, **kw)
    got = fb.calc(a, b)
    fb.close()
    assert_within_tolerance(got, O.calc(a, b, **kw), "the parity target")
    for v in O.VARIANTS:
        assert_within_tolerance(got, O.calc(a, b, variant=v, **kw), f"variant build `{v}`")


def test_at_4k_a_variant_build_parts_ways_only_in_a_patch_at_the_frame_border(FB):
    """What the envelope is NOT: at 4K the algorithm's one discontinuity -- FarnebackUpdateMatrices' in-frame test at a
    pixel whose sample point sits within float resolution of the frame's last row -- is decided the other way by the
    FMA-contracted build of the oracle on pair 0 of bench.py's clip (profiles/r06_oracle_envelope.txt: 207 pixels up to
    2.9 x the tolerance in rows 2146-2159), as a real FMA build of OpenCV would against a scalar one.  The HIP default
    mode follows the scalar statement there: no pixel beyond tolerance against the parity target, and against the FMA
    build exactly such a patch -- a few hundred pixels, all within two windows of the bottom row -- and nothing else."""
    import bench
    clip = bench.ClipSynth(2160, 3840, 256, 2000)
    prev, nxt = clip.frame(1), clip.frame(0)                      # BACKWARD order (cv.py:470-472), as the bench calls it
    fb = FB(3840, 2160, levels=5)
    got = fb.calc(prev, nxt)
    fb.close()
    ref = O.calc(prev, nxt, levels=5)
    assert_within_tolerance(got, ref, "4K pair 0 against the parity target")
    fma = O.calc(prev, nxt, levels=5, variant="fma")
    d = np.abs(got - fma).max(axis=2)
    ys, xs = np.nonzero(d > flow_tol(fma))
    assert 0 < len(ys) < 1000, f"{len(ys)} pixels beyond tolerance against the FMA build"
    assert ys.min() >= 2160 - 2 * 15 and xs.max() - xs.min() < 4 * 15, (ys.min(), ys.max(), xs.min(), xs.max())
    assert float(np.median(d)) <= 0.1 * flow_tol(fma)             # everywhere else: a few percent of the tolerance


def test_against_cv2_fixtures_when_present(FB, lib_option):
    """tools/pin_with_cv2.py's file, the day one is committed under tests/golden/: the HIP library against a real
    OpenCV build's flows -- default mode within the tolerance at every pixel, exact mode reported (bit-identical if that
    build contracts nothing)."""
    from tests.helpers import cv2_fixture_cases, cv2_fixture_files
    files = cv2_fixture_files()
    if not files:
        pytest.skip("no tests/golden/farneback_cv2_*.npz: run tools/pin_with_cv2.py where `import cv2` works (PARITY UNPINNED until then)")
    for path in files:
        meta, cases, skipped = cv2_fixture_cases(path)
        assert cases, f"{path}: no case regenerates its inputs here: {skipped}"
        for mode in (0, 1):
            lib_option("fb_exact_sums", mode)
            for c, a, b, init, ref in cases:
                fb = FB(c["w"], c["h"], flags=c["flags"], **c["params"])
                got = fb.calc(a, b, flow=init) if init is not None else fb.calc(a, b)
                fb.close()
                assert_within_tolerance(got, ref, f"{path} {c['key']} (exact={mode}) against cv2 {meta['cv2_version']}")
                print(f"{c['key']} exact={mode}: {int((got != ref).any(axis=2).sum())} pixels differ from cv2 {meta['cv2_version']}")


def test_bench_shape_1080p_levels5_four_consecutive_pairs_forward_remap(FB, lib_option):
    """BASELINE configs[2] as benched: 1080p, levels=5, consecutive pairs of one call sharing their frames'
    expansions, levels 0-1 on the one-kernel iteration (the default above 4 M pixels per level over the batch), then per
    pair the FORWARD scatter and the remap step that finishes post_process in registers (clip_flow=2).  Flow within
    tolerance at every pixel, and bit-identical with fb_exact_sums; layer state, rgba and frame bit-exact."""
    from oracle import remap_ref as OR
    from transflow_amd.remap import CompImage, RemapLayer
    h, w, P = 1080, 1920, 4
    frames = clip_frames(h, w, P + 1, seed=700)
    refs = [O.calc(frames[i], frames[i + 1], levels=5) for i in range(P)]
    fb = FB(w, h, levels=5, frame_slots=P + 1, max_pairs=P)
    for i, f in enumerate(frames):
        fb.set_frame(i, f)
    fb.calc_slots(list(range(P)), list(range(1, P + 1)))
    flows = [fb.get_flow(i) for i in range(P)]
    for i in range(P):
        assert_within_tolerance(flows[i], refs[i], f"pair {i}")
    pixmap = np.random.default_rng(1237).integers(0, 256, (h, w, 3), dtype=np.uint8)
    layer = RemapLayer(h, w)
    layer.set_sources([np.ones((h, w), np.uint8)])
    comp = CompImage(h, w, (255, 255, 255))
    ora = OR.MoveRefLayer(h, w, OR.LayerParams(), introduction_masks=[np.ones((h, w), bool)])
    from transflow_amd.device import DevBuffer
    pix = DevBuffer.from_array(pixmap)
    white = np.full((h, w, 3), 255, np.uint8)
    for i in range(P):
        layer.step_dev(comp, fb.post_process_scatter(i), pix.ptr, 3, clip_flow=2)
        ora.update(OR.post_process(flows[i].copy(), OR.FORWARD), [pixmap], None)
        data, rgba = layer.get_state()
        np.testing.assert_array_equal(data, ora.data)
        np.testing.assert_array_equal(rgba, ora.rgba)
        np.testing.assert_array_equal(comp.download(), OR.composite(white, [ora.render()]))
    assert not layer.out_of_frame()
    lib_option("fb_exact_sums", 1)
    fb.calc_slots(list(range(P)), list(range(1, P + 1)))
    for i in range(P):
        np.testing.assert_array_equal(fb.get_flow(i), refs[i])
    fb.close()


def test_bench_shape_1080p_one_scale(FB, lib_option):
    """BASELINE configs[1]: 1080p, a single scale (levels=0), a batch of pairs, both iteration forms."""
    h, w, P = 1080, 1920, 3
    frames = clip_frames(h, w, P + 1, seed=800)
    refs = [O.calc(frames[i], frames[i + 1], levels=0) for i in range(P)]
    for fused in (1, 0):
        lib_option("fb_fused", fused)
        fb = FB(w, h, levels=0, frame_slots=P + 1, max_pairs=P)
        for i, f in enumerate(frames):
            fb.set_frame(i, f)
        fb.calc_slots(list(range(P)), list(range(1, P + 1)))
        for i in range(P):
            assert_within_tolerance(fb.get_flow(i), refs[i], f"fb_fused={fused} pair {i}")
        fb.close()
    lib_option("fb_exact_sums", 1)
    fb = FB(w, h, levels=0, frame_slots=P + 1, max_pairs=P)
    for i, f in enumerate(frames):
        fb.set_frame(i, f)
    fb.calc_slots(list(range(P)), list(range(1, P + 1)))
    for i in range(P):
        np.testing.assert_array_equal(fb.get_flow(i), refs[i])
    fb.close()


def test_bench_shape_4k_two_pairs_sharing_a_frame_fused_on_every_level(FB, lib_option):
    """BASELINE configs[3]/[4] as benched: 4K, levels=5, two pairs that share a frame (its expansion computed once),
    the one-kernel iteration on every level.  Every pixel within tolerance (round 3 saw 207 border pixels of such a pair
    beyond it); the exact mode reproduces the oracle bit for bit."""
    h, w = 2160, 3840
    frames = clip_frames(h, w, 3, seed=900)
    refs = [O.calc(frames[i], frames[i + 1], levels=5) for i in range(2)]
    lib_option("fb_fused", 1)
    fb = FB(w, h, levels=5, frame_slots=3, max_pairs=2)
    for i, f in enumerate(frames):
        fb.set_frame(i, f)
    fb.calc_slots([0, 1], [1, 2])
    got = [fb.get_flow(i) for i in range(2)]
    lib_option("fb_exact_sums", 1)
    fb.calc_slots([0, 1], [1, 2])
    for i in range(2):
        np.testing.assert_array_equal(fb.get_flow(i), refs[i])
    fb.close()
    for i in range(2):
        assert_within_tolerance(got[i], refs[i], f"pair {i}")


@pytest.mark.parametrize("fused", [1, 0])
def test_row_segments_keep_the_column_sums_of_the_whole_march(FB, lib_option, fused):
    """A column of the marching kernels cut into row segments (option fb_segs) must give the flow of the column marched
    whole: with the carries handed down inside the launch (fb_chain = 1; one-kernel iteration only) the column sums are the
    same additions in the same order, so the flow is BIT-IDENTICAL; with the carries from a pre-pass (fb_chain = 0) the
    sums differ by the association of double additions (~1e-16) and the flow may differ in a last bit here and there.
    All of them: every pixel within tolerance of the oracle.  Heights that do not divide, segments shorter than the window."""
    lib_option("fb_fused", fused)
    for (h, w, kw, P) in [(270, 480, dict(levels=2), 3), (97, 211, dict(levels=1, winsize=11), 2), (61, 40, dict(levels=0, winsize=7, iterations=2), 1)]:
        frames = clip_frames(h, w, P + 1, seed=1000 + h)
        refs = [O.calc(frames[i], frames[i + 1], **kw) for i in range(P)]

        def run(segs, chain):
            lib_option("fb_segs", segs)
            lib_option("fb_chain", chain)
            fb = FB(w, h, frame_slots=P + 1, max_pairs=P, **kw)
            for i, f in enumerate(frames):
                fb.set_frame(i, f)
            fb.calc_slots(list(range(P)), list(range(1, P + 1)))
            out = [fb.get_flow(i) for i in range(P)]
            fb.close()
            return out
        whole = run(1, -1)
        for i in range(P):
            assert_within_tolerance(whole[i], refs[i], f"{h}x{w} whole pair {i}")
        for segs in (2, 3, 7, 64):
            for chain in ((1, 0) if fused else (0,)):
                got = run(segs, chain)
                for i in range(P):
                    what = f"{h}x{w} fb_fused={fused} fb_segs={segs} fb_chain={chain} pair {i}"
                    assert_within_tolerance(got[i], refs[i], what)
                    if chain == 1:
                        np.testing.assert_array_equal(got[i], whole[i], err_msg=what)
                    else:
                        assert (got[i] != whole[i]).any(axis=2).mean() < 1e-4, what


def test_segment_handoff_at_a_size_that_fills_the_chip(FB, lib_option):
    """The hand-off inside the launch where it is the default: 1080p x 12 pairs has 216 columns of workgroups, forced to 8
    segments = 1728 workgroups for 768 slots, so later segments are dispatched while earlier ones run, wait for them and
    take their sums.  Bit-identical to the whole march, twice (replay), and within tolerance of the oracle on a sample."""
    h, w, P = 1080, 1920, 12
    frames = clip_frames(h, w, P + 1, seed=1100)
    lib_option("fb_fused", 1)

    def run(segs, chain):
        lib_option("fb_segs", segs)
        lib_option("fb_chain", chain)
        fb = FB(w, h, levels=1, frame_slots=P + 1, max_pairs=P)
        for i, f in enumerate(frames):
            fb.set_frame(i, f)
        out = []
        for _ in range(2):
            fb.calc_slots(list(range(P)), list(range(1, P + 1)))
            out.append([fb.get_flow(i) for i in range(P)])
        fb.close()
        return out
    whole = run(1, -1)[0]
    for rep in run(8, 1):
        for i in range(P):
            np.testing.assert_array_equal(rep[i], whole[i], err_msg=f"pair {i}")
    ref = O.calc(frames[5], frames[6], levels=1)
    assert_within_tolerance(whole[5], ref, "pair 5")


def test_two_lanes_give_the_results_of_one(FB):
    """Farneback(lanes=2): calls alternate between a handle and its lane (tf_fb_create_lane: the same frame slots, the
    library's other call stream), consecutive calls are in flight together, and every call's flow is what one handle
    computes for it -- bit for bit, whichever lane ran it."""
    h, w, P = 270, 480, 3
    frames = clip_frames(h, w, 3 * P + 1, seed=1200)
    calls = [(list(range(c * P, c * P + P)), list(range(c * P + 1, c * P + P + 1))) for c in range(3)] * 2
    one = FB(w, h, frame_slots=len(frames), max_pairs=P)
    two = FB(w, h, frame_slots=len(frames), max_pairs=P, lanes=2)
    for i, f in enumerate(frames):
        one.set_frame(i, f)
        two.set_frame(i, f)
    want = []
    for prev, nxt in calls:
        one.calc_slots(prev, nxt)
        want.append([one.get_flow(i) for i in range(P)])
    # issue two calls back to back (both lanes busy), then read the second; then interleave reading and issuing
    two.calc_slots(*calls[0])
    two.calc_slots(*calls[1])
    for i in range(P):
        np.testing.assert_array_equal(two.get_flow(i), want[1][i])
    for c in range(2, len(calls)):
        two.calc_slots(*calls[c])
        for i in range(P):
            np.testing.assert_array_equal(two.get_flow(i), want[c][i], err_msg=f"call {c} pair {i}")
    with pytest.raises(ValueError):
        two.keep_expansions(True)
    one.close()
    two.close()


def test_async_io_keeps_a_frame_going_up_and_a_flow_coming_down_beside_the_kernels(FB):
    """tf_fb_async_io for a streaming caller (cv.py:460-490 per frame): frames go up on the library's upload stream, flows
    come down on its download stream.  A two-deep pipeline -- issue pair t + 1 (new frame into the slot the older frame
    held, expansions kept), THEN end pair t's download -- yields the flows of the plain sequence bit for bit, and the
    arrays handed out stay intact."""
    from transflow_amd.device import pinned_empty
    h, w, n = 270, 480, 7
    frames = clip_frames(h, w, n + 1, seed=1300)
    plain = FB(w, h, frame_slots=2, max_pairs=2)
    plain.keep_expansions(True)
    plain.set_frame(0, frames[0])
    want = []
    for t in range(n):
        plain.set_frame((t + 1) & 1, frames[t + 1])
        plain.calc_slots([(t + 1) & 1], [t & 1])
        want.append(plain.get_flow(0))
    plain.close()
    fb = FB(w, h, frame_slots=2, max_pairs=2)
    fb.keep_expansions(True)
    fb.async_io(True)
    with pytest.raises(ValueError):
        FB(w, h).get_flow_begin(0, np.empty((h, w, 2), np.float32))    # async io is off on that handle
    fb.set_frame(0, frames[0])
    outs = [pinned_empty((h, w, 2), np.float32) for _ in range(n)]
    held = None
    for t in range(n):
        fb.set_frame((t + 1) & 1, frames[t + 1])
        fb.calc_slots([(t + 1) & 1], [t & 1])
        tok = fb.get_flow_begin(0, outs[t])
        if held is not None:
            fb.get_flow_end(held)
        held = tok
    fb.get_flow_end(held)
    fb.get_flow_end(held)                                              # nothing pending: a no-op
    for t in range(n):
        np.testing.assert_array_equal(outs[t], want[t], err_msg=f"pair {t}")
    fb.async_io(False)
    fb.close()


def test_exact_sums_with_the_column_sums_straight_from_the_expansions(FB, lib_option):
    """fb_exact_sums where the iteration never stores M (fb_fused = 1: k_flow_carry_pc<.., STORE> marches every column once
    and stores OpenCV's chain at every row, k_exact_hsolve runs the rows): bit-identical to the oracle for widths that are
    not whole waves or whole 64-column chunks (113, 225, 339), heights that are not whole groups of three rows (97, 64),
    every window of the one-kernel form, the three sources of the flow (zero, the previous iteration, the coarser
    level), a batch, and twice (replay)."""
    lib_option("fb_exact_sums", 1)
    lib_option("fb_fused", 1)
    for (h, w, kw, P) in [(120, 113, dict(levels=1), 1), (97, 339, dict(levels=2, winsize=11), 2), (64, 225, dict(levels=0, winsize=7, iterations=2), 3),
                          (270, 480, dict(levels=2), 2)]:
        frames = clip_frames(h, w, P + 1, seed=1400 + w)
        refs = [O.calc(frames[i], frames[i + 1], **kw) for i in range(P)]
        fb = FB(w, h, frame_slots=P + 1, max_pairs=P, **kw)
        for i, f in enumerate(frames):
            fb.set_frame(i, f)
        for rep in range(2):
            fb.calc_slots(list(range(P)), list(range(1, P + 1)))
            for i in range(P):
                np.testing.assert_array_equal(fb.get_flow(i), refs[i], err_msg=f"{h}x{w} {kw} pair {i} run {rep}")
        fb.close()


def test_exact_sums_at_a_size_where_that_form_is_the_default(FB, lib_option):
    """1080p x 32 pairs: 576 columns of workgroups at level 0, above the threshold where fb_exact_sums takes its column sums
    straight from the expansions by itself; level 1 goes through M in memory: a sample of the pairs bit-identical to the
    oracle."""
    lib_option("fb_exact_sums", 1)
    h, w, P = 1080, 1920, 32
    frames = clip_frames(h, w, P + 1, seed=1500)
    fb = FB(w, h, levels=1, frame_slots=P + 1, max_pairs=P)
    for i, f in enumerate(frames):
        fb.set_frame(i, f)
    fb.calc_slots(list(range(P)), list(range(1, P + 1)))
    for i in (0, 13, 31):
        np.testing.assert_array_equal(fb.get_flow(i), O.calc(frames[i], frames[i + 1], levels=1), err_msg=f"pair {i}")
    fb.close()


def test_bgr_frame_into_a_slot_matches_the_host_conversion_and_validates(FB):
    """tf_fb_set_frame_bgr (cv.py:461-466 on the device): a strided BGR view, a frame of another size (nearest resize),
    and its argument checks; the flow computed from slots filled that way equals the flow from the oracle's grey frames."""
    from oracle import frames_ref
    h, w = 96, 128
    rng = np.random.default_rng(12)
    big = rng.integers(0, 256, (h + 7, w + 9, 3), dtype=np.uint8)
    view = big[3:3 + h, 4:4 + w]                      # row stride larger than the row
    other = rng.integers(0, 256, (61, 83, 3), dtype=np.uint8)
    fb = FB(w, h)
    fb.set_frame_bgr(0, view)
    fb.set_frame_bgr(1, other)                         # 83 x 61 -> 128 x 96 by nearest neighbour
    fb.calc_slots([0], [1])
    got = fb.get_flow(0)
    ga, gb = frames_ref.bgr_to_grey(np.ascontiguousarray(view)), frames_ref.bgr_to_grey(other, (w, h))
    ref = O.calc(ga, gb)
    assert np.abs(got - ref).max() <= flow_tol(ref)
    fb2 = FB(w, h)
    np.testing.assert_array_equal(fb2.calc(ga, gb), got)   # the same flow as from host-side grey frames
    fb2.close()
    with pytest.raises(ValueError):
        fb.set_frame_bgr(2, view)                      # slot out of range
    with pytest.raises(ValueError):
        fb.set_frame_bgr(0, view[:, :, :2])            # not three channels
    with pytest.raises(ValueError):
        fb.set_frame_bgr(0, view.astype(np.float32))   # not uint8
    fb.close()
