"""Pins for the C restatement of OpenCV's Farnebäck (oracle/farneback_ref.c).

The reference holds no flow values for this path (its tests/test_flow_source.py
:22-33 check type/shape/dtype only) and cv2 is absent from the build image, so
these are analytic known-answer tests plus a
comparison with cv2 that runs wherever ``import cv2`` works.  CPU only.
"""
import numpy as np
import pytest

from oracle import farneback as F


def _texture(h, w, tx=0.0, ty=0.0, seed=5):
    rng = np.random.default_rng(seed)
    comps = [(rng.uniform(.4, 1), rng.uniform(.01, .08) * rng.choice([-1, 1]), rng.uniform(.01, .08),
              rng.uniform(0, 6.28)) for _ in range(8)]
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    v = sum(a * np.sin(2 * np.pi * (fx * (xx - tx) + fy * (yy - ty)) + ph) for a, fx, fy, ph in comps)
    return np.clip(np.rint(128 + 25 * v), 0, 255).astype(np.uint8)


def test_gaussian_kernels():
    np.testing.assert_array_equal(F.gaussian_kernel(3, 0.0), np.float32([.25, .5, .25]))
    for n, s in [(3, .5), (9, 1.5), (19, 3.5), (39, 7.5), (79, 15.5)]:
        k = F.gaussian_kernel(n, s)
        assert abs(float(k.astype(np.float64).sum()) - 1) < 1e-6
        np.testing.assert_array_equal(k, k[::-1])
        x = np.arange(n) - (n - 1) / 2
        ref = np.exp(-x * x / (2 * s * s))
        np.testing.assert_allclose(k, ref / ref.sum(), rtol=2e-7)


def test_level_schedule_matches_survey_a1():
    # 1080p, pyr_scale .5, levels 5 -> scales k=5..0
    assert F.num_levels(1920, 1080, .5, 5) == 5
    assert F.num_levels(1920, 1080, .5, 3) == 3
    assert F.num_levels(1920, 1080, .5, 0) == 0
    assert F.num_levels(64, 64, .5, 5) == 1          # 32 ok, 16 < min_size
    assert F.num_levels(854, 480, .5, 3) == 3
    geo = [F.level_geometry(1920, 1080, .5, k) for k in range(6)]
    assert [(g[0], g[1], g[2]) for g in geo] == [(1920, 1080, 3), (960, 540, 3), (480, 270, 9), (240, 135, 19),
                                                 (120, 68, 39), (60, 34, 79)]
    assert [g[3] for g in geo] == [0.0, 0.5, 1.5, 3.5, 7.5, 15.5]
    assert F.level_geometry(3840, 2160, .5, 5)[:2] == (120, 68)


def test_resize_linear_identity_and_halving():
    rng = np.random.default_rng(0)
    a = rng.normal(0, 50, (12, 20)).astype(np.float32)
    np.testing.assert_array_equal(F.resize_linear(a, 20, 12), a)
    half = F.resize_linear(a, 10, 6)
    exp = ((a[0::2, 0::2] * np.float32(.5) + a[0::2, 1::2] * np.float32(.5)) * np.float32(.5)
           + (a[1::2, 0::2] * np.float32(.5) + a[1::2, 1::2] * np.float32(.5)) * np.float32(.5))
    np.testing.assert_array_equal(half, exp)
    # upscaling a 2-channel field: corners replicate
    f = rng.normal(0, 3, (5, 7, 2)).astype(np.float32)
    up = F.resize_linear(f, 14, 10)
    np.testing.assert_array_equal(up[0, 0], f[0, 0] * np.float32(.25) + f[0, 0] * np.float32(.75))
    np.testing.assert_allclose(up[-1, -1], f[-1, -1], rtol=1e-6)


def test_blur_constant_and_reflect():
    img = np.full((20, 30), 77, np.uint8)
    for ksz, s in [(3, 0.0), (9, 1.5), (19, 3.5)]:
        np.testing.assert_allclose(F.gaussian_blur_u8(img, ksz, s), 77, rtol=1e-6)
    # 3-tap at the border uses REFLECT_101: column -1 == column 1
    img = np.zeros((4, 6), np.uint8)
    img[:, 1] = 100
    out = F.gaussian_blur_u8(img, 3, 0.0)
    assert out[2, 0] == 50.0 and out[2, 1] == 50.0 and out[2, 2] == 25.0


def test_polyexp_reproduces_quadratic():
    """I = c + a x + b y + d x^2 + e y^2 + f xy  =>  R = (b', a', e, d, f) at every
    interior pixel, with a', b' the gradient at that pixel (SURVEY A.3 store order:
    R0 ~ y, R1 ~ x, R2 ~ yy, R3 ~ xx, R4 ~ xy)."""
    h, w, n = 40, 48, 5
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    c, a, b, d, e, f = 3.0, 0.5, -0.25, 0.02, -0.03, 0.01
    img = (c + a * xx + b * yy + d * xx * xx + e * yy * yy + f * xx * yy).astype(np.float32)
    r = F.polyexp(img, n, 1.2)[n:-n, n:-n].astype(np.float64)
    x, y = xx[n:-n, n:-n], yy[n:-n, n:-n]
    np.testing.assert_allclose(r[..., 0], b + 2 * e * y + f * x, atol=2e-4)
    np.testing.assert_allclose(r[..., 1], a + 2 * d * x + f * y, atol=2e-4)
    np.testing.assert_allclose(r[..., 2], e, atol=2e-5)
    np.testing.assert_allclose(r[..., 3], d, atol=2e-5)
    np.testing.assert_allclose(r[..., 4], f, atol=2e-5)


def test_identical_frames_zero_flow_in_interior():
    a = _texture(160, 200)
    # OpenCV treats R1 as absent on the last row/column (UpdateMatrices' bounds test), so a
    # small residual exists near the bottom/right edge; the box window (m=7) spreads it
    # 7 px per iteration at one scale, further through the coarse scales of a pyramid
    fl = F.calc(a, a, levels=0, iterations=1)
    assert np.abs(fl[:-8, :-8]).max() == 0.0
    assert np.abs(fl).max() > 0.0
    a = _texture(320, 400)
    fl = F.calc(a, a, levels=2)
    assert np.abs(fl[:100, :100]).max() == 0.0
    assert np.abs(fl).max() < 0.5


@pytest.mark.parametrize("t", [(2, 1), (-3, 4), (0.5, -0.25)])
def test_translation_recovered(t):
    a, b = _texture(200, 260), _texture(200, 260, *t)
    fl = F.calc(a, b)[30:-30, 30:-30]
    tol = 0.01 if all(float(v).is_integer() for v in t) else 0.1
    assert np.abs(fl[..., 0] - t[0]).max() < tol
    assert np.abs(fl[..., 1] - t[1]).max() < tol


def test_swap_negates():
    a, b = _texture(160, 200), _texture(160, 200, 2, -1)
    f1 = F.calc(a, b)[30:-30, 30:-30]
    f2 = F.calc(b, a)[30:-30, 30:-30]
    assert np.abs(f1 + f2).max() < 0.02


def test_single_scale_and_small_frames():
    a, b = _texture(48, 64), _texture(48, 64, 1, 0)
    fl = F.calc(a, b, levels=0)
    assert fl.shape == (48, 64, 2) and np.isfinite(fl).all()
    # frames smaller than min_size still run (K=0)
    fl = F.calc(a[:20, :24].copy(), b[:20, :24].copy(), levels=3)
    assert np.isfinite(fl).all()
    with pytest.raises(ValueError):
        F.calc(a, b, flags=8)          # only OPTFLOW_USE_INITIAL_FLOW (4) and OPTFLOW_FARNEBACK_GAUSSIAN (256) exist


def test_sensitivity_variants_stay_within_a_quarter_of_the_tolerance():
    """The envelope: the parity target is the scalar statement built without FMA contraction, and
    nothing pins it to a real OpenCV build.  What a real build may legitimately do differently -- contract
    multiply-adds in its SIMD bodies (`fma`), run INTER_AREA's scalar tail over more or fewer columns of a half-size
    level ([VERIFY] 4, `area`) -- and what a cheaper expansion would do (`polyf32`) moves the flow by a few percent of
    the 1e-4 tolerance on every shape of the GPU suites: the tolerance has an order of magnitude of headroom over
    build-level rounding.  (At 4K one discontinuity of the algorithm -- FarnebackUpdateMatrices' in-frame test at a
    border pixel whose sample point sits within float resolution of the last row -- can be decided the other way by
    such a build and move a patch of a few hundred pixels beyond the tolerance: tools/oracle_envelope.py,
    profiles/r06_oracle_envelope.txt.  The shapes here hold no such pixel.)"""
    from tests.helpers import FB_CASES, FB_SWEEP, synth_pair
    worst = {}
    for (h, w), kw in FB_CASES + FB_SWEEP:
        a, b = synth_pair(h, w, seed=70)
        ref = F.calc(a, b, **kw)
        tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
        for v in F.VARIANTS:
            got = F.calc(a, b, variant=v, **kw)
            dev = float(np.abs(got - ref).max()) / tol
            worst[v] = max(worst.get(v, 0.0), dev)
            assert dev <= 0.25, f"{v} build at {w}x{h} {kw}: {dev:.3f} of the tolerance"
    # the builds do differ (the variants are not no-ops), except that `area` needs a half-size level to act on
    assert worst["fma"] > 0 and worst["polyf32"] > 0 and worst["area"] > 0


def test_area_variant_acts_only_on_exactly_half_size_levels():
    from tests.helpers import synth_pair
    a, b = synth_pair(135, 241, seed=3)                     # odd sizes: no level is exactly half the frame
    np.testing.assert_array_equal(F.calc(a, b, levels=2, variant="area"), F.calc(a, b, levels=2))
    a, b = synth_pair(128, 192, seed=3)
    assert not np.array_equal(F.calc(a, b, levels=2, variant="area"), F.calc(a, b, levels=2))


def check_oracle_against_cv2_fixture(path):
    """The oracle against one file of tools/pin_with_cv2.py: every stored flow within 1e-4 * max(1, max|cv2|) -- the
    day such a file exists under tests/golden/ the Farneback half of the parity story is pinned to that OpenCV build.
    Returns (cases compared, cases bit-identical, largest deviation in units of the tolerance)."""
    from tests.helpers import cv2_fixture_cases
    meta, cases, skipped = cv2_fixture_cases(path)
    assert cases, f"{path}: no case regenerates its inputs here (numpy {np.__version__} vs {meta['numpy_version']}): {skipped}"
    identical, worst = 0, 0.0
    for c, a, b, init, ref in cases:
        got = F.calc(a, b, flags=c["flags"], flow=init, **c["params"])
        tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
        dev = float(np.abs(got - ref).max()) / tol
        assert dev <= 1.0, f"{path} {c['key']}: oracle {dev:.3f} x the tolerance from cv2 {meta['cv2_version']}"
        identical += bool(np.array_equal(got, ref))
        worst = max(worst, dev)
    return len(cases), identical, worst


def test_against_cv2_fixtures_when_present():
    from tests.helpers import cv2_fixture_files
    files = cv2_fixture_files()
    if not files:
        pytest.skip("no tests/golden/farneback_cv2_*.npz: run tools/pin_with_cv2.py where `import cv2` works (PARITY UNPINNED until then)")
    for path in files:
        n, same, worst = check_oracle_against_cv2_fixture(path)
        print(f"{path}: {n} flows, {same} bit-identical, worst {worst:.3f} of the tolerance")


def test_against_cv2_when_available():
    cv2 = pytest.importorskip("cv2")
    a, b = _texture(240, 320), _texture(240, 320, 2.5, -1.5)
    for levels in (0, 3):
        ref = cv2.calcOpticalFlowFarneback(a, b, None, 0.5, levels, 15, 3, 5, 1.2, 0)
        got = F.calc(a, b, levels=levels)
        tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= tol


def test_blur_matches_an_independent_separable_filter():
    """GaussianBlur(REFLECT_101) against scipy.ndimage's correlate1d(mode='mirror') with the same taps:
    an independent implementation of the same definition (float64 there, float32 taps/sums here)."""
    from scipy import ndimage
    img = _texture(60, 75)
    for ksz, sigma in [(3, 0.0), (9, 1.5), (19, 3.5), (39, 7.5)]:
        k = F.gaussian_kernel(ksz, sigma).astype(np.float64)
        ref = ndimage.correlate1d(ndimage.correlate1d(img.astype(np.float64), k, axis=1, mode="mirror"), k, axis=0,
                                  mode="mirror")
        np.testing.assert_allclose(F.gaussian_blur_u8(img, ksz, sigma), ref, rtol=0, atol=2e-3)


def test_polyexp_is_the_weighted_least_squares_fit():
    """Farnebäck's expansion is, by definition, the Gaussian-weighted least-squares fit of
    c + bx x + by y + axx x^2 + ayy y^2 + axy xy over the (2n+1)^2 neighbourhood: solved here with
    numpy.linalg.lstsq per pixel and compared with the oracle's separable closed form on a
    non-polynomial image (so the weights matter).  Store order R = (by, bx, ayy, axx, axy), SURVEY A.3."""
    n, sigma = 5, 1.2
    img = (_texture(40, 44).astype(np.float32) / 8).astype(np.float32)
    r = F.polyexp(img, n, sigma).astype(np.float64)
    d = np.arange(-n, n + 1, dtype=np.float64)
    g = np.exp(-d * d / (2 * sigma * sigma))
    g /= g.sum()
    dy, dx = np.meshgrid(d, d, indexing="ij")
    wgt = np.sqrt(np.outer(g, g)).ravel()
    basis = np.stack([np.ones_like(dx), dx, dy, dx * dx, dy * dy, dx * dy], axis=-1).reshape(-1, 6)
    rng = np.random.default_rng(0)
    for _ in range(25):
        i, j = int(rng.integers(n, 40 - n)), int(rng.integers(n, 44 - n))
        patch = img[i - n:i + n + 1, j - n:j + n + 1].astype(np.float64).ravel()
        coef = np.linalg.lstsq(basis * wgt[:, None], patch * wgt, rcond=None)[0]
        np.testing.assert_allclose(r[i, j], [coef[2], coef[1], coef[4], coef[3], coef[5]], rtol=0, atol=2e-4,
                                   err_msg=f"pixel ({i}, {j})")


def test_one_iteration_is_farnebacks_least_squares_displacement():
    """Farnebäck (2003), eq. 9-12, written down independently in float64: with the two expansions
    f(x) ~ x'A x + b'x + c, A = (A0 + A1(x + d)) / 2, db = -(b1(x + d) - b0) / 2 + A d, the displacement
    minimises sum_w |A d - db|^2 over the window: d = (sum A'A)^-1 sum A'db.  The oracle's
    update_matrices + update_flow_blur against that, from a non-zero prior flow, away from the
    borders (where OpenCV down-weights and treats R1 as absent).  The oracle adds 1e-3 to the
    determinant (of sums scaled by 1/winsize^2); so does this."""
    from scipy import ndimage
    n, sigma, winsize = 5, 1.2, 15
    a, b = _texture(120, 150), _texture(120, 150, 1.6, -0.9)
    r0 = F.polyexp(a.astype(np.float32), n, sigma).astype(np.float64)
    r1 = F.polyexp(b.astype(np.float32), n, sigma).astype(np.float64)
    h, w = a.shape
    rng = np.random.default_rng(2)
    prior = np.stack([np.full((h, w), 1.3), np.full((h, w), -0.6)], axis=-1) + rng.normal(0, 0.2, (h, w, 2))
    prior = ndimage.gaussian_filter(prior, (3, 3, 0)).astype(np.float32)      # smooth, non-integer
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    coords = [yy + prior[..., 1], xx + prior[..., 0]]                          # flow = (dx, dy)
    r1w = np.stack([ndimage.map_coordinates(r1[..., c], coords, order=1, mode="nearest") for c in range(5)], axis=-1)
    # channel order (by, bx, ayy, axx, axy): A = [[ayy, axy/2], [axy/2, axx]] in (y, x)
    ayy, axx, axy = (r0[..., 2] + r1w[..., 2]) / 2, (r0[..., 3] + r1w[..., 3]) / 2, (r0[..., 4] + r1w[..., 4]) / 4
    dby = (r0[..., 0] - r1w[..., 0]) / 2 + ayy * prior[..., 1] + axy * prior[..., 0]
    dbx = (r0[..., 1] - r1w[..., 1]) / 2 + axy * prior[..., 1] + axx * prior[..., 0]
    box = lambda v: ndimage.uniform_filter(v, winsize, mode="nearest")        # mean = sum / winsize^2
    gyy, gyx, gxx = box(ayy * ayy + axy * axy), box((ayy + axx) * axy), box(axx * axx + axy * axy)
    hy, hx = box(ayy * dby + axy * dbx), box(axy * dby + axx * dbx)
    det = gyy * gxx - gyx * gyx + 1e-3
    dx, dy = (gyy * hx - gyx * hy) / det, (gxx * hy - gyx * hx) / det
    m = F.update_matrices(r0.astype(np.float32), r1.astype(np.float32), prior)
    got, _ = F.update_flow_blur(r0.astype(np.float32), r1.astype(np.float32), prior, m, winsize, False)
    pad = 5 + winsize // 2 + 3
    inner = (slice(pad, -pad), slice(pad, -pad))
    np.testing.assert_allclose(got[..., 0][inner], dx[inner], rtol=0, atol=2e-4)
    np.testing.assert_allclose(got[..., 1][inner], dy[inner], rtol=0, atol=2e-4)
    # and the step goes towards the true displacement
    assert abs(np.median(got[..., 0][inner]) - 1.6) < 0.1 and abs(np.median(got[..., 1][inner]) + 0.9) < 0.1


# ---- fb_flags: OPTFLOW_USE_INITIAL_FLOW (4) and OPTFLOW_FARNEBACK_GAUSSIAN (256) ------------------------------

def _area_weights(ssize, dsize):
    """Independent statement of INTER_AREA's coverage: destination cell [d*s, (d+1)*s) over unit source cells."""
    s = ssize / dsize
    w = np.zeros((dsize, ssize))
    for d in range(dsize):
        lo, hi = d * s, min((d + 1) * s, ssize)
        for x in range(int(np.floor(lo)), int(np.ceil(hi))):
            w[d, x] = max(0.0, min(hi, x + 1) - max(lo, x))
        w[d] /= w[d].sum()
    return w


@pytest.mark.parametrize("src,dst", [((64, 96), (32, 48)), ((60, 90), (20, 30)), ((135, 241), (34, 60)), ((67, 53), (9, 7)),
                                     ((480, 854), (60, 107)), ((40, 40), (40, 40))])
def test_resize_area_is_the_cell_coverage_average(src, dst):
    rng = np.random.default_rng(21)
    x = rng.normal(0, 3, (*src, 2)).astype(np.float32)
    got = F.resize_area(x, dst[1], dst[0])
    wy, wx = _area_weights(src[0], dst[0]), _area_weights(src[1], dst[1])
    ref = np.einsum("ys,sxc->yxc", wy, np.einsum("xs,ysc->yxc", wx, x.astype(np.float64)))
    np.testing.assert_allclose(got, ref, atol=2e-6 * np.abs(x).max() * 4)
    if src[0] % dst[0] == 0 and src[1] % dst[1] == 0 and src != dst:
        fy, fx = src[0] // dst[0], src[1] // dst[1]
        blocks = x.reshape(dst[0], fy, dst[1], fx, 2).transpose(0, 2, 4, 1, 3).reshape(dst[0], dst[1], 2, fy * fx)
        acc = np.zeros(blocks.shape[:3], np.float32)                 # resizeAreaFast_: four at a time, float
        k = 0
        while k + 4 <= fy * fx:
            acc = acc + (((blocks[..., k] + blocks[..., k + 1]) + blocks[..., k + 2]) + blocks[..., k + 3])
            k += 4
        for j in range(k, fy * fx):
            acc = acc + blocks[..., j]
        np.testing.assert_array_equal(got, acc * np.float32(1.0 / (fy * fx)))
    if src == dst:
        np.testing.assert_array_equal(got, x)


def test_area_tables_cover_every_source_cell_once():
    for ssize, dsize in [(854, 107), (2160, 68), (100, 7), (33, 32)]:
        si, di, al = F.area_tab(ssize, dsize)
        assert (np.diff(di) >= 0).all() and di[0] == 0 and di[-1] == dsize - 1
        per_dst = np.bincount(di, weights=al.astype(np.float64), minlength=dsize)
        np.testing.assert_allclose(per_dst, 1.0, atol=1e-6)            # a cell's weights sum to one
        per_src = np.bincount(si, weights=al.astype(np.float64) * (ssize / dsize), minlength=ssize)
        np.testing.assert_allclose(per_src, 1.0, atol=1e-5)            # and every source cell is spent exactly once


def test_gaussian_window_taps_and_update():
    for m in (2, 5, 7, 10):
        k = F.gaussian_window(m)
        i = np.arange(m + 1)
        ref = np.exp(-i * i / (2 * (m * 0.3) ** 2))
        np.testing.assert_allclose(k, ref / (ref[0] + 2 * ref[1:].sum()), rtol=3e-7)
    from scipy import ndimage
    rng = np.random.default_rng(22)
    h, w, win = 37, 53, 15
    r = rng.normal(0, 3, (h, w, 5)).astype(np.float32)
    m = F.update_matrices(r, rng.normal(0, 3, (h, w, 5)).astype(np.float32), np.zeros((h, w, 2), np.float32))
    flow, _ = F.update_flow_gaussian(r, r, np.zeros((h, w, 2), np.float32), m, win, False)
    k = F.gaussian_window(win // 2).astype(np.float64)
    taps = np.concatenate([k[:0:-1], k])
    g = ndimage.correlate1d(ndimage.correlate1d(m.astype(np.float64), taps, axis=0, mode="nearest"), taps, axis=1,
                            mode="nearest")
    idet = 1.0 / (g[..., 0] * g[..., 2] - g[..., 1] ** 2 + 1e-3)
    ref = np.stack([(g[..., 0] * g[..., 4] - g[..., 1] * g[..., 3]) * idet,
                    (g[..., 2] * g[..., 3] - g[..., 1] * g[..., 4]) * idet], axis=-1)
    assert np.abs(flow - ref).max() <= 2e-4 * max(1.0, np.abs(ref).max())   # float sums vs float64 sums


def test_flags_initial_flow_and_gaussian_window():
    a, b = _texture(160, 224), _texture(160, 224, 2.5, -1.5)
    base = F.calc(a, b)
    # a zero initial flow is the plain call, bit for bit (resize of zeros, times the scale)
    np.testing.assert_array_equal(F.calc(a, b, flags=F.OPTFLOW_USE_INITIAL_FLOW, flow=np.zeros_like(base)), base)
    np.testing.assert_array_equal(F.calc(a, b, flags=F.OPTFLOW_USE_INITIAL_FLOW), base)
    # the true displacement as the initial flow keeps the estimate there (interior)
    truth = np.zeros_like(base)
    truth[..., 0], truth[..., 1] = 2.5, -1.5
    warm = F.calc(a, b, flags=F.OPTFLOW_USE_INITIAL_FLOW, flow=truth)
    assert np.abs(warm[20:-20, 20:-20] - truth[20:-20, 20:-20]).max() < 0.1
    # with a single scale and a single iteration the start decides: a far-off initial flow is NOT recovered
    cold = F.calc(a, b, levels=0, iterations=1, flags=F.OPTFLOW_USE_INITIAL_FLOW, flow=truth + 6.0)
    assert np.abs(cold[20:-20, 20:-20] - truth[20:-20, 20:-20]).mean() > 0.5
    # the Gaussian window still recovers the translation; it is a different estimator
    gauss = F.calc(a, b, flags=F.OPTFLOW_FARNEBACK_GAUSSIAN)
    err = np.abs(gauss[25:-25, 25:-25] - truth[25:-25, 25:-25])      # a narrower window (sigma = 2.1): noisier
    assert err.mean() < 0.05 and err.max() < 0.6
    assert not np.array_equal(gauss, base)
    both = F.calc(a, b, flags=F.OPTFLOW_FARNEBACK_GAUSSIAN | F.OPTFLOW_USE_INITIAL_FLOW, flow=truth)
    assert np.abs(both[25:-25, 25:-25] - truth[25:-25, 25:-25]).mean() < 0.05
    # the caller's array is never modified
    init = truth.copy()
    F.calc(a, b, flags=F.OPTFLOW_USE_INITIAL_FLOW, flow=init)
    np.testing.assert_array_equal(init, truth)


def test_flag_calls_against_cv2_when_available():
    cv2 = pytest.importorskip("cv2")
    a, b = _texture(240, 320), _texture(240, 320, 2.5, -1.5)
    first = cv2.calcOpticalFlowFarneback(a, b, None, 0.5, 3, 15, 3, 5, 1.2, 0)
    for flags in (cv2.OPTFLOW_FARNEBACK_GAUSSIAN, cv2.OPTFLOW_USE_INITIAL_FLOW,
                  cv2.OPTFLOW_FARNEBACK_GAUSSIAN | cv2.OPTFLOW_USE_INITIAL_FLOW):
        ref = cv2.calcOpticalFlowFarneback(a, b, first.copy(), 0.5, 3, 15, 3, 5, 1.2, flags)
        got = F.calc(a, b, flags=flags, flow=first)
        tol = 1e-4 * max(1.0, float(np.abs(ref).max()))
        assert np.abs(got - ref).max() <= tol, flags
