"""oracle/frames_ref.py: cv2.resize(INTER_NEAREST) + cv2.cvtColor(COLOR_BGR2GRAY) of the reference's ingest
(transflow/flow/sources/cv.py:461-466), restated from memory -- unpinned until a file of tools/pin_with_cv2.py is
committed.  CPU only."""
import numpy as np
import pytest

from oracle import frames_ref


def test_known_answers_of_the_fixed_point_weights():
    px = np.uint8([[[0, 0, 0], [255, 255, 255], [255, 0, 0], [0, 255, 0], [0, 0, 255], [10, 200, 30]]])   # B, G, R
    # 0.114 B + 0.587 G + 0.299 R, rounded: 29.07, 149.7, 76.2, 127.5(4)
    np.testing.assert_array_equal(frames_ref.bgr_to_grey(px)[0], [0, 255, 29, 150, 76, 128])
    np.testing.assert_array_equal(frames_ref.bgr_to_grey(px, weights="14bit")[0], [0, 255, 29, 150, 76, 128])
    for name, (wb, wg, wr, shift) in frames_ref.WEIGHTS.items():
        assert wb + wg + wr == 1 << shift, name          # white stays white


def test_the_two_roundings_differ_by_one_level_in_a_few_pixels_per_thousand():
    """What `unpinned` can cost for this step: were the target OpenCV's other fixed-point form, the grey frame would
    differ from the one the HIP kernel makes in ~0.27 % of uniformly random pixels, by exactly one level."""
    rng = np.random.default_rng(0)
    frame = rng.integers(0, 256, (480, 854, 3), dtype=np.uint8)
    a = frames_ref.bgr_to_grey(frame).astype(np.int32)
    b = frames_ref.bgr_to_grey(frame, weights="14bit").astype(np.int32)
    d = np.abs(a - b)
    assert d.max() == 1
    frac = float((d != 0).mean())
    assert 0.001 < frac < 0.01, frac
    print(f"15-bit vs 14-bit weights: {100 * frac:.3f} % of pixels one level apart")


def test_nearest_resize_indices():
    frame = np.arange(5 * 7 * 3, dtype=np.uint8).reshape(5, 7, 3)
    g = frames_ref.bgr_to_grey(frame)
    np.testing.assert_array_equal(frames_ref.bgr_to_grey(frame, (7, 5)), g)                 # same size: no resize
    np.testing.assert_array_equal(frames_ref.bgr_to_grey(frame, (14, 10)), np.repeat(np.repeat(g, 2, 0), 2, 1))
    small = frames_ref.bgr_to_grey(frame, (3, 2))
    np.testing.assert_array_equal(small, g[[0, 2]][:, [0, 2, 4]])                          # floor(i * 7/3), floor(j * 5/2)


def test_against_cv2_fixtures_when_present():
    from tests.helpers import cv2_fixture_files, cv2_fixture_grey
    files = cv2_fixture_files()
    if not files:
        pytest.skip("no tests/golden/farneback_cv2_*.npz: run tools/pin_with_cv2.py where `import cv2` works (PARITY UNPINNED until then)")
    for path in files:
        got = cv2_fixture_grey(path)
        assert got is not None, f"{path}: the BGR frame does not regenerate here"
        bgr, outs = got
        for w, h, grey in outs:
            np.testing.assert_array_equal(frames_ref.bgr_to_grey(bgr, (w, h)), grey, err_msg=f"{path} {w}x{h}")
