"""tools/pin_with_cv2.py and the tests that consume its file, exercised end to end WITHOUT OpenCV: a stand-in `cv2`
module whose calcOpticalFlowFarneback / resize / cvtColor are the CPU oracle's is put in sys.modules, the tool writes
its file into a temporary directory, and the consumer checks run on it.  This pins nothing (the stand-in IS the
oracle): it shows that the day someone runs the tool beside a real OpenCV, the file it writes is read, its inputs
regenerate, and the comparisons run.  CPU only."""
import importlib.util
import os
import sys
import types

import numpy as np
import pytest

from oracle import farneback as F
from oracle import frames_ref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _stand_in_cv2():
    m = types.ModuleType("cv2")
    m.__version__ = "0.0.0-oracle-stand-in"
    m.OPTFLOW_USE_INITIAL_FLOW, m.OPTFLOW_FARNEBACK_GAUSSIAN = 4, 256
    m.INTER_NEAREST, m.COLOR_BGR2GRAY = 0, 6

    def calc(prev, nxt, flow, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags):
        return F.calc(prev, nxt, pyr_scale, levels, winsize, iterations, poly_n, poly_sigma, flags, flow=flow)

    def resize(img, dsize, interpolation):
        assert interpolation == m.INTER_NEAREST
        w, h = dsize
        sh, sw = img.shape[:2]
        ix = np.minimum(np.floor(np.arange(w) * (1.0 / (w / sw))).astype(np.int64), sw - 1)
        iy = np.minimum(np.floor(np.arange(h) * (1.0 / (h / sh))).astype(np.int64), sh - 1)
        return np.ascontiguousarray(img[iy][:, ix])

    m.calcOpticalFlowFarneback = calc
    m.resize = resize
    m.cvtColor = lambda img, code: frames_ref.bgr_to_grey(img)
    m.getBuildInformation = lambda: "  Version control: stand-in\n  CPU/HW features:\n    Baseline: none\n"
    m.getNumThreads = lambda: 1
    return m


@pytest.fixture()
def pin_file(tmp_path, monkeypatch):
    monkeypatch.setitem(sys.modules, "cv2", _stand_in_cv2())
    spec = importlib.util.spec_from_file_location("pin_with_cv2", os.path.join(ROOT, "tools", "pin_with_cv2.py"))
    tool = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(tool)
    return tool, tool.main([str(tmp_path)])


def test_tool_stands_alone():
    """numpy, cv2 and the standard library: nothing of this repository."""
    import ast
    src = open(os.path.join(ROOT, "tools", "pin_with_cv2.py")).read()
    names = set()
    for node in ast.walk(ast.parse(src)):
        if isinstance(node, ast.Import):
            names |= {a.name.split(".")[0] for a in node.names}
        elif isinstance(node, ast.ImportFrom):
            names.add((node.module or "").split(".")[0])
    assert names <= {"json", "os", "sys", "zlib", "numpy", "cv2"}, names


def test_tool_regenerates_the_suites_inputs(pin_file):
    from tests.helpers import FB_CASES, synth_pair
    tool, _ = pin_file
    assert [(s, k) for s, k in tool.CASES] == [(s, k) for s, k in FB_CASES]
    for (h, w), _ in FB_CASES[:3]:
        for x, y in zip(tool.synth_pair(h, w, seed=70), synth_pair(h, w, seed=70)):
            np.testing.assert_array_equal(x, y)


def test_fixture_round_trip_through_the_consumers(pin_file):
    from tests.helpers import cv2_fixture_cases, cv2_fixture_files, cv2_fixture_grey
    from tests.test_oracle_farneback import check_oracle_against_cv2_fixture
    tool, path = pin_file
    assert cv2_fixture_files(os.path.dirname(path)) == [path]
    meta, cases, skipped = cv2_fixture_cases(path)
    assert not skipped and len(cases) == len(tool.CASES) + 3 and meta["format"] == tool.FORMAT
    assert sorted(c["flags"] for c, *_ in cases if c["h"] == 270) == [0, 4, 256, 260]
    assert all((init is not None) == bool(c["flags"] & 4) for c, _, _, init, _ in cases)
    n, same, worst = check_oracle_against_cv2_fixture(path)
    assert (n, same, worst) == (len(cases), len(cases), 0.0)       # the stand-in is the oracle
    bgr, outs = cv2_fixture_grey(path)
    assert len(outs) == len(tool.GREY_SIZES)
    for w, h, grey in outs:
        np.testing.assert_array_equal(frames_ref.bgr_to_grey(bgr, (w, h)), grey)


def test_a_fixture_whose_inputs_do_not_regenerate_is_skipped_not_failed(pin_file, tmp_path):
    import json
    from tests.helpers import cv2_fixture_cases
    _, path = pin_file
    z = dict(np.load(path))
    meta = json.loads(str(z["meta_json"]))
    meta["cases"][0]["crc_prev"] ^= 1
    z["meta_json"] = np.array(json.dumps(meta))
    other = str(tmp_path / "farneback_cv2_tampered.npz")
    np.savez_compressed(other, **z)
    _, cases, skipped = cv2_fixture_cases(other)
    assert skipped == [meta["cases"][0]["key"]] and len(cases) == len(meta["cases"]) - 1
