"""Oracle (and HIP path) against REAL OpenCV - wherever OpenCV exists.

Two sources, either is enough: a live `import cv2`, or the fixture tests/golden/opencv_pin.npz written by
tools/gen_golden_cv2.py on a machine that has OpenCV.  The build container has neither (no cv2, no
libopencv), so these tests are collected and skipped there; they are the hook that turns "parity unpinned" into
"pinned" the first time the repository is checked out next to an OpenCV install:

    python tools/gen_golden_cv2.py && python -m pytest tests/test_opencv_pin.py
"""
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURE = os.path.join(HERE, "golden", "opencv_pin.npz")


def _golden():
    if os.path.exists(FIXTURE):
        return dict(np.load(FIXTURE, allow_pickle=False))
    cv2 = pytest.importorskip("cv2", reason="no OpenCV here and no tests/golden/opencv_pin.npz (tools/gen_golden_cv2.py)")
    import importlib.util
    spec = importlib.util.spec_from_file_location("gen_golden_cv2", os.path.join(HERE, "..", "tools", "gen_golden_cv2.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    mod.main()
    return dict(np.load(FIXTURE, allow_pickle=False))


def _exact(gold):
    """3.2-era arithmetic is what the oracle restates; later OpenCV changed GaussianBlur / RANSAC internals."""
    v = str(gold["cv_version"])
    return v.startswith(("2.4", "3.0", "3.1", "3.2", "3.3"))


def test_pyramid_against_cv_resize(oracle, synth):
    gold = _golden()
    for t in (0, 5):
        img = synth.frame(t)
        for lv in range(1, 8):
            got = oracle.orb_level(img, lv).astype(int)
            want = gold[f"f{t}_level{lv}"].astype(int)
            assert got.shape == want.shape
            assert np.abs(got - want).max() <= (0 if _exact(gold) else 1), (t, lv)


def test_blur_against_cv_gaussianblur(oracle, synth):
    gold = _golden()
    for t in (0, 5):
        img = synth.frame(t)
        for lv in range(8):
            got = oracle.orb_level(img, lv, blurred=True).astype(int)
            want = gold[f"f{t}_blur{lv}"].astype(int)
            assert np.abs(got - want).max() <= (0 if _exact(gold) else 2), (t, lv)


def test_fast_cells_against_cv_fast(oracle, synth):
    """Key points of frame 0 before retainBest: same (x, y, response) set per level as cv::FAST run cell by cell."""
    gold = _golden()
    img = synth.frame(0)
    for lv in range(8):
        S = oracle.orb_score(img, lv).astype(int)
        want = gold[f"f0_fast{lv}"]
        assert len(want) > 0
        for x, y, resp, _, _ in want[:: max(1, len(want) // 300)]:
            assert S[int(y), int(x)] - 1 == int(resp), (lv, x, y)      # cornerScore = S - 1


def test_fast_atan2_against_cv(oracle):
    gold = _golden()
    l = oracle.lib()
    got = np.array([l.orb_ref_fast_atan2(float(y), float(x)) for x, y in gold["atan2_xy"]], np.float32)
    assert np.array_equal(got, gold["atan2_deg"])


@pytest.mark.parametrize("n", [300, 700, 1000])
def test_fundamental_mask_against_cv(oracle, n):
    gold = _golden()
    mask, ninl = oracle.fundamental_mask(gold[f"fm{n}_p1"], gold[f"fm{n}_p2"])
    want = gold[f"fm{n}_mask"]
    if _exact(gold):
        assert np.array_equal(mask, want)
    else:   # 3.4+/4.x reject degenerate subsets inside getSubset: the sample sequence differs, the consensus set barely
        assert (mask != want).mean() < 0.03


@pytest.mark.gpu
def test_hip_extractor_against_cv_fixture(synth):
    """The HIP pyramid / blur planes against real OpenCV's (through the debug plane download of the extractor)."""
    gold = _golden()
    from se2lam_amd import orb
    ex = orb.ORBextractor()
    img = synth.frame(0)
    ex(img)
    for lv in range(1, 8):
        got = ex.debug_level(0, lv).astype(int)
        assert np.abs(got - gold[f"f0_level{lv}"].astype(int)).max() <= (0 if _exact(gold) else 1)
        gotb = ex.debug_level(0, lv, blurred=True).astype(int)
        assert np.abs(gotb - gold[f"f0_blur{lv}"].astype(int)).max() <= (0 if _exact(gold) else 2)
