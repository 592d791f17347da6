"""CPU tests pinning the ORB oracle (oracle/orb_ref.cpp) to what the reference tree itself fixes.

The reference has no golden vectors; parity with real OpenCV is unpinned.  Pinned here:
the 256x4 pattern table (sha256 of the reference's bit_pattern_31_), umax, per-level quotas, level
sizes and cell grids derived from ORBextractor.cpp:463-556, EDGE_THRESHOLD/PATCH_SIZE, the Gaussian
taps, fastAtan2 accuracy, round-half-even, the FAST score definition against a brute-force
restatement of cv::FAST's segment test, and structural properties of the extractor output.
"""
import hashlib
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_pattern_table_hash(oracle):
    pat = oracle.orb_pattern()
    want = open(os.path.join(HERE, "golden", "orb_pattern_31.sha256")).read().strip()
    assert hashlib.sha256(",".join(map(str, pat.tolist())).encode()).hexdigest() == want
    assert pat[:8].tolist() == [8, -3, 9, 5, 4, 2, 7, -12]          # ORBextractor.cpp:205-206
    assert pat[-4:].tolist() == [-1, -6, 0, -11]                    # ORBextractor.cpp:460
    assert np.abs(pat).max() == 13
    # the HIP library carries the same table
    txt = open(os.path.join(HERE, "..", "se2lam_amd", "csrc", "orb_pattern_31.inc")).read()
    vals = [int(v) for line in txt.splitlines() if not line.startswith("//") for v in line.replace(",", " ").split()]
    assert vals == pat.tolist()


def test_constructor_tables(oracle):
    t = oracle.orb_tables()
    assert t["umax"].tolist() == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert t["quota"].tolist() == [217, 181, 151, 126, 105, 87, 73, 60] and t["quota"].sum() == 1000
    assert t["scale"][1] == np.float32(1.2) and t["scale"][0] == 1.0
    g = oracle.orb_geometry(480, 640)
    want = [(640, 480, 5, 6, 122, 75, 8), (533, 400, 5, 6, 101, 62, 7), (444, 333, 4, 5, 103, 61, 8),
            (370, 278, 4, 5, 85, 50, 7), (309, 231, 3, 4, 93, 50, 9), (257, 193, 3, 4, 75, 41, 8),
            (214, 161, 3, 4, 61, 33, 7), (179, 134, 3, 4, 49, 26, 5)]
    assert [tuple(r) for r in g.tolist()] == want
    assert sum(w * h for w, h, *_ in want) == 950532
    assert oracle.orb_gaussian_taps().tolist() == [18, 34, 49, 55, 49, 34, 18]


def test_fast_atan2_and_rounding(oracle):
    l = oracle.lib()
    rng = np.random.default_rng(0)
    xy = rng.normal(size=(2000, 2)).astype(np.float32) * 1000
    got = np.array([l.orb_ref_fast_atan2(float(y), float(x)) for x, y in xy])
    ref = np.degrees(np.arctan2(xy[:, 1].astype(np.float64), xy[:, 0].astype(np.float64))) % 360.0
    err = np.abs((got - ref + 180) % 360 - 180)
    assert err.max() < 0.3                         # cv::fastAtan2 accuracy ~0.3 deg
    assert l.orb_ref_fast_atan2(0.0, 1.0) == 0.0 and abs(l.orb_ref_fast_atan2(1.0, 0.0) - 90.0) < 1e-4
    assert [l.orb_ref_cv_round(v) for v in (0.5, 1.5, 2.5, -0.5, -1.5, 2.4999, 2.5001)] == [0, 2, 2, 0, -2, 2, 3]


def _segment_test(patch, thr):
    """cv::FAST 9/16 corner test, brute force: >= 9 contiguous circle pixels all > v+thr or all < v-thr."""
    ox = [0, 1, 2, 3, 3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1]
    oy = [3, 3, 2, 1, 0, -1, -2, -3, -3, -3, -2, -1, 0, 1, 2, 3]
    v = int(patch[3, 3])
    ring = [int(patch[3 + oy[k], 3 + ox[k]]) for k in range(16)]
    for sign in (1, -1):
        flags = [(sign * (p - v)) > thr for p in ring]
        run = 0
        for f in flags + flags:
            run = run + 1 if f else 0
            if run >= 9:
                return True
    return False


def test_fast_score_definition(oracle):
    """S > t  <=>  the pixel passes cv::FAST's segment test at threshold t  (S = cornerScore + 1)."""
    l = oracle.lib()
    rng = np.random.default_rng(3)
    for trial in range(300):
        if trial % 3 == 0:
            patch = rng.integers(0, 256, (7, 7)).astype(np.uint8)
        else:  # structured: bright/dark wedge so that real corners occur
            patch = np.full((7, 7), rng.integers(60, 200), np.uint8)
            ang = rng.uniform(0, 2 * np.pi); width = rng.uniform(0.5, 2.5)
            yy, xx = np.mgrid[-3:4, -3:4]
            m = np.abs(((np.arctan2(yy, xx) - ang + np.pi) % (2 * np.pi)) - np.pi) < width
            patch[m] = np.clip(int(patch[0, 0]) + rng.integers(-120, 120), 0, 255)
            patch = np.clip(patch.astype(int) + rng.integers(-4, 5, (7, 7)), 0, 255).astype(np.uint8)
        flat = np.ascontiguousarray(patch)
        S = l.orb_ref_fast_score(flat.ctypes.data + 3 * 7 + 3, 7)
        for t in (7, 20, 45):
            assert (S > t) == _segment_test(patch, t), (trial, t, S)


def test_extract_structure(oracle, synth):
    img = synth.frame(0)
    k, d = oracle.orb_extract(img)
    assert len(k) == 1000 and d.shape == (1000, 32)
    assert (np.diff(k["octave"]) >= 0).all()                                  # levels in order 0..7
    assert np.bincount(k["octave"], minlength=8).tolist() == [217, 181, 151, 126, 105, 87, 73, 60]
    t = oracle.orb_tables()
    geo = oracle.orb_geometry(480, 640)
    for lv in range(8):
        m = k["octave"] == lv
        x = k["x"][m] / t["scale"][lv]; y = k["y"][m] / t["scale"][lv]
        assert x.min() >= 16 - 1e-3 and x.max() < geo[lv, 0] - 16 + 1e-3      # EDGE_THRESHOLD margin
        assert y.min() >= 16 - 1e-3 and y.max() < geo[lv, 1] - 16 + 1e-3
        assert (k["size"][m] == float(int(31 * t["scale"][lv]))).all()
    assert ((k["angle"] >= 0) & (k["angle"] < 360)).all()
    assert (k["response"] >= 7).all() and (k["class_id"] == -1).all()
    # deterministic
    k2, d2 = oracle.orb_extract(img)
    assert np.array_equal(k, k2) and np.array_equal(d, d2)


def test_pyramid_and_blur_properties(oracle, synth):
    img = synth.frame(1)
    assert np.array_equal(oracle.orb_level(img, 0), img)
    flat = np.full((480, 640), 77, np.uint8)
    for lv in (1, 4, 7):
        assert (oracle.orb_level(flat, lv) == 77).all()                       # bilinear preserves constants
    # taps sum to 257: a constant c blurs to (c*257*257 + 2^15) >> 16, saturated
    for c in (0, 77, 200, 255):
        b = oracle.orb_level(np.full((480, 640), c, np.uint8), 0, blurred=True)
        assert (b == min(255, (c * 257 * 257 + (1 << 15)) >> 16)).all()
    l1 = oracle.orb_level(img, 1)
    assert l1.shape == (400, 533)


def test_flat_and_empty_images(oracle):
    k, d = oracle.orb_extract(np.full((480, 640), 128, np.uint8))
    assert len(k) == 0
    k, d = oracle.orb_extract(np.zeros((0, 0), np.uint8))
    assert len(k) == 0


def test_written_out_glibc_sincosf_is_this_machines_libm(oracle, tmp_path):
    """oracle/orb_ref.cpp, namespace glibc_flt32: glibc's sinf / cosf (2.28 and later) written out in double arithmetic - what the
    reference's `(float)cos(angle)` / `(float)sin(angle)` with a float angle compute when linked with glibc.  Held to THIS
    machine's libm on every float of [0, 2 pi] (1.09e9 arguments, a few seconds with OpenMP); skipped where libm is not glibc."""
    import ctypes
    import shutil
    import subprocess
    if shutil.which("gcc") is None:
        pytest.skip("no gcc")
    try:
        get_version = ctypes.CDLL("libc.so.6").gnu_get_libc_version
        get_version.restype = ctypes.c_char_p
        ver = tuple(int(v) for v in get_version().decode().split(".")[:2])
    except (OSError, AttributeError, ValueError):
        pytest.skip("libc is not glibc")
    if ver < (2, 28):
        pytest.skip("glibc before 2.28 has another sinf")
    oracle.lib()
    here = os.path.dirname(os.path.abspath(__file__))
    build = os.path.join(os.path.dirname(here), "oracle", "_build")
    exe = str(tmp_path / "chk")
    r = subprocess.run(["gcc", "-O2", "-fopenmp", "-fno-builtin", os.path.join(here, "c_glibc_sincosf_check.c"), "-o", exe, "-L", build, "-loracle",
                        "-Wl,-rpath," + build, "-lm"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([exe], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0 and "sinf != libm on 0, cosf on 0" in r.stdout, r.stdout + r.stderr
    # spot values through the Python wrapper: the argument the fuzz found, the small-angle branch, a quadrant boundary
    import numpy as np
    assert np.float32(oracle.glibc_sincosf(np.float32(0.509999156), False)).tobytes() == np.float32(float.fromhex("0x1.f3e48ap-2")).tobytes()
    assert oracle.glibc_sincosf(1e-5, False) == np.float32(1e-5) and oracle.glibc_sincosf(1e-5, True) == 1.0


def test_restated_nth_element_equals_libstdcxx(tmp_path):
    """oracle/stl_nth.h - libstdc++'s introselect written out, the order KeyPointsFilter::retainBest + resize keeps key points in
    (ORBextractor.cpp:692-694, 708-709) - against this machine's std::nth_element: the WHOLE permutation, on every input over a
    three-letter alphabet up to length 9, 60,000 random inputs from all-tied to tie-free, and median-of-three killers that run
    into the depth limit (heap-select branch)."""
    import shutil
    import subprocess
    if shutil.which("g++") is None:
        pytest.skip("no g++")
    here = os.path.dirname(os.path.abspath(__file__))
    exe = str(tmp_path / "cpp_stl_nth")
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-o", exe, os.path.join(here, "cpp_stl_nth.cpp")])
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and r.stdout.startswith("ok:"), r.stdout + r.stderr


def test_restatement_keeps_key_points_in_the_order_of_libstdcxx_nth_element(oracle, synth):
    """the restatement's two cuts go through stl_nth.h; its labelled alternative (the order of rounds 1-4) keeps the same number
    of points per level with the same responses, but not the same points in the same places"""
    k, d = oracle.orb_extract(synth.frame(3))
    oracle.orb_retain_stable(True)
    try:
        k4, d4 = oracle.orb_extract(synth.frame(3))
    finally:
        oracle.orb_retain_stable(False)
    assert len(k) == len(k4) == 1000 and not np.array_equal(k, k4)
    for lv in range(8):
        assert np.array_equal(np.sort(k["response"][k["octave"] == lv]), np.sort(k4["response"][k4["octave"] == lv]))
    # nth_element wrapper = std::nth_element on this machine
    rng = np.random.default_rng(2)
    for n in (4, 17, 240, 3000):
        e = (rng.integers(8, 70, n).astype(np.uint64) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
        for nth in (0, n // 2, n - 1):
            assert np.array_equal(oracle.nth_element(e, nth), oracle.nth_element(e, nth, std=True))
