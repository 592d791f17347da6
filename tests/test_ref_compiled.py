"""oracle/_ref: the REFERENCE'S OWN hot-path sources - /root/reference/src/ORBextractor.cpp, ORBmatcher.cpp, Frame.cpp,
Config.cpp, cvutil.cpp, compiled unmodified from where they lie against oracle/_shim (`make -C oracle ref`) - against the
CPU restatement (oracle/orb_ref.cpp, match_ref.cpp) and, with -m gpu, against the HIP path through the C ABI.

What this pins (VERDICT r03 "what's missing" #1): everything se2lam WROTE on the front-end path - the constructor's tables,
the pyramid loop, the cell grid with its quota redistribution and 20 / 7 threshold rule, the level loop, IC_Angle,
computeOrbDescriptor, HarrisResponses, the frame grid (PosInGrid's round()) and GetFeaturesInArea, the greedy passes of
MatchByWindow / MatchByProjection / SearchByBoW, ComputeThreeMaxima, DescriptorDistance, cvu::camprjc / se3map, the Se2
algebra - runs here as the reference compiled it.  What it does NOT pin: the OpenCV functions underneath (FAST, resize,
copyMakeBorder, GaussianBlur, fastAtan2) are oracle/_shim/cv_shim.cpp, a second, independently
written reading of OpenCV 3.2 - agreement between shim and restatement is two readings agreeing, not the library.

Order of the key points (round 5): the reference cuts every cell's and every level's list with KeyPointsFilter::retainBest +
resize, i.e. with std::nth_element - and the compiled reference calls THIS toolchain's libstdc++ for it (the stand-in's
retainBest is OpenCV's text: std::nth_element + std::partition).  Restatement (oracle/stl_nth.h) and HIP path
(introselect_wave) compute the same permutation, so extractor outputs are compared as the ARRAYS they are - no sorting on
either side: same key points in the same places, all seven cv::KeyPoint fields and the 32 descriptor bytes - and the
order-dependent matchers are run FROM IMAGES through both sides (test_hip_images_to_matches_equal_the_compiled_reference).

The library is built in this container (where /root/reference is) and travels to the GPU box prebuilt."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

from oracle import ref

pytestmark = pytest.mark.skipif(not ref.available(), reason="oracle/_ref is not built and /root/reference is not here")


def _same_features(a, b):
    """element for element, in the order the two sides emit them (no sorting)"""
    (ka, da), (kb, db) = a, b
    return len(ka) == len(kb) and np.array_equal(ka, kb) and np.array_equal(da, db)


def _prev(k):
    return np.ascontiguousarray(np.stack([k["x"], k["y"]], 1), np.float32)


@pytest.fixture(scope="module")
def ref_feats(synth):
    return [ref.orb_extract(synth.frame(t)) for t in range(10)]


# ------------------------------------------------------------------------------------------------------------------ CPU
def test_library_is_the_reference_compiled_where_it_lies():
    """The recipe compiles the reference's files by path; nothing of them is in this repository (no copy to drift)."""
    mk = open(os.path.join(os.path.dirname(ref.HERE), "oracle", "Makefile")).read()
    for f in ("src/ORBextractor.cpp", "src/ORBmatcher.cpp", "src/Frame.cpp", "src/Config.cpp", "src/cvutil.cpp"):
        assert "$(REF)/" + f in mk
    ref.lib()
    for root, _, files in os.walk(os.path.dirname(ref.HERE)):
        if any(s in root for s in ("/.git", "/gpurun_out", "/_ref", "/_build")):
            continue
        assert "bit_pattern_31_" not in files and "ORBextractor.cpp" not in files and "ORBmatcher.cpp" not in files, root


def test_extractor_config1_frames_equal_the_restatement(oracle, synth, ref_feats):
    """BASELINE configs[0] / [1]: the ten seeded 640x480 frames, 1000 features, 8 levels, FAST score."""
    for t in range(10):
        ko, do = oracle.orb_extract(synth.frame(t))
        kr, dr = ref_feats[t]
        assert len(kr) == 1000
        assert _same_features((ko, do), (kr, dr)), t
        assert np.all(np.diff(kr["octave"]) >= 0)


@pytest.mark.parametrize("nfeat,scale,levels,th,score", [(1000, 1.2, 8, 20, 0), (500, 1.2, 8, 20, 1), (300, 1.5, 4, 12, 1),
                                                         (2000, 1.1, 6, 30, 0), (1500, 1.2, 8, 7, 0)])
def test_extractor_other_parameters_and_harris_score(oracle, synth, nfeat, scale, levels, th, score):
    p = oracle.orb_params(nfeat, scale, levels, th, score)
    for t in (0, 3):
        assert _same_features(oracle.orb_extract(synth.frame(t), p, cap=8192), ref.orb_extract(synth.frame(t), p)), (t, score)


def test_extractor_odd_sizes_noise_and_flat_images(oracle, synth):
    rng = np.random.default_rng(11)
    base = synth.frame(2)
    cases = [base[:301, :433], base[37:400, 100:511], np.ascontiguousarray(base[::2, ::2]), base[:160, :200],
             rng.integers(0, 256, (240, 320)).astype(np.uint8),                               # noise: every cell full of corners
             np.full((200, 300), 90, np.uint8),                                                # flat: no key point at all
             np.concatenate([np.full((240, 160), 40, np.uint8), rng.integers(0, 256, (240, 160)).astype(np.uint8)], 1)]
    for i, img in enumerate(cases):
        img = np.ascontiguousarray(img)
        p = oracle.orb_params(400 + 150 * i, 1.2, 8 if min(img.shape) > 220 else 5)
        a, b = oracle.orb_extract(img, p, cap=8192), ref.orb_extract(img, p)
        assert _same_features(a, b), (i, img.shape, len(a[0]), len(b[0]))
    assert len(ref.orb_extract(cases[5])[0]) == 0


def test_descriptor_steering_goes_through_libm_float_trig_in_the_reference(oracle):
    """src/ORBextractor.cpp:166 writes `(float)cos(angle)` / `(float)sin(angle)` with a FLOAT angle under `using namespace std`:
    that resolves to the float overloads, i.e. libm's cosf / sinf, which glibc does not round correctly (sinf(0.509999156f) is
    one ulp below the rounded double sine); a steered sampling coordinate that lands exactly on a .5 tie then rounds to the
    other pixel.  Found by tools/fuzz_ref.py in round 4 (about one descriptor in two million, one or two bits).  Since round 5
    the restatement's default - and the HIP kernel k_angle_trig - is glibc's algorithm written out in double arithmetic
    (mode 2): equal to the compiled reference on such a frame, as is libm itself (mode 1); the rounded double trig (mode 0,
    rounds 1-4) differs in at most a few bits."""
    img = np.random.default_rng(151).integers(0, 256, (200, 240)).astype(np.uint8)
    p = oracle.orb_params(3000, 1.2, 4, 20, 0)
    kr, dr = ref.orb_extract(img, p)
    try:
        for mode in (2, 1):
            oracle.orb_trig_libm(mode)
            kc, dc = oracle.orb_extract(img, p, cap=16384)
            assert len(kr) == 3000 and np.array_equal(kc, kr) and np.array_equal(dc, dr), mode
        oracle.orb_trig_libm(0)
        ka, da = oracle.orb_extract(img, p, cap=16384)
    finally:
        oracle.orb_trig_libm(2)
    assert np.array_equal(ka, kr)
    assert (da != dr).any(1).sum() <= 1 and np.unpackbits(da ^ dr).sum() <= 2


def test_retain_best_of_rounds_1_to_4_differs_only_in_ties(oracle, synth):
    """SE2_REF_RETAIN=stable swaps the stand-in's retainBest (OpenCV's text over this libstdc++'s nth_element, the default) for
    the order rounds 1-4 defined: the n best by response, ties by position in the list.  One of the outcomes the C++ standard
    allows, not the one a GCC build produces: the same number of key points per level and the same multiset of responses (only
    WHICH of several equal-response corners survive differs), few points differ - and the restatement's labelled alternative
    (orb_retain_stable) reproduces it exactly."""
    code = ("import numpy as np, sys; sys.path.insert(0, %r); from oracle import ref; from se2lam_amd import synth;"
            "k, d = ref.orb_extract(synth.frame(1)); np.save(sys.argv[1], k)") % os.path.dirname(ref.HERE)
    out = os.path.join(os.environ.get("TMPDIR", "/tmp"), "ref_retain_stable.npy")
    subprocess.check_call([sys.executable, "-c", code, out], env=dict(os.environ, SE2_REF_RETAIN="stable"))
    ks = np.load(out)
    kc, _ = ref.orb_extract(synth.frame(1))
    assert len(ks) == len(kc)
    for lv in range(8):
        a, b = ks[ks["octave"] == lv], kc[kc["octave"] == lv]
        assert len(a) == len(b) and np.array_equal(np.sort(a["response"]), np.sort(b["response"])), lv
    key = lambda k: set(zip(k["octave"].tolist(), k["x"].tolist(), k["y"].tolist()))
    assert 0 < len(key(ks) ^ key(kc)) <= 0.1 * len(kc)
    oracle.orb_retain_stable(True)
    try:
        ko, _ = oracle.orb_extract(synth.frame(1))
    finally:
        oracle.orb_retain_stable(False)
    assert key(ko) == key(ks)


def test_images_to_matches_equal_the_compiled_reference(oracle, synth, ref_feats):
    """VERDICT r04 next #1, CPU half: extract(frame t), extract(frame t + 1), MatchByWindow with the first frame's key points as
    vbPrevMatched (Track.cpp:194) - each side on ITS OWN extraction, nothing sorted: key points, descriptors and vnMatches12 of
    the restatement equal the compiled reference's on the ten config-1 frames (round 4: 716 against 717 matches on pair (0, 1))."""
    feats = [oracle.orb_extract(synth.frame(t)) for t in range(10)]
    for t in range(9):
        (k1, d1), (k2, d2) = feats[t], feats[t + 1]
        (r1, e1), (r2, e2) = ref_feats[t], ref_feats[t + 1]
        assert np.array_equal(k1, r1) and np.array_equal(d1, e1), t
        m_o, n_o, p_o = oracle.match_window(k1, d1, k2, d2)
        m_r, n_r, p_r = ref.match_window(r1, e1, r2, e2)
        assert n_o == n_r and n_r > 500 and np.array_equal(m_o, m_r) and np.array_equal(p_o, p_r), (t, n_o, n_r)


def test_descriptor_distance_and_three_maxima(oracle):
    rng = np.random.default_rng(3)
    for _ in range(300):
        a = rng.integers(0, 256, 32).astype(np.uint8); b = rng.integers(0, 256, 32).astype(np.uint8)
        assert ref.hamming(a, b) == oracle.hamming(a, b) == int(np.unpackbits(a ^ b).sum())
    import ctypes as C
    from se2lam_amd import capi
    for trial in range(300):
        L = int(rng.integers(1, 31))
        h = rng.integers(0, 4 if trial % 3 else 60, L).astype(np.int32)
        if trial % 5 == 0:
            h[rng.integers(0, L)] = 500
        ind = [C.c_int(-1), C.c_int(-1), C.c_int(-1)]
        capi.check(capi.lib().se2gpu_three_maxima(h.ctypes.data, L, C.byref(ind[0]), C.byref(ind[1]), C.byref(ind[2])))
        assert tuple(i.value for i in ind) == ref.three_maxima(h), h      # the product's host function = the reference's member


def test_features_in_area_equal_the_restatement(oracle, ref_feats):
    rng = np.random.default_rng(4)
    k = ref_feats[0][0]
    for _ in range(300):
        x, y = float(rng.uniform(-30, 670)), float(rng.uniform(-30, 510))
        r = float(rng.choice([5, 20, 45, 120]))
        lo = int(rng.integers(-1, 7)); hi = lo if rng.random() < 0.3 else (lo + int(rng.integers(0, 4)) if lo >= 0 else -1)
        assert np.array_equal(ref.features_in_area(k, x, y, r, lo, hi), oracle.features_in_area(k, x, y, r, lo, hi)), (x, y, r, lo, hi)


@pytest.mark.parametrize("a,b", [(0, 1), (1, 2), (0, 5), (3, 3), (9, 2)])
def test_match_by_window_equals_the_restatement(oracle, ref_feats, a, b):
    (k1, d1), (k2, d2) = ref_feats[a], ref_feats[b]
    m_r, n_r, p_r = ref.match_window(k1, d1, k2, d2)
    m_o, n_o, p_o = oracle.match_window(k1, d1, k2, d2)
    assert n_r == n_o and np.array_equal(m_r, m_o) and np.array_equal(p_r, p_o)
    assert n_r > (100 if a != 9 else 0)
    for (win, lo, mn, mx, ratio) in ((8, 1, 0, 8, 0.9), (40, 2, 1, 5, 0.7), (20, 0, 0, 3, 0.6)):
        m_r, n_r, p_r = ref.match_window(k1, d1, k2, d2, None, win, lo, mn, mx, ratio)
        m_o, n_o, p_o = oracle.match_window(k1, d1, k2, d2, None, win, lo, mn, mx, ratio)
        assert n_r == n_o and np.array_equal(m_r, m_o) and np.array_equal(p_r, p_o), (win, lo, mn, mx, ratio)
    # duplicated queries: the eviction chain (vnMatches21) at work; chained vbPrevMatched
    kk = np.concatenate([k1[:200], k1[:200]]); dd = np.concatenate([d1[:200], d1[:200]])
    m_r, n_r, p_r = ref.match_window(kk, dd, k2, d2)
    m_o, n_o, p_o = oracle.match_window(kk, dd, k2, d2)
    assert n_r == n_o and np.array_equal(m_r, m_o) and np.array_equal(p_r, p_o)
    m_r2, n_r2, _ = ref.match_window(kk, dd, k1, d1, prev_xy=p_r)
    m_o2, n_o2, _ = oracle.match_window(kk, dd, k1, d1, prev_xy=p_o)
    assert n_r2 == n_o2 and np.array_equal(m_r2, m_o2)


def _projection_case(feats, seed, m=1500):
    rng = np.random.default_rng(seed)
    (k0, d0), (k1, d1) = feats[0], feats[1]
    fx = fy = 400.0; cx, cy = 320.0, 240.0
    src = rng.integers(0, len(k0), m)
    depth = rng.uniform(800, 6000, m).astype(np.float32)
    Xc = np.stack([(k0["x"][src] - cx) / fx * depth, (k0["y"][src] - cy) / fy * depth, depth], 1).astype(np.float32)
    th = 0.01
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], np.float32)
    t = np.array([15.0, -4.0, 8.0], np.float32)
    Tcw = np.concatenate([R, t[:, None]], 1).astype(np.float32)
    mp_pos = ((Xc - t) @ R).astype(np.float32)
    mp_desc = d0[src].copy()
    flip = rng.integers(0, 256, (m, 32)).astype(np.uint8) & ((rng.random((m, 32)) < 0.03).astype(np.uint8) * 255)
    mp_desc ^= flip.astype(np.uint8)
    mp_octave = k0["octave"][src].astype(np.int32)
    mp_skip = (rng.random(m) < 0.1).astype(np.uint8)
    kf_obs = (rng.random(len(k1)) < 0.2).astype(np.uint8)
    return mp_pos, mp_desc, mp_octave, mp_skip, Tcw, (fx, fy, cx, cy), k1, d1, kf_obs


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_match_by_projection_equals_the_restatement(oracle, ref_feats, seed):
    """cvu::se3map / camprjc, Frame::inImgBound, GetFeaturesInArea and the greedy pass, all the reference's own code."""
    args = _projection_case(ref_feats, seed)
    i_r, n_r = ref.match_projection(*args, 15, 2, 0.6)
    i_o, n_o = oracle.match_projection(*args, 15, 2, 0.6)
    assert n_r == n_o and n_r > 50 and np.array_equal(i_r, i_o)
    a = list(args); a[0] = args[0].copy(); a[0][::4, 2] -= 1e5       # some points behind the camera
    i_r, n_r = ref.match_projection(*a, 15, 2, 0.6)
    i_o, n_o = oracle.match_projection(*a, 15, 2, 0.6)
    assert n_r == n_o and np.array_equal(i_r, i_o)
    a = list(args)
    for k in range(4):
        a[k] = np.repeat(args[k], 3, axis=0)                          # chains of map points competing for one feature
    i_r, n_r = ref.match_projection(*a, 25, 1, 0.75)
    i_o, n_o = oracle.match_projection(*a, 25, 1, 0.75)
    assert n_r == n_o and np.array_equal(i_r, i_o)


def _feature_vector(desc, nbits):
    node = (desc[:, 0].astype(np.int32) | (desc[:, 1].astype(np.int32) << 8)) & ((1 << nbits) - 1)
    order = np.argsort(node, kind="stable")
    nodes, counts = np.unique(node[order], return_counts=True)
    ptr = np.concatenate([[0], np.cumsum(counts)]).astype(np.int32)
    return nodes.astype(np.int32), ptr, order.astype(np.int32)


@pytest.mark.parametrize("nbits,mp_only,ratio,ori", [(5, False, 0.6, True), (3, True, 0.6, True), (0, False, 0.9, False), (7, True, 0.75, True)])
def test_search_by_bow_equals_the_restatement(oracle, ref_feats, nbits, mp_only, ratio, ori):
    """DBoW2::FeatureVector is the reference's own class here (a std::map walked with lower_bound, ORBmatcher.cpp:128-276)."""
    (k1, d1), (k2, d2) = ref_feats[0], ref_feats[2]
    rng = np.random.default_rng(nbits)
    fv1, fv2 = _feature_vector(d1, nbits), _feature_vector(d2, nbits)
    h1 = (rng.random(len(k1)) < 0.7).astype(np.uint8); h2 = (rng.random(len(k2)) < 0.7).astype(np.uint8)
    m_r, n_r = ref.search_by_bow(k1, d1, fv1, h1, k2, d2, fv2, h2, mp_only, ratio, ori)
    m_o, n_o = oracle.search_by_bow(k1, d1, fv1, h1, k2, d2, fv2, h2, mp_only, ratio, ori)
    assert n_r == n_o and np.array_equal(m_r, m_o)


def test_frame_constructor_pipeline(oracle, synth):
    """Track's per-frame front end as the reference runs it: Frame::Frame (undistort with D = 0, extractor, grid assignment in
    the constructor itself) on two images, then MatchByWindow with the first frame's key points as vbPrevMatched."""
    (a, b, m12, nm, prev) = ref.track_two_frames(synth.frame(4), synth.frame(5))
    assert _same_features(a, oracle.orb_extract(synth.frame(4))) and _same_features(b, oracle.orb_extract(synth.frame(5)))
    m_o, n_o, p_o = oracle.match_window(a[0], a[1], b[0], b[1])        # the restatement on the reference's own arrays
    assert nm == n_o and nm > 300 and np.array_equal(m12, m_o) and np.array_equal(prev, p_o)


def test_se2_algebra_and_triangulation_helpers():
    """Se2::operator+ / - (src/Config.cpp:200-223) against the formulas the generator uses; cvu::triangulate recovers a point
    from two exact projections; cvu::checkParallax's thresholds."""
    from se2lam_amd import synth
    rng = np.random.default_rng(8)
    for _ in range(100):
        a = np.array([rng.uniform(-5e3, 5e3), rng.uniform(-5e3, 5e3), rng.uniform(-3.1, 3.1)], np.float32)
        b = np.array([rng.uniform(-500, 500), rng.uniform(-500, 500), rng.uniform(-3.1, 3.1)], np.float32)
        s = ref.se2_compose(a, b)
        c, sn = np.cos(np.float32(a[2])), np.sin(np.float32(a[2]))
        assert np.allclose(s[:2], [a[0] + b[0] * c - b[1] * sn, a[1] + b[0] * sn + b[1] * c], rtol=1e-5, atol=1e-2)
        assert abs(((s[2] - (a[2] + b[2]) + np.pi) % (2 * np.pi)) - np.pi) < 1e-5
        d = ref.se2_compose(s, a, minus=True)                         # (a + b) - a = b
        assert np.allclose(d[:2], b[:2], atol=2e-2) and abs(((d[2] - b[2] + np.pi) % (2 * np.pi)) - np.pi) < 1e-5
    K = np.array([[400, 0, 320], [0, 400, 240], [0, 0, 1]], np.float32)
    P1 = K @ np.eye(3, 4, dtype=np.float32)
    T2 = np.eye(4, dtype=np.float32); T2[0, 3] = -150.0
    P2 = K @ T2[:3]
    for _ in range(50):
        X = np.array([rng.uniform(-1500, 1500), rng.uniform(-800, 800), rng.uniform(1500, 6000), 1.0], np.float32)
        u1 = P1 @ X; u2 = P2 @ X
        got = ref.triangulate_point(u1[:2] / u1[2], u2[:2] / u2[2], P1, P2)
        assert np.allclose(got, X[:3], rtol=2e-2), (got, X)
    o1 = np.zeros(3, np.float32); o2 = np.array([150.0, 0, 0], np.float32)
    assert ref.check_parallax(o1, o2, np.array([75.0, 0, 500.0], np.float32), 2)          # 17 degrees
    assert not ref.check_parallax(o1, o2, np.array([75.0, 0, 50000.0], np.float32), 2)   # 0.17 degrees


def test_ba_edges_equal_the_compiled_reference(oracle, synth):
    """g2o::EdgeSE2XYZ::computeError / linearizeOplus (src/EdgeSE2XYZ.cpp:61-106, through SE2ToSE3, SE3Quat products and the
    camera map as the reference writes them) and g2o::PreEdgeSE2 (EdgeSE2XYZ.h:62-102) against the closed forms of
    oracle/ba_ref.cpp that the HIP kernels restate: residuals and both Jacobians at random states around config-3 edges."""
    g = synth.ba_graph(50, 5000)
    rng = np.random.default_rng(0)
    for _ in range(400):
        e = int(rng.integers(0, g.E))
        pose = g.poses[g.e_kf[e]] + rng.normal(0, [40, 40, 0.03])
        lw = g.lms[g.e_lm[e]] + rng.normal(0, 60, 3)
        a = oracle.ba_edge_se2xyz(g, pose, lw, g.e_uv[e])
        b = ref.edge_se2xyz(g, pose, lw, g.e_uv[e])
        for x, y in zip(a, b):
            assert np.abs(x - y).max() <= 1e-11 * max(1.0, np.abs(y).max())
    for _ in range(400):
        pi = np.array([rng.uniform(-8e3, 8e3), rng.uniform(-8e3, 8e3), rng.uniform(-3.1, 3.1)])
        pj = pi + rng.normal(0, [300, 300, 0.5])
        z = rng.normal(0, [300, 300, 0.5])
        a = oracle.ba_edge_pre_se2(pi, pj, z)
        b = ref.edge_pre_se2(pi, pj, z)
        for x, y in zip(a, b):
            assert np.abs(x - y).max() <= 1e-12 * max(1.0, np.abs(y).max())
    # PreEdgeSE2 has no angle wrap (EdgeSE2XYZ.h:80): the compiled reference shows it too
    e, _, _ = ref.edge_pre_se2([0, 0, 3.1], [0, 0, -3.1], [0, 0, 0.08])
    assert abs(e[2] - (-6.2 - 0.08)) < 1e-12
    # SE3ToSE2(SE2ToSE3(p)) = p, and d_inv_d_se2 is the derivative of the inverse (finite differences)
    for _ in range(50):
        p = np.array([rng.uniform(-5e3, 5e3), rng.uniform(-5e3, 5e3), rng.uniform(-3.0, 3.0)])
        back, dinv = ref.se2_se3_round_trip(p)
        assert np.allclose(back, p, rtol=0, atol=1e-9)

        def inv(q):
            c, s = np.cos(q[2]), np.sin(q[2])
            return np.array([-c * q[0] - s * q[1], s * q[0] - c * q[1], -q[2]])
        num = np.stack([(inv(p + h) - inv(p - h)) / 2e-6 for h in np.eye(3) * 1e-6], 1)
        assert np.allclose(dinv, num, rtol=1e-6, atol=1e-5)


def test_reduced_system_from_the_references_own_jacobians(oracle, synth):
    """The Schur-reduced pose system of a small window assembled in numpy from the COMPILED REFERENCE'S residuals and
    Jacobians (robust weights and constructQuadraticForm as g2o defines them: H += J' rho' Omega J, b -= J' rho' Omega e) equals
    the restatement's, which the HIP kernels are held to at 1e-11 (tests/test_ba_gpu.py)."""
    g = synth.ba_graph(8, 60)
    lam = 3.0
    P, L = g.P, g.L
    n = 3 * P
    Hpp = np.zeros((n, n)); bp = np.zeros(n); Hll = np.zeros((L, 3, 3)); bl = np.zeros((L, 3)); Hpl = {}
    for k in range(g.E):
        kf, lm = int(g.e_kf[k]), int(g.e_lm[k])
        e, Jp, Jl = ref.edge_se2xyz(g, g.poses[kf], g.lms[lm], g.e_uv[k])
        w = g.e_info[k]
        Om = np.array([[w[0], w[1]], [w[1], w[2]]])
        e2 = e @ Om @ e
        r1 = 1.0 if e2 <= g.huber ** 2 else g.huber / np.sqrt(e2)
        Hll[lm] += Jl.T @ (r1 * Om) @ Jl
        bl[lm] -= Jl.T @ (r1 * Om) @ e
        if not g.fixed[kf]:
            Hpp[3 * kf:3 * kf + 3, 3 * kf:3 * kf + 3] += Jp.T @ (r1 * Om) @ Jp
            bp[3 * kf:3 * kf + 3] -= Jp.T @ (r1 * Om) @ e
            Hpl[(kf, lm)] = Jp.T @ (r1 * Om) @ Jl
    for k in range(g.O):
        i, j = int(g.o_i[k]), int(g.o_j[k])
        e, A, B = ref.edge_pre_se2(g.poses[i], g.poses[j], g.o_meas[k])
        W = g.o_info[k].reshape(3, 3)
        for (a, Ja) in ((i, A), (j, B)):
            if g.fixed[a]:
                continue
            bp[3 * a:3 * a + 3] -= Ja.T @ W @ e
            for (b, Jb) in ((i, A), (j, B)):
                if not g.fixed[b]:
                    Hpp[3 * a:3 * a + 3, 3 * b:3 * b + 3] += Ja.T @ W @ Jb
    S = Hpp.copy(); bs = bp.copy()
    for p in range(P):
        if g.fixed[p]:
            S[3 * p:3 * p + 3, 3 * p:3 * p + 3] = np.eye(3)
        else:
            S[3 * p:3 * p + 3, 3 * p:3 * p + 3] += lam * np.eye(3)
    by_lm = {}
    for (kf, lm), blk in Hpl.items():
        by_lm.setdefault(lm, []).append((kf, blk))
    for lm, obs in by_lm.items():
        Dinv = np.linalg.inv(Hll[lm] + lam * np.eye(3))
        for (a, Ba) in obs:
            bs[3 * a:3 * a + 3] -= Ba @ Dinv @ bl[lm]
            for (b, Bb) in obs:
                S[3 * a:3 * a + 3, 3 * b:3 * b + 3] -= Ba @ Dinv @ Bb.T
    want = oracle.ba_reduced_system(g, lam)
    scale = np.abs(want["S"]).max()
    assert np.abs(S - want["S"]).max() <= 1e-10 * scale
    assert np.abs(bs - want["bs"]).max() <= 1e-10 * np.abs(want["bs"]).max()


# ---- src/optimizer.cpp and src/converter.cpp, compiled whole: the construction surface and the plane-motion priors
def _off_plane_poses(synth, n=6, seed=3):
    """camera poses Tcw of a body that is almost, but not exactly, on the plane (roll, pitch and height of a few mrad / mm)"""
    rng = np.random.default_rng(seed)
    out = []
    for _ in range(n):
        Tcw = synth.se2_to_Tcw(np.array([rng.uniform(-3000, 3000), rng.uniform(-3000, 3000), rng.uniform(-3.1, 3.1)]))
        d = synth.se3_exp_np(np.concatenate([rng.normal(0, 0.02, 3), rng.normal(0, 8.0, 3)]))
        out.append(d @ Tcw)
    return out


def _tbc32(synth, tilt=True):
    """an extrinsic that is exactly representable in float32 (Config::bTc is a CV_32F matrix), optionally not axis-aligned"""
    Tbc = np.eye(4)
    Tbc[:3, :3] = synth.RBC; Tbc[:3, 3] = synth.TBC
    if tilt:
        Tbc = Tbc @ synth.se3_exp_np(np.array([0.05, -0.11, 0.02, 3.0, -7.0, 11.0]))
        Tbc = Tbc.astype(np.float32).astype(np.float64)
    return Tbc


def _requat(T):
    """What SE3Quat(R, t) makes of a slightly non-orthonormal R (a float32-rounded rotation): Eigen's quaternion-from-matrix on
    the raw entries, then normalised.  The restatement is handed this rotation, the reference gets there by itself."""
    from scipy.spatial.transform import Rotation
    R = T[:3, :3]
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0); w = 0.5 * s; s = 0.5 / s
        x, y, z = (R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s
    else:
        i = int(np.argmax(np.diag(R))); j = (i + 1) % 3; k = (j + 1) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q = np.zeros(3); q[i] = 0.5 * s; s = 0.5 / s
        w = (R[k, j] - R[j, k]) * s; q[j] = (R[j, i] + R[i, j]) * s; q[k] = (R[k, i] + R[i, k]) * s
        x, y, z = q
    out = T.copy()
    out[:3, :3] = Rotation.from_quat(np.array([x, y, z, w])).as_matrix()
    return out


def test_window_built_with_the_references_add_calls_has_the_restatements_chi2(oracle, synth):
    """addCamPara / addVertexSE2 / addEdgeSE2 / addVertexSBAXYZ / addEdgeSE2XYZ of src/optimizer.cpp (compiled, unmodified)
    build the window; every edge's own computeError() and g2o's Huber rho give the cost the restatement starts from."""
    for (P, L, seed) in ((8, 200, 5), (20, 600, 11)):
        g = synth.ba_graph(P=P, L=L, seed=seed)
        total, chi_e, chi_o, (nedges, nfixed, nmarg) = ref.window_chi2(g)
        assert nedges == g.E + g.O and nfixed == int(g.fixed.sum()) and nmarg == g.L   # landmarks marginalised by default (optimizer.h:91)
        assert np.isclose(total, oracle.ba_chi2(g), rtol=1e-13, atol=0)
        for k in range(0, g.E, max(1, g.E // 50)):
            e, _, _ = oracle.ba_edge_se2xyz(g, g.poses[g.e_kf[k]], g.lms[g.e_lm[k]], g.e_uv[k])
            w = g.e_info[k]
            assert np.isclose(chi_e[k], e @ np.array([[w[0], w[1]], [w[1], w[2]]]) @ e, rtol=1e-11, atol=1e-18), k
        for k in range(g.O):
            e, _, _ = oracle.ba_edge_pre_se2(g.poses[g.o_i[k]], g.poses[g.o_j[k]], g.o_meas[k])
            assert np.isclose(chi_o[k], e @ g.o_info[k].reshape(3, 3) @ e, rtol=1e-10, atol=1e-18), k


def test_add_cam_para_takes_one_focal_length_from_K(oracle, synth):
    """addCamPara (optimizer.cpp:208-216) builds g2o::CameraParameters(K(0,0), (K(0,2), K(1,2)), 0): fy is never read."""
    g = synth.ba_graph(P=6, L=80, seed=2)
    K = np.array([[g.fx, 0, g.cx], [0, g.fx, g.cy], [0, 0, 1]])
    base = ref.window_chi2(g, K)[0]
    assert base == ref.window_chi2(g)[0]
    K[1, 1] = 123.0
    assert ref.window_chi2(g, K)[0] == base
    K[0, 0] = 401.0
    assert ref.window_chi2(g, K)[0] != base


def test_plane_motion_priors_of_the_compiled_reference_equal_the_restatement(oracle, synth):
    """addPlaneMotionSE3Expmap (optimizer.cpp:236-314) and addVertexSE3PlaneMotion (:336-470): measurement = pose with the
    body's roll, pitch and height removed, information = J' diag(...) J with J = adj(Tbc) resp. AdjTR(Tbc), made symmetric
    from the upper triangle; both vector orders."""
    for tilt in (False, True):
        Tbc = _tbc32(synth, tilt)
        Tq = _requat(Tbc)
        for T in _off_plane_poses(synth):
            for (xr, yr, z) in ((1e6, 1e6, 1.0), (2.5e5, 1e6, 4.0)):
                m, w, nedges = ref.plane_motion_prior(T, Tbc, xr, yr, z)
                mo, wo = oracle.plane_motion_prior(T, Tq, xr, yr, z)
                assert nedges == 1
                assert np.allclose(m, mo, rtol=0, atol=1e-9) and np.allclose(w, wo, rtol=1e-9, atol=1e-9 * np.abs(wo).max())
                assert np.array_equal(w, w.T)
                Twc = np.linalg.inv(T)
                m, w, para = ref.pg_plane_motion_prior(Twc, Tbc, xr, yr, z)
                mo, wo = oracle.pg_plane_motion_prior(Twc, Tq, xr, yr, z)
                assert para == 7        # the SE3-offset parameter id reaches the prior edge (optimizer.cpp:463)
                assert np.allclose(m, mo, rtol=0, atol=1e-9) and np.allclose(w, wo, rtol=1e-9, atol=1e-9 * np.abs(wo).max())


def test_expmap_prior_edge_and_information_order(oracle, synth):
    """EdgeSE3ExpmapPrior (optimizer.cpp:159-191): error = log(measurement * estimate^-1), Jacobian -I; addEdgeSE3Expmap
    (:482-500) swaps the (translation, rotation) blocks of its information argument - as se2lam_amd.optimizer.swapInfoBlocks and
    the C++ mirror (include/se2lam_amd/optimizer.h) do before the C ABI, which takes g2o's order."""
    from se2lam_amd.optimizer import swapInfoBlocks
    poses = _off_plane_poses(synth, 4, seed=8)
    for a, b in ((0, 1), (2, 3), (1, 1)):
        e, J = ref.prior_expmap_edge(poses[a], poses[b])
        want = oracle.se3_log(poses[a] @ np.linalg.inv(poses[b]))
        assert np.allclose(e, want, rtol=1e-9, atol=1e-9) and np.array_equal(J, -np.eye(6))
    rng = np.random.default_rng(4)
    A = rng.normal(size=(6, 6)); info = A @ A.T + np.diag([1, 2, 3, 4, 5, 6.0])
    got = ref.edge_se3expmap_info(info)
    assert got is not None and np.array_equal(got, swapInfoBlocks(info)) and np.array_equal(swapInfoBlocks(got), info)
    assert got[0, 0] == info[3, 3] and got[5, 0] == info[2, 3] and got[0, 5] == info[3, 2]
    bad = info.copy(); bad[4, 1] += 1e-3
    assert ref.edge_se3expmap_info(bad) is None          # verifyInfo (optimizer.cpp:573-581): symmetric to 1e-4


def test_so3_jacobians_and_converter_round_trips(synth):
    """Jl / invJl / invJJl (optimizer.cpp:64-157) are inverse to each other where the text says so; toSE3Quat(cv::Mat) ->
    toCvMat and toIsometry3D(cv::Mat) -> toCvMat (converter.cpp) return the float32 pose they were given."""
    rng = np.random.default_rng(1)
    for _ in range(6):
        v = rng.normal(0, 0.7, 3)
        a, b = ref.so3_jacobians(v)
        assert np.allclose(a @ b, np.eye(3), atol=1e-12)
        v6 = np.concatenate([v, rng.normal(0, 50.0, 3)])
        M = ref.inv_jjl(v6)
        assert np.allclose(M[:3, :3], b, atol=1e-12) and np.allclose(M[3:, 3:], b, atol=1e-12) and np.all(M[:3, 3:] == 0)
        # the lower-left block is -invJl Q invJl with Q linear in the translation part: doubling rho doubles it
        M2 = ref.inv_jjl(np.concatenate([v, 2 * v6[3:]]))
        assert np.allclose(M2[3:, :3], 2 * M[3:, :3], rtol=1e-10, atol=1e-12)
    for T in _off_plane_poses(synth, 3, seed=2):
        q, i = ref.converter_round_trip(T)
        T32 = T.astype(np.float32); T32[3] = (0, 0, 0, 1)
        assert np.allclose(q, T32, rtol=0, atol=2e-6 * max(1.0, np.abs(T32).max())) and np.allclose(i, T32, rtol=0, atol=2e-6 * max(1.0, np.abs(T32).max()))


# ---- src/sparsifier.cpp, compiled whole (the feature constraint between two key frames)
_SPARSIFY_CASES = ((12, 0, 400.0), (80, 1, 250.0), (200, 2, 800.0), (10, 4, 100.0), (150, 5, 600.0))
_SPARSIFY_HARD = (220, 196366, 484.40666147511246)     # found by tools/fuzz_gpu.py: cond(H_marginal) 5.6e15


def test_sparsifier_of_the_compiled_reference_equals_the_restatement(oracle, synth):
    """Sparsifier::DoMarginalizeSE3XYZ with JacobianSE3XYZ / HessianSE3XYZ / InfoSE3 / JacobianSE3 as the reference wrote
    them (forward differences of delta 1e-6 over g2o's minimal vector, the 1e-6 I regulariser, ldlt().solve, the SVD clamp):
    the relative pose is bit-identical; InfoSE3 on the SAME marginal Hessian agrees to round-off; the marginal Hessian
    assembled from the reference's per-measurement Hessians equals the restatement's; the end-to-end information agrees as
    far as the conditioning of the reference's construction lets two exact solvers agree (cond(H_marginal) ~ 1e15: the
    restatement inverts the 3x3 point blocks in closed form, the reference runs a dense LDL')."""
    for (N, seed, base) in _SPARSIFY_CASES + (_SPARSIFY_HARD,):
        kf, mp, m_kf, m_mp, m_info = synth.kf_pair(N, seed, base)
        z, info = ref.sparsify(kf, mp, m_kf, m_mp, m_info)
        zo, io, Hm = oracle.sparsify(kf, mp, m_kf, m_mp, m_info)
        assert np.array_equal(z, zo)
        assert np.abs(ref.sparsify_info_se3(kf, Hm) - io).max() <= 1e-12 * np.abs(io).max()
        H = np.zeros((12 + 3 * N, 12 + 3 * N))
        for k, m, W in zip(m_kf, m_mp, m_info):
            J, Hl = ref.sparsify_hessian(kf[k], mp[m], W)
            assert np.abs(Hl - J.T @ W @ J).max() <= 1e-12 * np.abs(Hl).max()
            idx = np.r_[6 * k + np.arange(6), 12 + 3 * m + np.arange(3)]
            H[np.ix_(idx, idx)] += Hl
        H[:12, :12] += 1e-6 * np.eye(12)
        Hm2 = H[:12, :12] - H[:12, 12:] @ np.linalg.solve(H[12:, 12:], H[12:, :12])
        assert np.abs(Hm2 - Hm).max() <= 1e-12 * np.abs(Hm).max()
        tol = 1e-4 if (N, seed, base) == _SPARSIFY_HARD else 1e-5      # (observed: 7e-8 ... 2e-6, the hard pair 1.9e-5)
        assert np.abs(info - io).max() <= tol * np.abs(io).max(), (N, seed)
        assert np.array_equal(info, info.T)
        lam = np.linalg.eigvalsh(info)
        assert lam.min() >= 1e-6 * (1 - 1e-9) and lam.max() <= 1e4 * (1 + 1e-9)


def test_sparsifier_ignores_a_third_key_frame_like_the_restatement(oracle, synth):
    """measurements whose idKF is neither 0 nor 1 are skipped (sparsifier.cpp:117-119) on both sides"""
    kf, mp, m_kf, m_mp, m_info = synth.kf_pair(12, 3)
    z0, i0 = ref.sparsify(kf, mp, m_kf, m_mp, m_info)
    z1, i1 = ref.sparsify(kf, np.r_[mp, [[1.0, 2.0, 3.0]]], np.r_[m_kf, 2], np.r_[m_mp, 12], np.concatenate([m_info, np.eye(3)[None]]))
    assert np.array_equal(z0, z1) and np.array_equal(i0, i1)


# ---- src/Map.cpp + src/KeyFrame.cpp + src/MapPoint.cpp, compiled with their own headers (libse2lam_ref_map.so)
def _random_ref_map(rng, K, M, reach):
    """tests/test_mapview.py's random map, filled into a se2lam::Map through the reference's own calls"""
    from test_mapview import _random_map
    kf_id, covisible, kf_obs, mp_id, mp_obs = _random_map(rng, K, M, reach)
    m = ref.RefMap(np.eye(3), np.eye(4), 2.0)
    for a in range(K):
        m.add_kf(int(kf_id[a]), a, [0.0, 0.0, 0.0], np.zeros((len(kf_obs[a]), 2)))
    for j in range(M):
        m.add_mp(int(mp_id[j]), [0.0, 0.0, 1000.0])
    for a in range(K):
        for f, j in enumerate(kf_obs[a]):
            m.observe(a, j, f)
    for a in range(K):
        for b in covisible[a]:
            m.covisible(a, b)
    return m, (kf_id, covisible, kf_obs, mp_id, mp_obs)


def test_update_local_graph_of_the_compiled_reference_equals_the_map_view():
    """Map::updateLocalGraph (src/Map.cpp:285-331) running on real KeyFrame / MapPoint objects - three covisibility hops
    from the current key frame, getAllObsMPs(false) of the local key frames, every other observer as a reference key frame,
    the three vectors ordered by mIdKF / mId - against se2gpu_map_update_local_graph on the CSR view of the same map."""
    from se2lam_amd.mapview import updateLocalGraph
    rng = np.random.default_rng(17)
    for K, M, reach in ((6, 40, 2), (40, 500, 3), (90, 1200, 4)):
        m, view = _random_ref_map(rng, K, M, reach)
        kf_id, mp_id = view[0], view[3]
        for cur in (0, K // 2, K - 1):
            lk_r, rk_r, lm_r = m.update_local_graph(cur)
            lk, rk, lm = updateLocalGraph(*view, cur, 3)
            assert kf_id[lk].tolist() == lk_r.tolist() and kf_id[rk].tolist() == rk_r.tolist() and mp_id[lm].tolist() == lm_r.tolist(), (K, cur)
            assert len(lk_r) >= 1 and not set(lk_r.tolist()) & set(rk_r.tolist())


def _reference_window(synth, n_ref):
    """A 12-key-frame window (tests/test_ba_oracle.py::_info_inputs) as a se2lam::Map: key frames with their key points,
    octaves and camera-frame points, map points, observations both ways, the odometry chain, a star of covisibility around
    key frame 0 that leaves the last n_ref key frames outside - they observe local map points and become reference key frames."""
    from test_ba_oracle import _info_inputs
    inp, g, level = _info_inputs(synth, 12, 200)
    K = np.array([[g.fx, 0, g.cx], [0, g.fx, g.cy], [0, 0, 1]], np.float32)
    bTc = np.eye(4); bTc[:3, :3] = g.Rbc; bTc[:3, 3] = g.tbc
    huber = np.float32(g.huber)
    m = ref.RefMap(K, bTc, huber)
    P, nL = g.P, g.P - n_ref
    twb = np.c_[inp["twb_xy"], g.poses[:, 2].astype(np.float32)].astype(np.float32)
    uv32 = g.e_uv.astype(np.float32)
    frame_id = np.arange(100, 100 + P); frame_id[4] = 50          # Frame::id: the smallest one is local key frame 4
    ftr = np.zeros(g.E, int); cnt = np.zeros(P, int)
    for k in range(g.E):
        ftr[k] = cnt[g.e_kf[k]]; cnt[g.e_kf[k]] += 1
    for a in range(P):
        sel = np.nonzero(g.e_kf == a)[0]
        m.add_kf(10 + a, int(frame_id[a]), twb[a], uv32[sel], level[sel], inp["lc"][sel])
    for l in range(g.L):
        m.add_mp(1000 + l, g.lms[l].astype(np.float32))
    for k in range(g.E):
        m.observe(int(g.e_kf[k]), int(g.e_lm[k]), int(ftr[k]))
    for a in range(1, nL):
        m.covisible(0, a); m.covisible(a, 0)
    cov = {}
    for k in range(g.O):
        i, j = int(g.o_i[k]), int(g.o_j[k])
        cov[i] = (j, g.o_meas[k], np.linalg.inv(g.o_info[k].reshape(3, 3)))
        m.set_odo(i, j, *cov[i][1:])
    return m, dict(g=g, inp=inp, level=level, K=K, huber=huber, twb=twb, uv32=uv32, frame_id=frame_id, nL=nL, odo=cov)


@pytest.mark.parametrize("n_ref", [0, 3])
def test_load_local_graph_of_the_compiled_reference(oracle, synth, n_ref):
    """Map::loadLocalGraph(SlamOptimizer&) (src/Map.cpp:891-1053), compiled, on a map built with the reference's own calls:
    the vertex numbering (local key frames, reference key frames, map points from nLocal + nRef + 1), the fixed rule
    (:900-913, 927, 969), cov^-1 of the pre-integrated odometry (:943-953), and per observation the information
    Sigma_all^-1 of :1024-1049 - against the restatement (oracle.ba_edge_information, which the device kernel
    k_edge_information is held to) and, for the whole graph, the restatement's robust cost."""
    m, w = _reference_window(synth, n_ref)
    g, inp, nL = w["g"], w["inp"], w["nL"]
    lk, rk, lm = m.update_local_graph(0)
    assert lk.tolist() == [10 + a for a in range(nL)] and rk.tolist() == [10 + a for a in range(nL, g.P)]
    local_mp = sorted(set(int(l) for k, l in zip(g.e_kf, g.e_lm) if k < nL))
    assert lm.tolist() == [1000 + l for l in local_mp]
    out = m.load_local_graph()
    P, maxKFid = g.P, g.P + 1
    # vertices
    assert out["v_id"].tolist() == list(range(P)) + [maxKFid + i for i in range(len(local_mp))]
    assert np.array_equal(out["v_est"][:P], w["twb"].astype(np.float64)) and not out["v_kind"][:P].any() and out["v_kind"][P:].all()
    assert np.array_equal(out["v_est"][P:], g.lms[local_mp].astype(np.float32).astype(np.float64))
    fixed = np.zeros(P, bool)
    if n_ref == 0:
        fixed[4] = True                                   # the key frame with the smallest Frame::id (:900-913)
    else:
        fixed[nL:] = True                                 # reference key frames (:969); no local one is fixed then
    assert np.array_equal(out["v_fixed"][:P], fixed) and not out["v_fixed"][P:].any()
    assert out["v_marginalized"][P:].all() and not out["v_marginalized"][:P].any()
    # odometry edges: local pairs only, information = cov^-1
    want_odo = [(i, j) for i, (j, _, _) in sorted(w["odo"].items()) if i < nL and j < nL]
    assert [tuple(x) for x in out["o_ids"].tolist()] == want_odo
    for (i, j), meas, info in zip(want_odo, out["o_meas"], out["o_info"]):
        assert np.array_equal(meas, w["odo"][i][1]) and np.allclose(info, np.linalg.inv(w["odo"][i][2]), rtol=1e-11, atol=0)
    # observation edges: every (key frame, local map point) pair once, measurement, Huber delta, information
    Rcw = np.stack([m.kf_pose(a)[:3, :3] for a in range(P)]).reshape(P, 9)         # the reference's own float Tcw = cTb * Twb^-1
    info = oracle.ba_edge_information(**dict(inp, Rcw=Rcw))
    pos_of = {1000 + l: i for i, l in enumerate(local_mp)}
    edge_of = {(int(k), maxKFid + pos_of[1000 + int(l)]): e for e, (k, l) in enumerate(zip(g.e_kf, g.e_lm)) if int(l) in set(local_mp)}
    assert len(out["e_ids"]) == len(edge_of) and len(set(map(tuple, out["e_ids"].tolist()))) == len(edge_of)
    for ids, uv, W, delta in zip(out["e_ids"].tolist(), out["e_uv"], out["e_info"], out["e_delta"]):
        e = edge_of[tuple(ids)]
        assert np.array_equal(uv, w["uv32"][e].astype(np.float64)) and delta == float(w["huber"])
        assert np.allclose(W, info[e], rtol=1e-10, atol=0), e
    # the whole graph's robust cost, through the restatement on the same numbers
    import dataclasses
    keep = np.array([int(l) in set(local_mp) for l in g.e_lm])
    remap = np.full(g.L, -1); remap[local_mp] = np.arange(len(local_mp))
    oi = np.array([i for i, _ in want_odo], np.int32); oj = np.array([j for _, j in want_odo], np.int32)
    g2 = dataclasses.replace(
        g, poses=w["twb"].astype(np.float64), fixed=fixed.astype(np.uint8), lms=g.lms[local_mp].astype(np.float32).astype(np.float64),
        e_kf=g.e_kf[keep].astype(np.int32), e_lm=remap[g.e_lm[keep]].astype(np.int32), e_uv=w["uv32"][keep].astype(np.float64),
        e_info=np.stack([info[keep][:, 0, 0], info[keep][:, 0, 1], info[keep][:, 1, 1]], axis=1), o_i=oi, o_j=oj,
        o_meas=np.stack([w["odo"][i][1] for i, _ in want_odo]), o_info=np.stack([np.linalg.inv(w["odo"][i][2]).reshape(-1) for i, _ in want_odo]),
        huber=float(w["huber"]), poses_true=None, lms_true=None)
    assert np.isclose(out["chi2"], oracle.ba_chi2(g2), rtol=1e-10, atol=0)


def _se3_window(synth, n_ref):
    """_reference_window plus the SE3 odometry constraints KeyFrame::mOdoMeasureFrom holds (4x4 / 6x6 CV_32F, the information
    in (translation, rotation) order), and the reference's SE3-expmap local graph as a synth.BA3Graph"""
    m, w = _reference_window(synth, n_ref)
    g, nL = w["g"], w["nL"]
    rng = np.random.default_rng(5 + n_ref)
    Tcw = np.stack([m.kf_pose(a) for a in range(g.P)]).astype(np.float64)
    for a in range(nL - 1):                                    # key frame a -> a + 1, as LocalMapper sets them
        noise = synth.se3_exp_np(rng.normal(0, 1.0, 6) * np.array([1e-3, 1e-3, 2e-3, 2.0, 2.0, 2.0]))
        meas = (noise @ Tcw[a + 1] @ np.linalg.inv(Tcw[a])).astype(np.float32)
        A = np.diag([0.2, 0.2, 0.2, 5e5, 5e5, 2e5]) + 0.02 * np.diag([0.45, 0.45, 0.45, 700, 700, 450]) @ rng.normal(0, 1, (6, 6))
        info = (0.5 * (A + A.T) + np.diag([0.02, 0.02, 0.02, 1e4, 1e4, 1e4])).astype(np.float32)
        info = 0.5 * (info + info.T)
        m.set_odo_se3(a, a + 1, meas, info)
    m.update_local_graph(0)
    out = m.load_local_graph_se3()
    P = g.P
    kf = out["v_kind"] == 2
    assert out["v_id"][kf].tolist() == list(range(P)) and kf[:P].all()
    local_mp = sorted(set(int(l) for k, l in zip(g.e_kf, g.e_lm) if k < nL))
    maxKFid = P + 1
    assert out["v_id"][~kf].tolist() == [maxKFid + i for i in range(len(local_mp))]
    poses = np.tile(np.eye(4), (P, 1, 1))
    poses[:, :3, :3] = out["v_est"][:P, :9].reshape(P, 3, 3); poses[:, :3, 3] = out["v_est"][:P, 9:]
    has_prior = np.zeros(P, np.uint8); has_prior[out["p_id"]] = 1
    prior_meas = np.tile(np.eye(4), (P, 1, 1)); prior_meas[out["p_id"]] = out["p_meas"]
    prior_info = np.zeros((P, 6, 6)); prior_info[out["p_id"]] = out["p_info"]
    g3 = synth.BA3Graph(poses=poses, fixed=out["v_fixed"][:P].astype(np.uint8), lms=out["v_est"][P:, :3].copy(),
                        e_kf=out["e_ids"][:, 1].astype(np.int32), e_lm=(out["e_ids"][:, 0] - maxKFid).astype(np.int32), e_uv=out["e_uv"].copy(),
                        e_w=out["e_info"][:, 0, 0].copy(), has_prior=has_prior, prior_meas=prior_meas, prior_info=prior_info,
                        o_i=out["o_ids"][:, 0].astype(np.int32), o_j=out["o_ids"][:, 1].astype(np.int32), o_meas=out["o_meas"], o_info=out["o_info"],
                        fx=float(w["K"][0, 0]), cx=float(w["K"][0, 2]), cy=float(w["K"][1, 2]), huber=float(w["huber"]))
    return m, w, out, g3


@pytest.mark.parametrize("n_ref", [0, 3])
def test_se3_local_graph_of_the_compiled_reference(oracle, synth, n_ref):
    """Map::loadLocalGraph(optimizer, vpEdgesAll, vnAllIdx) (src/Map.cpp:414-566), compiled: one VertexSE3Expmap and one
    plane-motion prior per local key frame (fixed by the mIdKF rule), reference key frames fixed without a prior, the odometry
    edges with their information re-ordered to g2o's (rotation, translation), one EdgeProjectXYZ2UV per observation - added to
    the optimizer twice by the reference, kept once as g2o's edge set does - with information invSigma2 I; and the cost of
    that graph, every edge evaluated by its own computeError, against the restatement of the SE3-expmap local BA."""
    from se2lam_amd.optimizer import swapInfoBlocks
    m, w, out, g3 = _se3_window(synth, n_ref)
    g, nL, P = w["g"], w["nL"], w["g"].P
    assert out["p_id"].tolist() == list(range(nL))                                  # a prior per local key frame, none for the reference ones
    fixed = np.zeros(P, bool)
    if n_ref == 0:
        fixed[0] = True                                                             # smallest mIdKF (:426-438)
    else:
        fixed[nL:] = True
    assert np.array_equal(out["v_fixed"][:P], fixed)
    assert [tuple(x) for x in out["o_ids"].tolist()] == [(a + 1, a) for a in range(nL - 1)]   # (mOdoMeasureFrom.first, this key frame)
    assert len(out["e_ids"]) == out["edges_returned"] == len(set(map(tuple, out["e_ids"].tolist())))
    assert (out["e_level"] == 0).all() and (out["e_delta"] == float(w["huber"])).all()
    assert np.array_equal(out["e_info"][:, 0, 1], np.zeros(len(out["e_ids"]))) and np.array_equal(out["e_info"][:, 0, 0], out["e_info"][:, 1, 1])
    sf = np.ones(8, np.float32)
    for i in range(1, 8):
        sf[i] = sf[i - 1] * np.float32(1.2)
    assert set(np.unique(out["e_info"][:, 0, 0])) <= set((np.float32(1.0) / (sf * sf)).astype(np.float64))
    # priors: what addPlaneMotionSE3Expmap makes of the key frame's own pose (held to the restatement in the test above)
    Tbc = np.eye(4); Tbc[:3, :3] = g.Rbc; Tbc[:3, 3] = g.tbc
    for a, meas, info in zip(out["p_id"], out["p_meas"], out["p_info"]):
        mo, wo = oracle.plane_motion_prior(g3.poses[a], _requat(Tbc))
        assert np.allclose(meas, mo, rtol=0, atol=1e-9) and np.allclose(info, wo, rtol=1e-9, atol=1e-9 * np.abs(wo).max())
    total, _ = oracle.ba3_chi2(g3)
    assert np.isclose(out["chi2"], total, rtol=1e-9, atol=0)
    # the odometry informations went through addEdgeSE3Expmap's block swap: swapping back gives the float matrices that went in
    for info in out["o_info"]:
        back = swapInfoBlocks(info)
        assert np.array_equal(back, back.astype(np.float32).astype(np.float64)) and back[0, 0] < 10 < back[3, 3]


# ---- the threads, compiled (src/Track.cpp, src/Localizer.cpp ... in libse2lam_ref_map.so)
def _triangulation_scene(n, seed):
    from test_triangulate import scene
    k1, k2, match, has_obs, P1, P2, Ocam, X = scene(n, seed)
    K = np.array([[400.0, 0, 320.0], [0, 400.0, 240.0], [0, 0, 1]], np.float32)
    th = 0.03
    Tcr = np.eye(4, dtype=np.float32)
    Tcr[:3, :3] = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], np.float32)
    Tcr[:3, 3] = np.array([-150.0, 5.0, 20.0], np.float32)
    return K, Tcr, k1, k2, match, has_obs, P1, P2, Ocam, X.astype(np.float32)


@pytest.mark.parametrize("seed,n", [(7, 600), (11, 1000), (3, 1)])
def test_do_triangulate_of_the_compiled_reference_equals_the_restatement(oracle, seed, n):
    """Track::doTriangulate (src/Track.cpp:373-415) on a Track object filled with the scene: which matches are triangulated
    (not the ones the key frame already observes: their position is mViewMPs[i], counted as tracked), the depth gate that
    drops a match, the 2-degree parallax flag, the counters - identical; positions to the accuracy two SVD routines share."""
    K, Tcr, k1, k2, match, has_obs, P1, P2, Ocam, X = _triangulation_scene(n, seed)
    pos, good, m, ng, nold = ref.track_triangulate(K, k1, k2, match, has_obs, X, Tcr, 500.0, 8000.0)
    po, go, mo, ngo, noldo = oracle.triangulate(k1, k2, match, has_obs, P1, P2, Ocam, 500.0, 8000.0, 2)
    assert (nold, ng) == (noldo, ngo) and np.array_equal(m, mo) and np.array_equal(good, go)
    seen = (match >= 0) & (has_obs == 1)
    assert np.array_equal(pos[seen], X[seen])                                     # Track.cpp:389-393
    acc = (match >= 0) & (has_obs == 0) & (m >= 0)
    if acc.any():
        assert np.abs(pos[acc] - po[acc]).max() <= 1e-5 * np.abs(po[acc]).max()
    untouched = ~(seen | acc)
    assert (pos[untouched] == -1).all() and not po[untouched].any()               # mLocalMPs keeps its initial (-1, -1, -1) there
    # a frame closer than nMinFrames = 8 to the key frame is not triangulated at all (:374-376)
    assert ref.track_triangulate(K, k1, k2, match, has_obs, X, Tcr, 500.0, 8000.0, frame_gap=7)[3:] == (0, 0)


def test_update_frame_pose_of_the_compiled_reference(synth):
    """Track::updateFramePose (src/Track.cpp:162-188): Trb / Tcr / Tcw / Twb of the frame from the odometry, and one step of the
    SE(2) pre-integration - measurement and covariance - against the generator's restatement of the same recursion
    (se2lam_amd.synth._preintegrate, which include/se2lam_amd/preintegration.h mirrors in C++).  The covariance array is
    written through Eigen::Map<Matrix3d, RowMajor>, i.e. in Matrix3d's own column-major order (the second template argument
    of Eigen::Map is an alignment option): symmetric, so either reading gives the same matrix."""
    rng = np.random.default_rng(9)
    bTc = np.eye(4); bTc[:3, :3] = synth.RBC; bTc[:3, 3] = synth.TBC
    noise = np.array([2.0, 2.0, 0.002], np.float32)
    kf_odom = np.array([1200.0, -300.0, 0.4], np.float32)
    kf_twb = np.array([5000.0, 800.0, 1.1], np.float32)
    meas = np.zeros(3); cov = np.zeros(9)
    last = kf_odom.copy()
    want_m = np.zeros(3); want_c = np.zeros((3, 3))
    Sv = np.diag((noise * noise).astype(np.float64))          # float products (Track.cpp:183-185)
    for step in range(12):
        odom = (last + np.array([rng.uniform(5, 40), rng.uniform(-4, 4), rng.uniform(-0.03, 0.03)])).astype(np.float32)
        out = ref.track_update_frame_pose(bTc, noise, kf_odom, kf_twb, last, odom, meas, cov)
        # Se2 odok = mFrame.odom - lastOdom (float arithmetic of Se2::operator-, src/Config.cpp:214-223)
        odok = ref.se2_compose(odom, last, minus=True).astype(np.float64)
        Phi = synth._rot2(want_m[2])
        A = np.eye(3); B = np.eye(3)
        A[:2, 2] = Phi @ np.array([-odok[1], odok[0]]); B[:2, :2] = Phi
        want_m[:2] += Phi @ odok[:2]; want_m[2] += odok[2]
        want_c = A @ want_c @ A.T + B @ Sv @ B.T
        assert np.allclose(out["meas"], want_m, rtol=1e-13, atol=1e-13)
        got_c = out["cov"].reshape(3, 3)
        assert np.allclose(got_c, want_c, rtol=1e-12, atol=1e-15) and np.allclose(got_c, got_c.T, rtol=1e-14, atol=1e-18)
        # the frame's poses: Trb = odom - kfodom, Twb = kfTwb + Trb, Tcr = cTb * (kfodom - odom) * bTc, Tcw = Tcr * kfTcw
        trb = ref.se2_compose(odom, kf_odom, minus=True)
        assert np.array_equal(out["Trb"], trb) and np.array_equal(out["Twb"], ref.se2_compose(kf_twb, trb))
        th = float(out["Twb"][2])
        Twb = np.eye(4); Twb[:2, :2] = synth._rot2(th); Twb[:2, 3] = out["Twb"][:2]
        assert np.allclose(out["Tcw"], np.linalg.inv(Twb @ bTc), rtol=0, atol=2e-3)     # float products of 4x4 matrices with mm translations
        meas, cov, last = out["meas"], out["cov"], odom


def test_cpp_preintegration_mirror_equals_the_compiled_update_frame_pose(tmp_path, synth):
    """include/se2lam_amd/preintegration.h run by tests/cpp_adapters_compile.cpp over its twelve odometry readings, against the
    reference's own compiled Track::updateFramePose on the same readings: measurement and covariance to 1e-12 (the numpy
    test of tests/test_capi.py holds them to 1e-5 only, float libm differences aside)."""
    import subprocess
    from test_capi import _build_adapter_binary
    exe = _build_adapter_binary(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True)
    line = [l for l in r.stdout.splitlines() if l.startswith("PRESE2")][0]
    v = np.array(line.split()[1:], dtype=np.float64)
    meas_c, cov_c = v[:3], v[3:12].reshape(3, 3)
    f32 = np.float32
    bTc = np.eye(4); bTc[:3, :3] = synth.RBC; bTc[:3, 3] = synth.TBC
    noise = np.array([2.0, 2.0, 0.002], f32)
    last = np.array([100, -20, 0.3], f32)
    meas, cov = np.zeros(3), np.zeros(9)
    for k in range(1, 13):
        now = np.array([f32(100) + f32(35) * f32(k) + f32(3) * f32(k % 3), f32(-20) + f32(4) * f32(k) - f32(2) * f32(k % 2),
                        f32(0.3) + f32(0.021) * f32(k)], f32)
        out = ref.track_update_frame_pose(bTc, noise, last, [0.0, 0.0, 0.0], last, now, meas, cov)
        meas, cov, last = out["meas"], out["cov"], now
    assert np.allclose(meas_c, meas, rtol=1e-12, atol=1e-12)
    assert np.allclose(cov_c, cov.reshape(3, 3), rtol=1e-12, atol=1e-18)


def _pose_only_case(seed, n):
    from test_pose_ba import _case, TBC, F, CX, CY, DELTA
    Tcw_true, Tcw0, Xw, uv, w, pose = _case(seed, n, 0.1)
    kps = np.zeros(n, KP_DTYPE_REF)
    lv = np.round(np.log(1.0 / w) / (2 * np.log(1.2))).astype(np.int32)
    kps["x"], kps["y"], kps["octave"] = uv[:, 0], uv[:, 1], lv
    return Tcw0.astype(np.float32), Xw.astype(np.float32), kps, TBC, F, CX, CY, DELTA


KP_DTYPE_REF = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


@pytest.mark.parametrize("seed,n", [(0, 400), (2, 37), (4, 6)])
def test_do_local_ba_of_the_compiled_reference(oracle, seed, n):
    """Localizer::DoLocalBA (src/Localizer.cpp:233-302) as compiled, stopped where it calls optimize(30): one free pose vertex
    with its plane-motion prior, every observed map point with good parallax as a FIXED, non-marginalised vertex with id
    mIdKF + mId, one EdgeProjectXYZ2UV each (invSigma2 I of the key point's octave, Huber TH_HUBER) - and the cost of that
    graph at the start against the restatement of the pose-only bundle adjustment."""
    Tcw0, Xw, kps, TBC, F, CX, CY, DELTA = _pose_only_case(seed, n)
    K = np.array([[F, 0, CX], [0, F, CY], [0, 0, 1]], np.float32)
    good = np.ones(n, np.uint8); good[::9] = 0                                  # points without parallax are skipped (:262)
    out = ref.localizer_do_local_ba(K, TBC, np.float32(DELTA), Tcw0, kps, Xw, good)
    ng = int(good.sum())
    assert (out["n_vertices"], out["n_fixed"], out["n_edges"], out["n_priors"], out["iterations"]) == (1 + ng, ng, ng, 1, 30)
    assert sorted(out["e_point"].tolist()) == np.nonzero(good)[0].tolist()
    sf = np.ones(8, np.float32)
    for i in range(1, 8):
        sf[i] = sf[i - 1] * np.float32(1.2)
    inv_sigma2 = (np.float32(1.0) / (sf * sf)).astype(np.float64)
    sel = out["e_point"]
    assert np.array_equal(out["e_w"], inv_sigma2[kps["octave"][sel]]) and (out["e_delta"] == float(np.float32(DELTA))).all()
    assert np.array_equal(out["e_uv"], np.stack([kps["x"][sel], kps["y"][sel]], axis=1).astype(np.float64))
    # the restatement on the same numbers: pose = toSE3Quat(float Tcw), prior from that pose
    T0 = _requat(Tcw0.astype(np.float64))
    meas, info = oracle.plane_motion_prior(T0, _requat(np.asarray(TBC, np.float64)))
    assert np.allclose(out["prior_meas"], meas, rtol=0, atol=1e-9) and np.allclose(out["prior_info"], info, rtol=1e-9, atol=1e-9 * np.abs(info).max())
    _, st = oracle.pose_only_ba(T0, meas, info, Xw[sel].astype(np.float64), out["e_uv"], out["e_w"], float(K[0, 0]), float(K[0, 2]), float(K[1, 2]),
                                float(np.float32(DELTA)), 30)
    assert np.isclose(out["chi2"], st["chi2_init"], rtol=1e-9, atol=0)


def _global_map(synth, P=14, seed=3):
    """All key frames of a small map for GlobalMapper::GlobalBA: mIdKF 0 .. P-1 on the generator's circle, each with the
    odometry constraint to its predecessor (mOdoMeasureFrom) and feature constraints to key frames ahead (mFtrMeasureFrom);
    measurements and informations as the CV_32F matrices the reference stores.  One map point per key frame, seen at key point 0."""
    rng = np.random.default_rng(seed)
    g = synth.ba_graph(P, 40, seed=seed)
    K = np.array([[g.fx, 0, g.cx], [0, g.fx, g.cy], [0, 0, 1]], np.float32)
    bTc = np.eye(4); bTc[:3, :3] = g.Rbc; bTc[:3, 3] = g.tbc
    m = ref.RefMap(K, bTc, np.float32(g.huber))
    twb = g.poses.astype(np.float32)
    lc = np.array([[10.0, -20.0, 2500.0]], np.float32)
    for a in range(P):
        m.add_kf(a, 100 + a, twb[a], np.array([[320.0, 240.0]]), np.array([1]), lc)
    Tcw = np.stack([m.kf_pose(a) for a in range(P)]).astype(np.float64)
    Twc = np.linalg.inv(Tcw)
    for a in range(P):                                       # a map point exactly where key frame a sees it: re-anchoring leaves it in place
        pw = (Twc[a] @ np.r_[lc[0].astype(np.float64), 1.0])[:3]
        m.add_mp(500 + a, pw.astype(np.float32))
        m.observe(a, a, 0)

    def constraint(i, j, sig_t, sig_r):
        Z = (np.linalg.inv(Twc[i]) @ Twc[j] @ synth.from_mqt_np(rng.normal(0, 1.0, 6) * np.array([sig_t] * 3 + [sig_r] * 3))).astype(np.float32)
        A = np.diag([1 / sig_t ** 2] * 3 + [1 / sig_r ** 2] * 3)
        B = rng.normal(0, 1, (6, 6)) * 0.05
        S = np.diag(np.sqrt(np.diag(A)))
        W = (A + S @ (B + B.T) @ S).astype(np.float32)
        return Z, 0.5 * (W + W.T)
    odo, ftr = [], []
    for a in range(1, P):
        Z, W = constraint(a, a - 1, 3.0, 1e-3)
        m.set_odo_se3(a, a - 1, Z, W); odo.append((a, a - 1, Z, W))
    for a in range(P):
        for d in (2, 3):
            if a + d < P and (a + d) % 2 == 0:
                Z, W = constraint(a, a + d, 8.0, 2e-3)
                m.add_ftr_measure(a, a + d, Z, W); ftr.append((a, a + d, Z, W))
    return m, dict(P=P, Twc=Twc, Tcw=Tcw, odo=odo, ftr=ftr, bTc=bTc, lc=lc)


def test_global_ba_graph_of_the_compiled_reference(oracle, synth):
    """GlobalMapper::GlobalBA (src/GlobalMapper.cpp:328-535) as compiled, on a map built through the reference's calls and
    stopped at optimize(GLOBAL_ITER): a VertexSE3 (T_w_c) per key frame, key frame 0 fixed, the plane-motion prior of
    addVertexSE3PlaneMotion on every one, EdgeSE3 odometry edges (this key frame -> mOdoMeasureFrom.first) and feature edges
    (mFtrMeasureFrom), informations as stored; the cost of that graph - EdgeSE3 / EdgeSE3Prior evaluated by g2o's definition
    in the stand-in - against the restatement of the pose-graph optimisation.  (The feature-edge rejection loop of
    :421-483 sits behind PRE_REJECT_FTR_OUTLIER, which nothing in the reference defines: it is not part of the compiled code.)
    Afterwards the reference writes the estimates back and re-anchors every map point on its main key frame (:497-531)."""
    m, w = _global_map(synth)
    P = w["P"]
    before = np.stack([m.mp_pos(a) for a in range(P)])
    out = m.global_ba(global_iter=15)
    assert out["iterations"] == 15 and out["v_id"].tolist() == list(range(P)) and out["v_fixed"].tolist() == [True] + [False] * (P - 1)
    assert out["p_id"].tolist() == list(range(P))
    # T_w_c = cvu::inv(Tcw) in float, through toIsometry3D's quaternion
    for a in range(P):
        assert np.allclose(out["v_est"][a], np.linalg.inv(w["Tcw"][a]), rtol=0, atol=2e-3)
    want = [(i, j) for i, j, _, _ in w["odo"]] + sorted((i, j) for i, j, _, _ in w["ftr"])
    got = [tuple(x) for x in out["e_ids"].tolist()]
    assert got[:P - 1] == want[:P - 1] and sorted(got[P - 1:]) == want[P - 1:]
    by_pair = {(i, j): (Z, W) for i, j, Z, W in w["odo"] + w["ftr"]}
    for (i, j), Z, W in zip(got, out["e_meas"], out["e_info"]):
        assert np.array_equal(W, by_pair[(i, j)][1].astype(np.float64))             # toMatrix6d: the float matrix as it is
        # toIsometry3D(cv::Mat) (converter.cpp:20-29): the rotation of the quaternion of the float matrix, NOT normalised - equal to the
        # float matrix to its own rounding
        assert np.allclose(Z, by_pair[(i, j)][0].astype(np.float64), rtol=0, atol=1e-6)
    # the same graph through the restatement
    Tq = _requat(w["bTc"].astype(np.float32).astype(np.float64))
    pm = [oracle.pg_plane_motion_prior(_requat(out["v_est"][a]), Tq) for a in range(P)]     # toSE3Quat(pose) normalises (optimizer.cpp:430)
    for a in range(P):
        assert np.allclose(out["p_meas"][a], pm[a][0], rtol=0, atol=1e-9) and np.allclose(out["p_info"][a], pm[a][1], rtol=1e-9, atol=1e-9 * np.abs(pm[a][1]).max())
    pg = synth.PoseGraph(poses=out["v_est"], fixed=out["v_fixed"].astype(np.uint8), has_prior=np.ones(P, np.uint8), prior_meas=out["p_meas"],
                         prior_info=out["p_info"], o_i=out["e_ids"][:, 0].astype(np.int32), o_j=out["e_ids"][:, 1].astype(np.int32),
                         o_meas=out["e_meas"], o_info=out["e_info"])
    total, chi = oracle.pg_chi2(pg)
    assert np.isclose(out["chi2"], total, rtol=1e-9, atol=0) and np.allclose(out["e_chi2"], chi, rtol=1e-8, atol=1e-12)
    # write-back: poses unchanged (nothing was optimised) up to float conversions; every map point sits where its main key frame sees it
    for a in range(P):
        assert np.allclose(m.kf_pose(a), w["Tcw"][a], rtol=0, atol=5e-3)
        assert np.allclose(m.mp_pos(a), before[a], rtol=0, atol=5e-3)


# ---- the vendored DBoW2 vocabulary, compiled from the reference's tree
@pytest.mark.parametrize("k,L,scoring,weighting,levelsup,seed", [(10, 4, 0, 0, 2, 0), (6, 6, 0, 0, 4, 1), (4, 3, 1, 1, 4, 2), (5, 4, 5, 0, 1, 3),
                                                                 (3, 5, 2, 2, 0, 4), (8, 3, 3, 3, 1, 5), (7, 4, 4, 0, 3, 6)])
def test_vocabulary_mirror_equals_the_compiled_dbow2(tmp_path, k, L, scoring, weighting, levelsup, seed):
    """include/se2lam_amd/ORBVocabulary.h (driven by tests/cpp_vocabulary.cpp) against se2lam::ORBVocabulary itself -
    DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>, vendored in the reference's tree and compiled from there: the same
    binary vocabulary file read by both (loadFromBinaryFile), transform(features, bow, fv, levelsup) of two key frames' 400
    descriptors each - words, values, feature vectors - and score(), for all six scoring and all four weighting types."""
    import subprocess
    import test_vocabulary as tv
    rng = np.random.default_rng(seed)
    parent, desc, weight, leaf = tv._random_vocabulary(rng, k, L, scoring, weighting)
    voc = tmp_path / "voc.bin"
    tv._write(voc, k, L, scoring, weighting, parent, desc, weight, leaf)
    sets = []
    for s_ in range(2):
        base = desc[rng.choice(np.nonzero(leaf)[0], 400)]
        noise = np.packbits(rng.random((400, 32, 8)) < 0.06, axis=2).reshape(400, 32)
        f = base ^ noise
        f[:40] = sets[0][:40] if s_ else f[:40]
        sets.append(f)
        (tmp_path / f"d{s_}.bin").write_bytes(f.tobytes())
    exe = tmp_path / "cpp_vocabulary"
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp_vocabulary.cpp"), "-o", str(exe)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe), str(voc), str(tmp_path / "d0.bin"), str(tmp_path / "d1.bin"), str(levelsup), str(tmp_path / "resaved.bin")],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stdout + r.stderr
    head, bows, fvs, scores = tv._parse(r.stdout)
    v = ref.RefVocabulary(voc)
    # the reference's loader runs its `while (!f.eof())` body once more after the last node (TemplatedVocabulary.h:1497-1517): the
    # failed read leaves the buffer as it was, so the last node is entered a second time - one phantom child of its parent, and one
    # phantom word when it is a leaf.  transform() takes the FIRST child of least distance, so the duplicate is never chosen and
    # neither BowVectors nor scores can see it; only size() counts it.  The mirror reads the file as written.
    phantom = 1 if leaf[-1] else 0
    assert v.loaded and [v.k, v.L, v.words, v.scoring, v.weighting] == [int(head[0]), int(head[1]), int(head[3]) + phantom, int(head[4]), int(head[5])]
    got = [v.transform(f, levelsup) for f in sets]
    for s_ in range(2):
        words, vals, fv = got[s_]
        assert bows[s_][0] == words and len(words) > 20
        assert np.allclose(bows[s_][1], vals, rtol=1e-15, atol=0)
        assert fvs[s_] == fv
    assert scores[0] == pytest.approx(v.score(got[0][:2], got[1][:2]), rel=1e-14, abs=1e-16)
    assert scores[1] == pytest.approx(v.score(got[0][:2], got[0][:2]), rel=1e-14, abs=1e-16)


def test_back_end_sweep_smoke(capfd):
    """A few rounds of every kind of tools/fuzz_ref_backend.py (the long runs are kept under profiles/): random maps, windows,
    priors, key-frame pairs, two-view scenes and pose-only problems through the compiled reference and the restatement."""
    import importlib.util
    spec = importlib.util.spec_from_file_location("fuzz_ref_backend", os.path.join(ROOT, "tools", "fuzz_ref_backend.py"))
    fz = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(fz)
    rng = np.random.default_rng(20260926)
    for _ in range(3):
        fz.fuzz_map(rng); fz.fuzz_graph(rng); fz.fuzz_tri(rng); fz.fuzz_poseba(rng)
        for _ in range(5):
            fz.fuzz_prior(rng)
        assert fz.fuzz_sparsify(rng) >= 0.0
    capfd.readouterr()          # (Localizer::DoLocalBA prints its timing to stderr)


# ---- src/MapStorage.cpp, compiled: the map file as a node structure (on-disk map)
_FS_DT = {"u": np.uint8, "c": np.int8, "w": np.uint16, "s": np.int16, "i": np.int32, "f": np.float32, "d": np.float64}


def _fs_events(docs):
    """A parsed OpenCV-YAML file as the node lines of oracle/_shim/cv_shim.hpp (shim_fs_dump): exact - reals by their bits."""
    from test_mapstorage import _Matrix
    out = []

    def val(v):
        if isinstance(v, _Matrix):
            a = np.array(v.data if v.data is not None else [], np.float64).astype(_FS_DT[v.dt])
            assert a.size == v.rows * v.cols
            out.append(f"M {v.rows} {v.cols} {v.dt} {a.tobytes().hex()}" if a.size else "M 0 0 u ")
        elif isinstance(v, bool):
            raise AssertionError("boolean in a map file")
        elif isinstance(v, int):
            out.append(f"I {v}")
        elif isinstance(v, float):
            out.append("R " + np.float64(v).tobytes()[::-1].hex())
        elif isinstance(v, str):
            out.append("S " + v)
        elif isinstance(v, list) or v is None:
            out.append("[")
            for x in v or []:
                val(x)
            out.append("]")
        elif isinstance(v, dict):
            out.append("{")
            for k, x in v.items():
                out.append("K " + k); val(x)
            out.append("}")
        else:
            raise AssertionError(type(v))
    for d in docs:
        out.append("D")
        for k, x in d.items():
            out.append("K " + k); val(x)
    return out


def _fs_canonical(lines):
    """Layout-free, order-free where the reference's own containers are: flow marks dropped; the entries of FtrGraphPairs (walked
    out of a std::map keyed by shared_ptr, i.e. in address order) sorted, duplicates of one (from, to) pair reduced to the first."""
    lines = [l[0] if l[:2] in ("[:", "{:") else l for l in lines if l]
    at = lines.index("K FtrGraphPairs")
    assert lines[at + 1] == "["
    entries, depth, cur, k = [], 0, [], at + 2
    while not (depth == 0 and lines[k] == "]"):
        cur.append(lines[k])
        depth += lines[k] in ("[", "{")
        depth -= lines[k] in ("]", "}")
        if depth == 0:
            entries.append(tuple(cur)); cur = []
        k += 1
    seen, keep = set(), []
    for e in entries:
        assert e[1] == "K PairId"
        pair = (e[3], e[4])
        if pair not in seen:
            seen.add(pair); keep.append(e)
    return lines[:at + 2] + [l for e in sorted(keep, key=lambda e: (int(e[3][2:]), int(e[4][2:]))) for l in e] + lines[k:], len(entries) - len(keep)


@pytest.fixture(scope="module")
def mapstorage_exe(tmp_path_factory):
    pytest.importorskip("yaml")
    out = tmp_path_factory.mktemp("msref") / "cpp_mapstorage"
    r = subprocess.run(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp_mapstorage.cpp"), "-o", str(out)],
                       capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    return str(out)


@pytest.mark.parametrize("seed,nkf,nmp", [(1, 6, 12), (2, 11, 30), (3, 23, 60)])
def test_map_file_through_the_compiled_map_storage(mapstorage_exe, tmp_path, capfd, seed, nkf, nmp):
    """The mirror (include/se2lam_amd/MapStorage.h) writes a map file; the REFERENCE'S MapStorage::loadMap - src/MapStorage.cpp
    compiled unmodified, over its own Map / KeyFrame / MapPoint - reads its node structure, and the reference's saveMap writes
    the map back: the same documents, keys in the same order, the same node types, every number and every matrix bit for bit.
    So the reference accepts what the mirror writes, and for that map it writes what the mirror wrote.  In between, the state
    loadMap builds in the reference's data model (observations on both sides, covisibility both ways, the odometry chain from /
    to, the feature constraints) is the one the mirror's loadMap reports.  What this does not cover is text: the YAML emitter /
    parser and the bitmaps are OpenCV's, restated in the mirror and held to PyYAML and an independent emitter in
    tests/test_mapstorage.py."""
    from test_mapstorage import _parse
    a, b = (str(tmp_path / n) + "/" for n in "ab")
    os.makedirs(a); os.makedirs(b)
    subprocess.run([mapstorage_exe, "gen", a, str(seed), str(nkf), str(nmp)], check=True, capture_output=True)
    docs, top = _parse(open(a + "se2lam.map").read())
    ev_mirror = _fs_events(docs)
    m = ref.RefMap(np.eye(3), np.eye(4), 2.0)
    nk = m.storage_load("\n".join(ev_mirror) + "\n")
    assert nk == len(top["KeyFrames"]) and m.storage_counts() == (nk, len(top["MapPoints"] or []))
    ev_ref = m.storage_save().split("\n")
    capfd.readouterr()
    want, dups = _fs_canonical(ev_mirror)
    got, none = _fs_canonical(ev_ref)
    assert none == 0 and got == want, next((i, g, w) for i, (g, w) in enumerate(zip(got + [None], want + [None])) if g != w)
    # ---- the reference's map after loadMap against the file and against what the mirror's loadMap reports
    nm = len(top["MapPoints"] or [])
    O = np.array(top["Observations"].data or [], np.int64).reshape(nk, nm)
    I = np.array(top["ObservationIndex"].data or [], np.int64).reshape(nk, nm)
    Cv = np.array(top["CovisibilityGraph"].data, np.int64).reshape(nk, nk)
    nxt = [e["NextId"] for e in top["OdoGraphNextKF"]]
    pairs = {tuple(e["PairId"]) for e in top["FtrGraphPairs"] or []}
    r = subprocess.run([mapstorage_exe, "copy", a, b], check=True, capture_output=True, text=True).stdout
    mirror = {int(l.split()[1]): l.split() for l in r.splitlines() if l.startswith("KF ")}
    for i in range(nk):
        st = m.storage_kf_state(i)
        kps = len(top["KeyFrames"][i]["KeyPoints"] or [])
        assert (st["kps"], st["kps_un"], st["desc_rows"], st["view_mps"], st["view_infos"]) == (kps,) * 5
        assert st["obs"] == int(O[i].sum()) and st["covisible"] == int(((Cv[i] + Cv[:, i]) > 0).sum())
        assert st["odo_from"] == nxt[i] and st["odo_to"] == (nxt.index(i) if i in nxt else -1)
        assert st["ftr_from"] == sum(1 for p in pairs if p[0] == i) and st["ftr_to"] == sum(1 for p in pairs if p[1] == i)
        mk = mirror[i]      # KF i kps n obs n covis n next d ftr n img RxC
        assert (int(mk[3]), int(mk[5]), int(mk[7]), int(mk[9])) == (st["kps"], st["obs"], st["covisible"], st["odo_from"])
        assert int(mk[11]) == st["ftr_from"]
    for j in range(nm):
        st = m.storage_mp_state(j)
        assert st == dict(obs=int(O[:, j].sum()), good_prl=True, null=False, id=j)
        for i in np.nonzero(O[:, j])[0]:
            assert m.storage_mp_ftr_idx(j, int(i)) == int(I[i, j])
    # ---- the key-frame trajectory text: OdoSLAM::saveMap (src/OdoSLAM.cpp:198-212, compiled) on the map the reference has just loaded
    # against the file the mirror's saveKeyFrameTrajectory wrote for the same map - byte for byte (stream formatting of the floats,
    # cvu::inv(bTc Tcw), the yaw through g2o's toEuler)
    bTc = np.eye(4, dtype=np.float32); bTc[:3, :3] = [[0, 0, 1], [-1, 0, 0], [0, -1, 0]]; bTc[0, 3] = 100; bTc[2, 3] = 300
    alive = [i for i in range(nkf) if i % 5 != 3]                         # the generator's key frames that are not null ...
    tdir = str(tmp_path / "traj"); os.makedirs(tdir)
    got_txt = m.save_trajectory(bTc, tdir, frame_ids=[7 * i + 3 for i in alive])      # ... and their frame ids
    capfd.readouterr()
    assert got_txt == open(a + "se2lam_kf_trajectory.txt").read() and got_txt.count("\n") == nk
    # ---- a file that names one pair of key frames twice: KeyFrame::addFtrMeasureFrom is std::map::insert, the first constraint stays.
    # The first entry of FtrGraphPairs is repeated behind itself with another measurement; reference and mirror both load the file
    # and save it again - the copy is gone from both, the first entry's measurement is the one that is kept
    text = open(a + "se2lam.map").read()
    head, _, tail = text.partition("FtrGraphPairs:\n")
    entries = tail.split("\n   -\n")
    if len(entries) > 1 and tail.startswith("   -\n"):
        first = entries[0][len("   -\n"):] if entries[0].startswith("   -\n") else entries[0]
        import re
        twin = re.sub(r"data: \[ [^,\]]*", "data: [ 1.2500000000000000e+02", first, count=1)
        assert twin != first
        c = str(tmp_path / "c") + "/"; d = str(tmp_path / "d") + "/"
        os.makedirs(c); os.makedirs(d)
        for f in os.listdir(a):
            if f.endswith(".bmp") or f.endswith(".txt"):
                open(c + f, "wb").write(open(a + f, "rb").read())
        open(c + "se2lam.map", "w").write(head + "FtrGraphPairs:\n   -\n" + first + "\n   -\n" + twin + "\n   -\n" + "\n   -\n".join(entries[1:]))
        docs_c, top_c = _parse(open(c + "se2lam.map").read())
        assert len(top_c["FtrGraphPairs"]) == len(top["FtrGraphPairs"]) + 1
        m2 = ref.RefMap(np.eye(3), np.eye(4), 2.0)
        m2.storage_load("\n".join(_fs_events(docs_c)) + "\n")
        got2, _ = _fs_canonical(m2.storage_save().split("\n"))
        capfd.readouterr()
        subprocess.run([mapstorage_exe, "copy", c, d], check=True, capture_output=True)
        docs_d, top_d = _parse(open(d + "se2lam.map").read())
        want2, dup2 = _fs_canonical(_fs_events(docs_d))
        assert dup2 == 0 and len(top_d["FtrGraphPairs"]) == len(pairs) and got2 == want2 == want
    # ---- sortMapPoints (MapStorage.cpp:98-118): points without good parallax leave the file, ids and matrix columns close up
    drop = list(range(0, nm, 4))
    for j in drop:
        m.storage_mp_set_good_prl(j, False)
    _, top2 = None, None
    ev2 = m.storage_save().split("\n")
    capfd.readouterr()
    keep = [j for j in range(nm) if j not in drop]
    O2, I2 = O[:, keep], I[:, keep]
    line = {l.split()[1]: ev2[k + 1] for k, l in enumerate(ev2) if l in ("K Observations", "K ObservationIndex")}
    assert line["Observations"] == f"M {nk} {len(keep)} i {O2.astype(np.int32).tobytes().hex()}" if keep and nk else True
    assert line["ObservationIndex"] == f"M {nk} {len(keep)} i {I2.astype(np.int32).tobytes().hex()}" if keep and nk else True
    at = ev2.index("K MapPoints")
    ids = [int(l[2:]) for k, l in enumerate(ev2[at:ev2.index("K Observations")]) if ev2[at + k - 1] == "K Id"]
    assert ids == list(range(len(keep)))


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_extractor_equals_the_compiled_reference(synth, ref_feats):
    """_ref == HIP on the ten config-1 frames, key points and descriptors bit-exact."""
    from se2lam_amd import orb
    ex = orb.ORBextractor()
    for t in range(10):
        assert _same_features(ex(synth.frame(t)), ref_feats[t]), t
    ex2 = orb.ORBextractor(500, 1.2, 8, orb.HARRIS_SCORE, 20)
    from oracle import oracle
    p = oracle.orb_params(500, 1.2, 8, 20, oracle.HARRIS_SCORE)
    for t in (0, 7):
        assert _same_features(ex2(synth.frame(t)), ref.orb_extract(synth.frame(t), p)), t


@pytest.mark.gpu
def test_hip_images_to_matches_equal_the_compiled_reference(synth, ref_feats):
    """FROM IMAGES through the C ABI - extract(frame t), extract(frame t + 1), MatchByWindow with the first
    frame's key points as vbPrevMatched - against the compiled reference run the same way on its own extraction.  No lexsort:
    the key-point ARRAYS, the descriptor rows and vnMatches12 are equal as they come, on the ten config-1 frames and on a
    sweep of other sizes / parameters (incl. noise: every cell and every level cut, heavy ties at both cuts)."""
    from se2lam_amd import capi, orb
    from se2lam_amd.matcher import ORBmatcher
    from oracle import oracle
    ex = orb.ORBextractor()
    mt = ORBmatcher(0.9)
    feats = [ex(synth.frame(t)) for t in range(10)]
    for t in range(10):
        assert np.array_equal(feats[t][0], ref_feats[t][0]) and np.array_equal(feats[t][1], ref_feats[t][1]), t
    for t in range(9):
        (k1, d1), (k2, d2) = feats[t], feats[t + 1]
        prev = _prev(k1)
        nm, m12 = mt.MatchByWindow(k1, d1, k2, d2, prev, 20)
        m_r, n_r, p_r = ref.match_window(*ref_feats[t], *ref_feats[t + 1])
        assert nm == n_r and np.array_equal(m12, m_r) and np.array_equal(prev, p_r), (t, nm, n_r)
    rng = np.random.default_rng(77)
    tex = synth.texture()
    ndone = 0
    for case in range(16):
        h, w = int(rng.integers(150, 600)), int(rng.integers(200, 800))
        nf, nl, th, sc = int(rng.integers(100, 2500)), int(rng.integers(2, 8)), int(rng.integers(7, 35)), int(rng.integers(0, 2))
        while min(h, w) / 1.2 ** (nl - 1) < 64:
            nl -= 1
        if case % 3 == 2:
            a = rng.integers(0, 256, (h, w)).astype(np.uint8); b = np.ascontiguousarray(np.roll(a, (-1, -2), (0, 1)))
        else:
            y0, x0 = int(rng.integers(0, 960 - h - 4)), int(rng.integers(0, 1280 - w - 6))
            a = np.ascontiguousarray(tex[y0:y0 + h, x0:x0 + w]); b = np.ascontiguousarray(tex[y0 + 1:y0 + 1 + h, x0 + 3:x0 + 3 + w])
        p = oracle.orb_params(nf, 1.2, nl, th, sc)
        try:
            (r1, e1), (r2, e2) = ref.orb_extract(a, p, cap=8192), ref.orb_extract(b, p, cap=8192)
        except ValueError:      # a cell rectangle outside its level: the reference itself raises on this geometry
            continue
        ex2 = orb.ORBextractor(nf, 1.2, nl, sc, th, max_rows=h, max_cols=w)   # (case 0: a grid whose last cell column is empty)
        (k1, d1), (k2, d2) = ex2(a), ex2(b)
        ndone += 1
        assert np.array_equal(k1, r1) and np.array_equal(d1, e1) and np.array_equal(k2, r2) and np.array_equal(d2, e2), (case, h, w, nf, nl, th, sc)
        if len(k1) == 0 or len(k2) == 0:
            continue
        prev = _prev(k1)
        nm, m12 = mt.MatchByWindow(k1, d1, k2, d2, prev, 20)
        m_r, n_r, p_r = ref.match_window(r1, e1, r2, e2)
        assert nm == n_r and np.array_equal(m12, m_r), (case, nm, n_r)
    assert ndone == 16


@pytest.mark.gpu
def test_hip_extractor_refuses_exactly_the_geometries_on_which_the_reference_raises(synth):
    """A cell window outside its pyramid level makes the reference raise (cv::Mat::colRange / rowRange, ORBextractor.cpp:608):
    with many features on a wide, low image the last-but-one cell column ends more than 13 px beyond the scan area.  The library
    refuses those geometries (SE2GPU_ERR_INVALID at the first call) and runs every other one: 120 random (size, features,
    levels, scale) combinations, the two sides must fail on the same ones (rounds 1-4: the library refused 16 % the reference
    runs, and returned key points on the 3 % where it raises)."""
    from se2lam_amd import capi, orb
    from oracle import oracle
    rng = np.random.default_rng(2026)
    tex = synth.texture()
    n_raise = n_run = 0
    for _ in range(120):
        h, w = int(rng.integers(120, 700)), int(rng.integers(160, 900))
        nl = int(rng.integers(1, 9)); scale = float(rng.choice([1.1, 1.2, 1.3, 1.5]))
        nf = int(rng.integers(50, 3000))
        if rng.random() < 0.5:          # wide and low with many features: where the reference's rectangles leave the level
            h, nf = int(rng.integers(120, 220)), int(rng.integers(1500, 3000))
        while min(h, w) / scale ** (nl - 1) < 64:
            nl -= 1
        img = np.ascontiguousarray(tex[:h, :w])
        try:
            kr, dr = ref.orb_extract(img, oracle.orb_params(nf, scale, nl, 20, 1), cap=16384)
            raised = False
        except ValueError:
            raised = True
        try:
            k, d = orb.ORBextractor(nf, scale, nl, 1, 20, max_rows=h, max_cols=w)(img)
            refused = False
        except capi.Se2GpuError as e:
            assert "the reference raises" in str(e), e
            refused = True
        assert raised == refused, (h, w, nf, scale, nl, raised, refused)
        if not raised:
            assert np.array_equal(k, kr) and np.array_equal(d, dr), (h, w, nf, scale, nl)
        n_raise += raised; n_run += not raised
    assert n_raise >= 5 and n_run >= 60, (n_raise, n_run)


@pytest.mark.gpu
def test_hip_matchers_equal_the_compiled_reference(ref_feats):
    from se2lam_amd.matcher import ORBmatcher
    mt = ORBmatcher(0.9)
    for a, b in ((0, 1), (1, 2), (0, 5), (3, 3)):
        (k1, d1), (k2, d2) = ref_feats[a], ref_feats[b]
        prev = _prev(k1)
        nm, m12 = mt.MatchByWindow(k1, d1, k2, d2, prev, 20)
        m_r, n_r, p_r = ref.match_window(k1, d1, k2, d2)
        assert nm == n_r and np.array_equal(m12, m_r) and np.array_equal(prev, p_r), (a, b)
    for seed in (0, 1):
        args = _projection_case(ref_feats, seed)
        nm, idx = ORBmatcher().MatchByProjection(*args, 15, 2)
        i_r, n_r = ref.match_projection(*args, 15, 2, 0.6)
        assert nm == n_r and np.array_equal(idx, i_r), seed
    (k1, d1), (k2, d2) = ref_feats[0], ref_feats[2]
    rng = np.random.default_rng(5)
    fv1, fv2 = _feature_vector(d1, 5), _feature_vector(d2, 5)
    h1 = (rng.random(len(k1)) < 0.7).astype(np.uint8); h2 = (rng.random(len(k2)) < 0.7).astype(np.uint8)
    nm, m12 = ORBmatcher(0.6).SearchByBoW(k1, d1, fv1, h1, k2, d2, fv2, h2, bIfMPOnly=False)
    m_r, n_r = ref.search_by_bow(k1, d1, fv1, h1, k2, d2, fv2, h2, False, 0.6, True)
    assert nm == n_r and np.array_equal(m12, m_r)


@pytest.mark.gpu
def test_hip_side_graph_construction_equals_the_compiled_reference(oracle, synth):
    """The product's mirrors of the optimizer.h calls against the reference's compiled optimizer.cpp: the initial robust cost of
    a window loaded through addCamPara ... addEdgeSE2XYZ, and both plane-motion priors."""
    from se2lam_amd import optimizer as op
    from se2lam_amd.localizer import addPlaneMotionSE3Expmap
    g = synth.ba_graph(P=20, L=600, seed=11)
    o = op.SlamOptimizer()
    o.load(g)
    o.initializeOptimization()
    assert np.isclose(o.activeRobustChi2(), ref.window_chi2(g)[0], rtol=1e-12, atol=0)
    for tilt in (False, True):
        Tbc = _tbc32(synth, tilt)      # what the reference reads out of its CV_32F Config::bTc
        Tq = _requat(Tbc)              # the same extrinsic as a rotation in double, which is what the C ABI takes
        for T in _off_plane_poses(synth, 4, seed=6):
            m, w = addPlaneMotionSE3Expmap(T, Tq)
            mr, wr, _ = ref.plane_motion_prior(T, Tbc)
            assert np.allclose(m, mr, rtol=0, atol=1e-9) and np.allclose(w, wr, rtol=1e-9, atol=1e-9 * np.abs(wr).max())
            pg = op.SlamOptimizer()
            Twc = np.linalg.inv(T)
            m, w = op.addVertexSE3PlaneMotion(pg, Twc, 0, Tq)
            mr, wr, _ = ref.pg_plane_motion_prior(Twc, Tbc)
            assert np.allclose(m, mr, rtol=0, atol=1e-9) and np.allclose(w, wr, rtol=1e-9, atol=1e-9 * np.abs(wr).max())


@pytest.mark.gpu
def test_hip_sparsifier_equals_the_compiled_reference(synth):
    from se2lam_amd.sparsifier import DoMarginalizeSE3XYZ_batch
    cases = _SPARSIFY_CASES + (_SPARSIFY_HARD,)
    pairs = [synth.kf_pair(N, s, b) for N, s, b in cases]
    got = DoMarginalizeSE3XYZ_batch(pairs)
    for case, (kf, mp, m_kf, m_mp, m_info), (z, info) in zip(cases, pairs, got):
        zr, ir = ref.sparsify(kf, mp, m_kf, m_mp, m_info)
        assert np.allclose(z, zr, atol=1e-12)
        tol = 1e-4 if case == _SPARSIFY_HARD else 1e-5      # BASELINE's BA tolerance; the hard pair: cond(H_marginal) 5.6e15
        assert np.abs(info - ir).max() <= tol * np.abs(ir).max(), case


@pytest.mark.gpu
@pytest.mark.parametrize("n_ref", [0, 3])
def test_hip_pod_loader_equals_the_compiled_load_local_graph(synth, n_ref):
    """se2gpu_ba_load_local_graph (vertex numbering, fixed rule, cov^-1, k_edge_information on the device) fed with the flat
    arrays of the map the compiled reference holds: the robust cost of the loaded graph equals that of the graph
    Map::loadLocalGraph put into the recording optimizer; the fixed key frames do not move."""
    from se2lam_amd import optimizer as op
    m, w = _reference_window(synth, n_ref)
    g, inp, nL = w["g"], w["inp"], w["nL"]
    m.update_local_graph(0)
    out = m.load_local_graph()
    P = g.P
    local_mp = sorted(set(int(l) for k, l in zip(g.e_kf, g.e_lm) if k < nL))
    remap = np.full(g.L, -1); remap[local_mp] = np.arange(len(local_mp))
    keep = remap[g.e_lm] >= 0
    order = np.argsort(remap[g.e_lm[keep]], kind="stable")                        # observations grouped by map point
    Rcw = np.stack([m.kf_pose(a)[:3, :3] for a in range(P)]).reshape(P, 9)
    odo_to = np.full(nL, -1, np.int32); odo_meas = np.zeros((nL, 3)); odo_cov = np.tile(np.eye(3).reshape(-1), (nL, 1))
    for i, (j, meas, cov) in w["odo"].items():
        if i < nL and j < nL:
            odo_to[i] = j; odo_meas[i] = meas; odo_cov[i] = cov.reshape(-1)
    pod = op.SlamOptimizer()
    op.loadLocalGraph(pod, kf_id=w["frame_id"].astype(np.int32), kf_Twb=w["twb"], kf_Rcw=Rcw, n_local=nL, odo_to=odo_to, odo_meas=odo_meas,
                      odo_cov=odo_cov, mp_pos=g.lms[local_mp].astype(np.float32), obs_mp=remap[g.e_lm[keep]][order], obs_kf=g.e_kf[keep][order],
                      obs_uv=w["uv32"][keep][order], obs_lc=inp["lc"][keep][order], obs_sigma2=inp["sigma2"][keep][order], K=w["K"], Rbc=g.Rbc,
                      tbc=g.tbc, huber=w["huber"])
    pod.initializeOptimization(0)
    assert np.isclose(pod.activeRobustChi2(), out["chi2"], rtol=1e-9, atol=0)
    pod.optimize(3)
    for a in np.nonzero(out["v_fixed"][:P])[0]:
        assert np.array_equal(op.estimateVertexSE2(pod, int(a)), w["twb"][a].astype(np.float64))
    moved = [a for a in range(P) if not out["v_fixed"][a] and not np.array_equal(op.estimateVertexSE2(pod, a), w["twb"][a].astype(np.float64))]
    assert len(moved) == P - int(out["v_fixed"][:P].sum())


@pytest.mark.gpu
@pytest.mark.parametrize("n_ref", [0, 3])
def test_hip_se3_local_graph_cost_equals_the_compiled_reference(synth, n_ref):
    """The SE3-expmap graph the compiled Map::loadLocalGraph(optimizer, vpEdgesAll, vnAllIdx) built, loaded into the library
    through the optimizer.h call surface: same robust cost, same per-edge chi2 (what LocalMapper::removeOutlierChi2 reads)."""
    from se2lam_amd import optimizer as op
    m, w, out, g3 = _se3_window(synth, n_ref)
    o = op.SlamOptimizer()
    op.load_se3_graph(o, g3)
    o.initializeOptimization(0)
    assert np.isclose(o.activeRobustChi2(), out["chi2"], rtol=1e-9, atol=0)
    chi = op.edgeChi2(o, g3.E)
    assert np.allclose(chi, out["e_chi2"], rtol=1e-8, atol=1e-12)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n", [(7, 600), (11, 1000)])
def test_hip_triangulate_equals_the_compiled_do_triangulate(seed, n):
    from se2lam_amd.matcher import doTriangulate
    K, Tcr, k1, k2, match, has_obs, P1, P2, Ocam, X = _triangulation_scene(n, seed)
    pos, good, m, ng, nold = ref.track_triangulate(K, k1, k2, match, has_obs, X, Tcr, 500.0, 8000.0)
    got = doTriangulate(k1, k2, match, has_obs, P1, P2, Ocam, 500.0, 8000.0, 2)
    assert got[3:] == (ng, nold) and np.array_equal(got[1], good) and np.array_equal(got[2], m)
    acc = (match >= 0) & (has_obs == 0) & (m >= 0)
    assert np.abs(got[0][acc] - pos[acc]).max() <= 1e-5 * np.abs(pos[acc]).max()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n", [(0, 400), (2, 37)])
def test_hip_pose_only_ba_starts_from_the_compiled_graph(seed, n):
    """se2gpu_track_pose_ba on the numbers of the graph the compiled Localizer::DoLocalBA built: same cost at the start."""
    from se2lam_amd.localizer import Localizer, addPlaneMotionSE3Expmap
    Tcw0, Xw, kps, TBC, F, CX, CY, DELTA = _pose_only_case(seed, n)
    K = np.array([[F, 0, CX], [0, F, CY], [0, 0, 1]], np.float32)
    good = np.ones(n, np.uint8); good[::9] = 0
    out = ref.localizer_do_local_ba(K, TBC, np.float32(DELTA), Tcw0, kps, Xw, good)
    sel = out["e_point"]
    T0 = _requat(Tcw0.astype(np.float64))
    loc = Localizer()
    loc.DoLocalBA(T0, _requat(np.asarray(TBC, np.float64)), Xw[sel].astype(np.float64), out["e_uv"], out["e_w"], float(K[0, 0]), float(K[0, 2]),
                  float(K[1, 2]), float(np.float32(DELTA)), 30)
    assert np.isclose(loc.stats["chi2_init"], out["chi2"], rtol=1e-9, atol=0)


@pytest.mark.gpu
def test_hip_pose_graph_cost_equals_the_compiled_global_ba(synth):
    """The pose graph the compiled GlobalMapper::GlobalBA built, loaded through the optimizer.h call surface: same cost."""
    from se2lam_amd import optimizer as op
    m, w = _global_map(synth)
    out = m.global_ba()
    P = w["P"]
    pg = synth.PoseGraph(poses=out["v_est"], fixed=out["v_fixed"].astype(np.uint8), has_prior=np.ones(P, np.uint8), prior_meas=out["p_meas"],
                         prior_info=out["p_info"], o_i=out["e_ids"][:, 0].astype(np.int32), o_j=out["e_ids"][:, 1].astype(np.int32),
                         o_meas=out["e_meas"], o_info=out["e_info"])
    o = op.SlamOptimizer()
    op.load_pose_graph(o, pg)
    o.initializeOptimization(0)
    assert np.isclose(o.activeRobustChi2(), out["chi2"], rtol=1e-9, atol=0)
    chi = op.edgeChi2(o, pg.O)
    assert np.allclose(chi, out["e_chi2"], rtol=1e-8, atol=1e-12)
