"""Pins of the oracle (and, with -m gpu, of the HIP path) that do not come from the restatement itself.

The reference ships no golden vectors and OpenCV / g2o cannot be built here, so the oracle's outputs
cannot be compared with the real libraries in this container.  What CAN be done without them:

(a) BA: the generator's own numpy projection model + scipy.optimize.least_squares (trust-region reflective,
    finite-difference Jacobian) define the robust optimum independently of oracle/ba_ref.cpp and csrc/ba.hip:
    the cost function agrees at arbitrary states, a cold-started scipy run and the LM restatement reach the same
    optimum on a small graph, and at BASELINE config 3 the converged estimate is a stationary point of the independent cost.
(b) ORB: float references of the OpenCV primitives (half-pixel bilinear, sigma = 2 7x7 Gaussian) bound the
    fixed-point restatements to +-1 grey level, so a gross convention error (pixel-centre mapping, border, taps)
    cannot hide behind "GPU == oracle".
(c) tests/test_opencv_pin.py regenerates the fixtures with real cv2 wherever OpenCV exists.
"""
import numpy as np
import pytest

from independent import BAProblemNumpy, bilinear_half_pixel, gaussian_float


# ------------------------------------------------------------------------------------------------ (a) BA
@pytest.mark.parametrize("P,L", [(8, 60), (21, 800), (50, 5000)])
def test_independent_cost_equals_oracle_chi2(oracle, synth, P, L):
    """chi^2 (Huber, information convention, PreEdgeSE2 without angle wrap) from the definitions == the oracle's
    activeRobustChi2, at the initial state and after 10 LM iterations."""
    g = synth.ba_graph(P, L)
    pb = BAProblemNumpy(g)
    assert pb.cost(g.poses, g.lms) == pytest.approx(oracle.ba_chi2(g), rel=1e-12)
    p, l, st = oracle.ba_optimize(g, 10, 0)
    assert pb.cost(p, l) == pytest.approx(st["chi2_final"], rel=1e-12)
    assert pb.cost(p, l) == pytest.approx(oracle.ba_chi2(g, p, l), rel=1e-12)
    # the Huber branch is really exercised by this graph
    f = pb.fun(pb.pack(g.poses, g.lms))[:2 * g.E].reshape(-1, 2)
    assert ((f * f).sum(1) > g.huber ** 2).sum() > 0


def test_small_graph_scipy_optimum_equals_converged_lm(oracle, synth):
    """8 KF / 60 landmarks, cold start: scipy's trust-region solver on the independent cost and the restated g2o
    Levenberg-Marquardt converge to the same robust cost (1e-9) and the same poses (1e-5 of the update - BASELINE's bar)."""
    g = synth.ba_graph(8, 60)
    pb = BAProblemNumpy(g)
    r = pb.solve(pb.pack(g.poses, g.lms))
    assert r.status > 0
    p_ref, l_ref, st = oracle.ba_optimize(g, 60, 0)
    assert st["terminated"]                                        # LM stopped by itself: converged
    assert 2 * r.cost == pytest.approx(st["chi2_final"], rel=1e-9)
    ps, ls = pb.unpack(r.x)
    upd = np.abs(p_ref - g.poses).max()
    assert np.abs(ps - p_ref).max() <= 1e-5 * upd
    # landmarks: compare through what they are observed by (a 2-view landmark's depth is a flat direction)
    u1, v1, _, _ = synth._project(ps, ls, g.e_kf, g.e_lm, pb.Rcb, pb.tcb, g.fx, g.cx, g.cy)
    u2, v2, _, _ = synth._project(p_ref, l_ref, g.e_kf, g.e_lm, pb.Rcb, pb.tcb, g.fx, g.cx, g.cy)
    assert max(np.abs(u1 - u2).max(), np.abs(v1 - v2).max()) < 1e-4     # pixels


def test_config3_converged_estimate_is_stationary_for_scipy(oracle, synth):
    """BASELINE config 3 (50 KF / 5k landmarks / ~30k edges): at the LM restatement's converged estimate the
    independent cost has a vanishing finite-difference gradient (1e-7 of the initial one) and scipy, started
    there, neither lowers the cost (1e-9) nor moves a pose (1e-6 of the update)."""
    g = synth.ba_graph(50, 5000)
    pb = BAProblemNumpy(g)
    p_ref, l_ref, st = oracle.ba_optimize(g, 200, 0)
    x0, xw = pb.pack(g.poses, g.lms), pb.pack(p_ref, l_ref)
    from scipy.optimize._numdiff import approx_derivative, group_columns
    sp = pb.sparsity()
    groups = group_columns(sp)

    def grad(x):
        J = approx_derivative(pb.fun, x, method="3-point", sparsity=(sp, groups))
        return J.T @ pb.fun(x)

    g0, gw = grad(x0), grad(xw)
    # the flattest directions (depth of two-view landmarks) keep LM drifting by 1e-7 of the cost per iteration; pose
    # gradient is what the north star bounds
    assert np.abs(gw[:pb.npz]).max() <= 1e-7 * np.abs(g0[:pb.npz]).max()
    assert np.abs(gw).max() <= 1e-5 * np.abs(g0).max()
    r = pb.solve(xw, max_nfev=10)
    assert 2 * r.cost <= st["chi2_final"] * (1 + 1e-12)
    assert 2 * r.cost >= st["chi2_final"] * (1 - 1e-9)
    ps, _ = pb.unpack(r.x)
    assert np.abs(ps - p_ref).max() <= 1e-6 * np.abs(p_ref - g.poses).max()


@pytest.mark.gpu
@pytest.mark.parametrize("P,L", [(8, 60), (50, 5000), (200, 20000)])
def test_hip_estimate_has_the_independent_cost_it_reports(synth, P, L):
    """The HIP path's reported chi^2 (initial, after 10 iterations) equals the independent numpy cost of the
    estimates it hands back - no oracle involved."""
    from se2lam_amd.optimizer import SlamOptimizer
    g = synth.ba_graph(P, L)
    pb = BAProblemNumpy(g)
    o = SlamOptimizer()
    o.load(g)
    o.initializeOptimization(0)
    assert o.activeRobustChi2() == pytest.approx(pb.cost(g.poses, g.lms), rel=1e-12)
    o.optimize(10)
    poses, lms = o.estimates()
    assert o.stats["chi2_final"] == pytest.approx(pb.cost(poses, lms), rel=1e-11)
    assert o.stats["chi2_final"] < 0.5 * o.stats["chi2_init"]


@pytest.mark.gpu
def test_hip_small_graph_converges_to_the_scipy_optimum(synth):
    from se2lam_amd.optimizer import SlamOptimizer
    g = synth.ba_graph(8, 60)
    pb = BAProblemNumpy(g)
    r = pb.solve(pb.pack(g.poses, g.lms))
    o = SlamOptimizer()
    o.load(g)
    o.initializeOptimization(0)
    o.optimize(60)
    assert o.stats["terminated"]
    assert o.stats["chi2_final"] == pytest.approx(2 * r.cost, rel=1e-9)
    ps, _ = pb.unpack(r.x)
    poses, _ = o.estimates()
    assert np.abs(ps - poses).max() <= 1e-5 * np.abs(poses - g.poses).max()


# ------------------------------------------------------------------------------------------------ (b) ORB
def test_resize_within_one_grey_level_of_float_bilinear(oracle, synth):
    """Every pyramid level of the oracle (cv::resize restatement: 11-bit fixed-point coefficients, level k from level
    k-1, ORBextractor.cpp:809) is within +-1 of float bilinear interpolation with the half-pixel mapping of that
    level's source (mean error 0.26, that of rounding plus OpenCV's truncating vertical pass).  torch's interpolate(align_corners=False) is the same
    definition and is checked against the numpy one."""
    import torch
    geo = oracle.orb_geometry(480, 640)
    for t in (0, 5):
        img = synth.frame(t)
        prev = img
        for lv in range(1, 8):
            w, h = int(geo[lv, 0]), int(geo[lv, 1])
            got = oracle.orb_level(img, lv)
            assert got.shape == (h, w)
            ref = bilinear_half_pixel(prev, w, h)
            tref = torch.nn.functional.interpolate(torch.from_numpy(prev.astype(np.float64))[None, None], size=(h, w),
                                                   mode="bilinear", align_corners=False, antialias=False)[0, 0].numpy()
            assert np.abs(ref - tref).max() < 1e-9
            d = np.abs(got.astype(np.float64) - ref)
            assert d.max() <= 1.0, (lv, d.max())
            # OpenCV's 8-bit path truncates twice in the vertical pass ((b0 * (S0 >> 4)) >> 16 + (b1 * (S1 >> 4)) >> 16 + 2) >> 2:
            # a known bias of about -1/8 grey level on top of the rounding error of 1/4
            assert np.abs(d).mean() <= 0.3 and -0.2 < (got - ref).mean() < 0.05, (lv, np.abs(d).mean(), (got - ref).mean())
            prev = got


def test_blur_within_one_grey_level_of_float_gaussian(oracle, synth):
    """The 8-bit fixed-point GaussianBlur restatement (ORBextractor.cpp:769) against the float sigma = 2 7x7 Gaussian with
    BORDER_REFLECT_101.  OpenCV 3.2 rounds each float tap to 8 bits ({18,34,49,55,49,34,18}: sum 257, not 256), so its
    output carries a gain of (257/256)^2 = 1.0078 - up to 2 grey levels at white - which the float reference is given
    too; what is left must be rounding (+-1, mean < 0.3).  The level is a view into a bordered image, so cv::GaussianBlur
    reads the real frame pixels at the level's edge: only pixels >= 3 px inside are compared."""
    taps = np.exp(-((np.arange(7) - 3.0) ** 2) / 8.0)
    assert np.rint(256 * taps / taps.sum()).astype(int).tolist() == oracle.orb_gaussian_taps().tolist()
    gain = (257.0 / 256.0) ** 2
    for t in (0, 7):
        img = synth.frame(t)
        for lv in (0, 2, 5, 7):
            src = oracle.orb_level(img, lv)
            got = oracle.orb_level(img, lv, blurred=True)
            ref = np.minimum(gaussian_float(src) * gain, 255.0)
            d = np.abs(got.astype(np.float64) - ref)[3:-3, 3:-3]
            assert d.max() <= 1.0, (lv, d.max())
            assert d.mean() < 0.3


def test_reflect101_frame_of_the_pyramid(oracle, synth):
    """copyMakeBorder(BORDER_REFLECT_101) frame around every level (ORBextractor.cpp:815-826) against numpy's 'reflect' pad."""
    img = synth.frame(2)
    for lv in (0, 1, 6):
        inner = oracle.orb_level(img, lv)
        framed = oracle.orb_level(img, lv, bordered=True)
        assert np.array_equal(framed, np.pad(inner, 16, mode="reflect"))


def test_fast_corners_agree_with_brute_force_segment_test(oracle, synth):
    """On a real pyramid level: the oracle's raw score plane S satisfies  S > t  <=>  the brute-force FAST-9/16 segment
    test passes at t  (t = 7, 20: the two thresholds of ORBextractor.cpp:616,622), and every key point the extractor
    returns on that level is a brute-force corner at threshold 7 that no in-cell neighbour out-scores."""
    from test_orb_oracle import _segment_test
    img = synth.frame(3)
    lv = 4
    src = oracle.orb_level(img, lv)
    S = oracle.orb_score(img, lv).astype(int)
    h, w = src.shape
    rng = np.random.default_rng(1)
    ys = rng.integers(3, h - 3, 1500); xs = rng.integers(3, w - 3, 1500)
    hits = 0
    for y, x in zip(ys, xs):
        for t in (7, 20):
            ok = _segment_test(src[y - 3:y + 4, x - 3:x + 4], t)
            assert (S[y, x] > t) == ok, (y, x, t, S[y, x])
            hits += ok
    assert hits > 50
    k, _ = oracle.orb_extract(img)
    sc = oracle.orb_tables()["scale"][lv]
    m = k["octave"] == lv
    assert m.sum() > 50
    for kp in k[m]:
        x, y = int(round(kp["x"] / sc)), int(round(kp["y"] / sc))
        assert _segment_test(src[y - 3:y + 4, x - 3:x + 4], 7)
        assert kp["response"] == S[y, x] - 1                                   # cornerScore = S - 1
