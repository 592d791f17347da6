"""Track::doTriangulate (Track.cpp:378-419; cvu::triangulate / checkParallax cvutil.cpp:46-59, 92-98) - SURVEY 8(f).3.

CPU: the oracle recovers synthetic 3-D points from their two projections and applies the depth / parallax gates.
GPU: the HIP path (se2gpu_triangulate) equals the oracle bit for bit (same FP64 Jacobi sequence, no contraction).
Parity with OpenCV's FP32 cv::SVD iteration is unpinned (OpenCV is not installed); Track::doTriangulate itself runs as compiled in
tests/test_ref_compiled.py (which matches, gates and counters: identical; positions to 1e-5)."""
import numpy as np
import pytest

KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"),
               ("class_id", "<i4")])


def scene(n=600, seed=7):
    rng = np.random.default_rng(seed)
    K = np.array([[400.0, 0, 320.0], [0, 400.0, 240.0], [0, 0, 1]], np.float32)
    # current camera: small rotation about y and a 150 mm baseline (se2lam works in millimetres)
    th = 0.03
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], np.float32)
    t = np.array([-150.0, 5.0, 20.0], np.float32)
    Tcr = np.eye(4, dtype=np.float32); Tcr[:3, :3] = R; Tcr[:3, 3] = t
    P1 = (K @ np.eye(3, 4, dtype=np.float32)).astype(np.float32)       # Config::PrjMtrxEye
    P2 = (K @ Tcr[:3]).astype(np.float32)                              # Config::Kcam * Tcr.rowRange(0,3)
    X = np.stack([rng.uniform(-1500, 1500, n), rng.uniform(-800, 800, n), rng.uniform(300, 12000, n)], 1).astype(np.float32)
    Xh = np.concatenate([X, np.ones((n, 1), np.float32)], 1)
    u1 = (P1 @ Xh.T).T; u1 = u1[:, :2] / u1[:, 2:]
    u2 = (P2 @ Xh.T).T; u2 = u2[:, :2] / u2[:, 2:]
    k1 = np.zeros(n, KP); k2 = np.zeros(n, KP)
    k1["x"], k1["y"] = u1[:, 0], u1[:, 1]
    perm = rng.permutation(n)
    k2["x"][perm], k2["y"][perm] = u2[:, 0], u2[:, 1]
    match = perm.astype(np.int32)
    match[::17] = -1                                  # unmatched features
    has_obs = np.zeros(n, np.uint8); has_obs[5::23] = 1
    Ocam = np.linalg.inv(Tcr)[:3, 3].astype(np.float32)
    return k1, k2, match, has_obs, P1, P2, Ocam, X


def test_oracle_recovers_points_and_gates(oracle):
    k1, k2, match, has_obs, P1, P2, Ocam, X = scene()
    lower, upper = 500.0, 8000.0
    pos, good, m, ng, nold = oracle.triangulate(k1, k2, match, has_obs, P1, P2, Ocam, lower, upper, 2)
    done = (match >= 0) & (has_obs == 0)
    assert nold == int(((match >= 0) & (has_obs == 1)).sum())
    # Config::acceptDepth: a point is kept (mLocalMPs[i] = pos, Track.cpp:407-408) only inside [lower, upper]; outside, the
    # match is dropped and the position stays untouched (zero here).  Points within 1 % of a bound may fall either way.
    clear = (np.abs(X[:, 2] - lower) > 0.01 * lower) & (np.abs(X[:, 2] - upper) > 0.01 * upper)
    inside = (X[:, 2] >= lower) & (X[:, 2] <= upper)
    acc = done & (m >= 0)
    assert np.array_equal(acc[clear & done], inside[clear & done])
    err = np.abs(pos[acc] - X[acc]).max(axis=1) / X[acc, 2]
    assert err.max() < 2e-3                                     # float pixel coordinates: ~1e-4 of the depth
    assert not pos[~acc].any()
    assert np.array_equal(m[acc], match[acc])
    assert np.array_equal(m[~done], match[~done])
    inside = acc
    # parallax: cos of the angle at the point between the two camera centres < 0.9994 (2 degrees)
    p1, p2 = pos, pos - Ocam
    cosp = np.abs((p1 * p2).sum(1)) / (np.linalg.norm(p1, axis=1) * np.linalg.norm(p2, axis=1) + 1e-30)
    sel = done & inside & (np.abs(cosp - 0.9994) > 1e-5)
    assert np.array_equal(good[sel].astype(bool), cosp[sel] < 0.9994)
    assert ng == int(good.sum()) and 0 < ng < done.sum()


@pytest.mark.gpu
@pytest.mark.parametrize("seed,n", [(7, 600), (11, 1000), (3, 1)])
def test_hip_triangulate_equals_oracle(oracle, seed, n):
    from se2lam_amd.matcher import doTriangulate
    k1, k2, match, has_obs, P1, P2, Ocam, X = scene(n, seed)
    for mind in (1, 2, 4):
        ref = oracle.triangulate(k1, k2, match, has_obs, P1, P2, Ocam, 500.0, 8000.0, mind)
        got = doTriangulate(k1, k2, match, has_obs, P1, P2, Ocam, 500.0, 8000.0, mind)
        assert np.array_equal(got[0], ref[0])          # positions: bit-exact
        assert np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
        assert got[3:] == ref[3:]
    got = doTriangulate(k1[:0], k2, match[:0], None, P1, P2, Ocam, 500.0, 8000.0)
    assert len(got[0]) == 0 and got[3] == 0


@pytest.mark.gpu
def test_hip_track_workspace_triangulate_equals_oracle(oracle):
    """the same pass through the tracking thread's persistent workspace (se2gpu_track_triangulate), sizes varying
    between calls of one handle"""
    from se2lam_amd.track import Track
    tr = Track()
    for seed, n in ((7, 600), (3, 1), (11, 1000), (5, 37)):
        k1, k2, match, has_obs, P1, P2, Ocam, X = scene(n, seed)
        for ho in (has_obs, None):
            ref = oracle.triangulate(k1, k2, match, ho, P1, P2, Ocam, 500.0, 8000.0, 2)
            got = tr.doTriangulate(k1, k2, match, ho, P1, P2, Ocam, 500.0, 8000.0, 2)
            assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2])
            assert got[3:] == ref[3:]
    got = tr.doTriangulate(k1[:0], k2, match[:0], None, P1, P2, Ocam, 500.0, 8000.0)
    assert len(got[0]) == 0 and got[3] == 0


@pytest.mark.gpu
def test_match_past_the_current_frame_is_an_error(oracle):
    """a match index >= the current frame's feature count is a caller error (ERR_INVALID), not "no match" - both entry
    points; the reference would read past keyPointsUn there"""
    from se2lam_amd import capi
    from se2lam_amd.matcher import doTriangulate
    from se2lam_amd.track import Track
    k1, k2, match, has_obs, P1, P2, Ocam, X = scene(50, 2)
    bad = match.copy()
    bad[np.nonzero(bad >= 0)[0][3]] = len(k2)
    for fn in (doTriangulate, Track().doTriangulate):
        with pytest.raises(capi.Se2GpuError) as e:
            fn(k1, k2, bad, has_obs, P1, P2, Ocam, 500.0, 8000.0, 2)
        assert e.value.code == capi.ERR_INVALID
