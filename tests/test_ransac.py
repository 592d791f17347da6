"""Track::removeOutliers (cv::findFundamentalMat RANSAC / LMedS mask, Track.cpp:308-344): device path against the CPU
restatement, bit for bit; plus properties of the restatement itself (no GPU)."""
import numpy as np
import pytest


def _two_views(seed, n, outlier_frac=0.3, noise=0.5):
    rng = np.random.default_rng(seed)
    X = np.stack([rng.uniform(-3000, 3000, n), rng.uniform(-2000, 2000, n), rng.uniform(3000, 9000, n)], 1)
    K = np.array([[400, 0, 320], [0, 400, 240], [0, 0, 1.0]])
    th = rng.uniform(-0.08, 0.08)
    R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
    t = rng.uniform(-300, 300, 3)

    def proj(R, t):
        uv = (X @ R.T + t) @ K.T
        return uv[:, :2] / uv[:, 2:]

    p1 = proj(np.eye(3), np.zeros(3)) + rng.normal(0, noise, (n, 2))
    p2 = proj(R, t) + rng.normal(0, noise, (n, 2))
    out = rng.random(n) < outlier_frac
    p2[out] += rng.uniform(-80, 80, (int(out.sum()), 2))
    return p1.astype(np.float32), p2.astype(np.float32), out


# ---------------------------------------------------------------- restatement properties (CPU)

def test_oracle_exact_geometry_keeps_everything(oracle):
    for seed in range(4):
        p1, p2, _ = _two_views(seed, 300, outlier_frac=0.0, noise=0.0)
        mask, ni = oracle.fundamental_mask(p1, p2)
        assert ni == 300 and mask.all()


def test_oracle_rejects_gross_outliers(oracle):
    p1, p2, out = _two_views(11, 500)
    mask, ni = oracle.fundamental_mask(p1, p2)
    assert ni == mask.sum()
    assert mask[~out].mean() > 0.7 and mask[out].mean() < 0.1


def test_oracle_small_counts(oracle):
    p1, p2, _ = _two_views(3, 40, outlier_frac=0.0)
    for n in range(0, 7):       # findFundamentalMat returns before the mask exists
        mask, ni = oracle.fundamental_mask(p1[:n], p2[:n])
        assert ni == 0 and not mask.any()
    mask, ni = oracle.fundamental_mask(p1[:7], p2[:7])
    assert ni == 7 and mask.all()
    for n in range(8, 15):      # LMedS keeps at least the sample size
        mask, ni = oracle.fundamental_mask(p1[:n], p2[:n])
        assert 7 <= ni <= n and ni == mask.sum()


def test_oracle_remove_outliers_semantics(oracle, synth):
    k1, _ = oracle.orb_extract(synth.frame(0))
    k2 = k1.copy()
    m = np.full(len(k1), -1, np.int32)
    m[:9] = np.arange(9)                      # 9 matches can never give 10 inliers
    out, ni = oracle.remove_outliers(k1, k2, m)
    assert ni == 0 and (out == -1).all()


def test_oracle_seven_point_models_satisfy_their_definition(oracle):
    """run7Point: every returned matrix is singular, puts the seven sample correspondences on their epipolar lines and
    has F(3,3) = 1; on exact two-view geometry one of them is the true fundamental matrix (up to scale)."""
    p1, p2, _ = _two_views(5, 200, outlier_frac=0.0, noise=0.0)
    rng = np.random.default_rng(1)
    found_true = 0
    for _ in range(20):
        idx = rng.choice(200, 7, replace=False)
        Fs = oracle.seven_point(p1, p2, idx)
        assert 1 <= len(Fs) <= 3
        for Fm in Fs:
            assert abs(Fm[2, 2] - 1.0) < 1e-12
            sv = np.linalg.svd(Fm, compute_uv=False)
            assert sv[2] < 1e-7 * sv[0]                                          # det F = 0
            x1 = np.c_[p1[idx].astype(np.float64), np.ones(7)]; x2 = np.c_[p2[idx].astype(np.float64), np.ones(7)]
            r = np.einsum("ij,jk,ik->i", x2, Fm, x1)
            assert np.abs(r).max() < 1e-6 * np.abs(Fm).max() * 640 * 640         # x2^T F x1 = 0 on the sample
        a1 = np.c_[p1.astype(np.float64), np.ones(200)]; a2 = np.c_[p2.astype(np.float64), np.ones(200)]
        res = [np.abs(np.einsum("ij,jk,ik->i", a2, Fm, a1)).max() / (np.abs(Fm).max() * 640 * 640) for Fm in Fs]
        found_true += min(res) < 1e-4                                           # float32 pixel coordinates
    assert found_true == 20


def test_oracle_ransac_samples_are_distinct_and_reproducible(oracle):
    a = oracle.ransac_subsets(700, 50); b = oracle.ransac_subsets(700, 50)
    assert np.array_equal(a, b) and a.min() >= 0 and a.max() < 700
    assert all(len(set(row)) == 7 for row in a)
    assert np.array_equal(oracle.ransac_subsets(700, 10), a[:10])
    assert not np.array_equal(oracle.ransac_subsets(40, 5), a[:5])


# ---------------------------------------------------------------- device path

@pytest.mark.gpu
@pytest.mark.parametrize("seed,n,frac", [(0, 700, 0.3), (1, 400, 0.5), (2, 1000, 0.1), (3, 60, 0.2), (4, 15, 0.0),
                                         (5, 16, 0.3), (6, 250, 0.7), (7, 900, 0.0)])
def test_fundamental_mask_matches_restatement(oracle, seed, n, frac):
    from se2lam_amd.track import Track
    p1, p2, _ = _two_views(seed, n, outlier_frac=frac)
    tr = Track()
    mask, ni = tr.findFundamentalMat(p1, p2)
    mask_ref, ni_ref = oracle.fundamental_mask(p1, p2)
    assert ni == ni_ref
    assert np.array_equal(mask, mask_ref)
    info = tr.last_ransac()
    assert info["inliers"] == ni and 0 <= info["sample"] < 1000 and 0 <= info["iterations"] <= 1000   # all inliers => 0 left


@pytest.mark.gpu
def test_fundamental_mask_small_counts_and_lmeds(oracle):
    from se2lam_amd.track import Track
    tr = Track()
    for seed in range(3):
        p1, p2, _ = _two_views(20 + seed, 14, outlier_frac=0.15)
        for n in range(0, 15):
            mask, ni = tr.findFundamentalMat(p1[:n], p2[:n])
            mask_ref, ni_ref = oracle.fundamental_mask(p1[:n], p2[:n])
            assert ni == ni_ref and np.array_equal(mask, mask_ref), (seed, n)


@pytest.mark.gpu
def test_fundamental_mask_degenerate_inputs(oracle):
    """duplicated correspondences, identical frames (rank-deficient systems, NaN models): same answer on both sides"""
    from se2lam_amd.track import Track
    tr = Track()
    p1, p2, _ = _two_views(31, 100)
    cases = [(p1, p1.copy()),                                               # no motion at all
             (np.repeat(p1[:10], 10, 0), np.repeat(p2[:10], 10, 0)),         # 10 distinct pairs, 10 copies each
             (p1, p1 + np.float32(5.0)),                                     # pure image translation
             (np.zeros_like(p1), np.zeros_like(p2))]                         # everything at the origin
    for a, b in cases:
        mask, ni = tr.findFundamentalMat(a, b)
        mask_ref, ni_ref = oracle.fundamental_mask(a, b)
        assert ni == ni_ref and np.array_equal(mask, mask_ref)


@pytest.mark.gpu
def test_remove_outliers_on_orb_matches(oracle, synth):
    """Track::mTrack order: MatchByWindow then removeOutliers on the same vectors (Track.cpp:131-134)"""
    from se2lam_amd.matcher import ORBmatcher
    from se2lam_amd.track import Track
    (k1, d1), (k2, d2) = oracle.orb_extract(synth.frame(0)), oracle.orb_extract(synth.frame(2))
    prev = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], 1), np.float32)
    nm, m12 = ORBmatcher(0.9).MatchByWindow(k1, d1, k2, d2, prev, 20)
    rng = np.random.default_rng(0)
    m12 = m12.copy()
    bad = rng.choice(np.flatnonzero(m12 >= 0), 60, replace=False)           # corrupt some matches
    m12[bad] = rng.integers(0, len(k2), 60)
    ref, ni_ref = oracle.remove_outliers(k1, k2, m12)
    got = m12.copy()
    ni = Track().removeOutliers(k1, k2, got)
    assert ni == ni_ref and ni > 100
    assert np.array_equal(got, ref)
    assert (got[bad] >= 0).mean() < 0.2
    # fewer than 10 inliers: everything is dropped
    few = np.full(len(k1), -1, np.int32)
    few[:9] = m12[:9]
    assert Track().removeOutliers(k1, k2, few) == 0 and (few == -1).all()
    # nothing matched at all
    none = np.full(len(k1), -1, np.int32)
    assert Track().removeOutliers(k1, k2, none) == 0
