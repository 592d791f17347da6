"""CPU tests of the matcher oracle (oracle/match_ref.cpp): constants and integer arithmetic that the
reference tree pins (ORBmatcher.cpp:45-47,110-126; Frame.h:26-27; Frame.cpp:209-286)."""
import numpy as np
import pytest


def test_descriptor_distance_is_popcount(oracle):
    rng = np.random.default_rng(0)
    for _ in range(200):
        a = rng.integers(0, 256, 32).astype(np.uint8)
        b = rng.integers(0, 256, 32).astype(np.uint8)
        want = int(np.unpackbits(a ^ b).sum())
        assert oracle.hamming(a, b) == want
    z = np.zeros(32, np.uint8)
    assert oracle.hamming(z, z) == 0 and oracle.hamming(z, ~z) == 256


def test_library_host_hamming_matches(oracle):
    from se2lam_amd.matcher import ORBmatcher
    rng = np.random.default_rng(1)
    for _ in range(50):
        a = rng.integers(0, 256, 32).astype(np.uint8)
        b = rng.integers(0, 256, 32).astype(np.uint8)
        assert ORBmatcher.DescriptorDistance(a, b) == oracle.hamming(a, b)


def _kps(xy, octave):
    k = np.zeros(len(xy), oracle_kp_dtype())
    k["x"], k["y"] = xy[:, 0], xy[:, 1]
    k["octave"] = octave
    k["class_id"] = -1
    return k


def oracle_kp_dtype():
    from oracle import oracle
    return oracle.KP_DTYPE


def test_features_in_area_semantics(oracle):
    """PosInGrid uses round(): a feature at x=639.9 rounds to cell 64 and is DROPPED from the grid
    (Frame.cpp:211-216); results come in (cell x, cell y, insertion) order; square window test."""
    xy = np.array([[100.0, 100.0], [104.0, 97.0], [95.0, 108.0], [639.9, 100.0], [100.0, 479.9], [121.0, 100.0]], np.float32)
    k = _kps(xy, np.array([0, 1, 2, 0, 0, 0]))
    got = oracle.features_in_area(k, 100.0, 100.0, 20.0, 0, 8)
    # cells: x=95->9 (9.5 rounds to 10? round(9.5)=10), 100->10, 104->10; order by cell x then y then index
    assert set(got.tolist()) == {0, 1, 2}
    assert 3 not in oracle.features_in_area(k, 635.0, 100.0, 20.0, 0, 8).tolist()     # dropped from the grid
    assert 4 not in oracle.features_in_area(k, 100.0, 475.0, 20.0, 0, 8).tolist()
    assert oracle.features_in_area(k, 100.0, 100.0, 20.0, 1, 1).tolist() == [1]        # bSameLevel
    assert 5 not in oracle.features_in_area(k, 100.0, 100.0, 20.0, 0, 8).tolist()      # |dx| = 21 > r
    assert 5 in oracle.features_in_area(k, 100.0, 100.0, 21.0, 0, 8).tolist()          # |dx| <= r inclusive
    # brute force equivalence on a real frame
    from se2lam_amd import synth
    kk, _ = oracle.orb_extract(synth.frame(0))
    rng = np.random.default_rng(2)
    for _ in range(50):
        x, y = rng.uniform(0, 640), rng.uniform(0, 480)
        lo = int(rng.integers(0, 6)); hi = lo + int(rng.integers(0, 3))
        got = oracle.features_in_area(kk, x, y, 20.0, lo, hi)
        px = np.round((kk["x"]) * np.float32(0.1)); py = np.round(kk["y"] * np.float32(0.1))
        ingrid = (px >= 0) & (px < 64) & (py >= 0) & (py < 48)
        want = np.nonzero(ingrid & (kk["octave"] >= lo) & (kk["octave"] <= hi) & (np.abs(kk["x"] - np.float32(x)) <= 20)
                          & (np.abs(kk["y"] - np.float32(y)) <= 20))[0]
        assert sorted(got.tolist()) == want.tolist()
        key = (px[got] * 48 + py[got]) * 100000 + got
        assert (np.diff(key) > 0).all()


def test_match_window_on_consecutive_frames(oracle, synth):
    k0, d0 = oracle.orb_extract(synth.frame(0))
    k1, d1 = oracle.orb_extract(synth.frame(1))
    m12, nm, prev = oracle.match_window(k0, d0, k1, d1)
    assert nm == (m12 >= 0).sum() and nm > 300
    good = m12 >= 0
    # frame 1 is frame 0 shifted by (-3, -1): matched key points move accordingly (up to level rounding)
    dx = k1["x"][m12[good]] - k0["x"][good]; dy = k1["y"][m12[good]] - k0["y"][good]
    assert np.median(dx) == pytest.approx(-3.0, abs=0.5) and np.median(dy) == pytest.approx(-1.0, abs=0.5)
    assert len(set(m12[good].tolist())) == good.sum()                      # one-to-one after eviction
    assert np.array_equal(prev[good], np.stack([k1["x"][m12[good]], k1["y"][m12[good]]], 1))
    assert np.array_equal(prev[~good], np.stack([k0["x"][~good], k0["y"][~good]], 1))
    # matching a frame against itself: every key point in the grid matches itself unless a duplicate descriptor
    m, n, _ = oracle.match_window(k0, d0, k0, d0)
    assert (m[m >= 0] == np.nonzero(m >= 0)[0]).mean() > 0.95


def test_compute_three_maxima_host_utility():
    """ORBmatcher::ComputeThreeMaxima (ORBmatcher.h:57, ORBmatcher.cpp:64-105) as the public member it is in the reference:
    se2gpu_three_maxima is the function the resolve kernels run, callable on the host (no device needed).  Against a
    sort-based restatement on random histograms with ties, and the 10 % rule."""
    import ctypes as C
    from se2lam_amd import capi
    rng = np.random.default_rng(5)

    def ref(h):
        order = sorted(range(len(h)), key=lambda i: (-h[i], i))      # strict '>' keeps the earliest of equal bins first
        top = [i for i in order[:3] if h[i] > 0] + [-1] * 3
        i1, i2, i3 = top[:3]
        m1 = h[i1] if i1 >= 0 else 0
        m2 = h[i2] if i2 >= 0 else 0
        m3 = h[i3] if i3 >= 0 else 0
        if m2 < np.float32(0.1) * np.float32(m1):
            i2 = i3 = -1
        elif m3 < np.float32(0.1) * np.float32(m1):
            i3 = -1
        return i1, i2, i3

    for trial in range(300):
        L = int(rng.integers(1, 31))
        h = rng.integers(0, 4 if trial % 3 else 60, L).astype(np.int32)
        if trial % 5 == 0:
            h[rng.integers(0, L)] = 500
        ind = [C.c_int(-1), C.c_int(-1), C.c_int(-1)]
        capi.check(capi.lib().se2gpu_three_maxima(h.ctypes.data, L, C.byref(ind[0]), C.byref(ind[1]), C.byref(ind[2])))
        assert tuple(i.value for i in ind) == ref(list(map(int, h))), (h, [i.value for i in ind])
