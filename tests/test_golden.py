"""Committed fixtures (tests/golden/oracle_r01.json, written by tools/gen_golden.py).

The reference ships no golden vectors and cannot be built in this image, so these are outputs of the CPU ORACLE on the
seeded synthetic inputs - they freeze the restatement (any accidental change of the oracle, or a GPU box whose oracle
build differs, shows up here) and give the HIP path a committed target next to the live oracle.  Parity with the real
OpenCV / g2o stays unpinned.
"""
import hashlib
import json
import os

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
GOLD = json.load(open(os.path.join(HERE, "golden", "oracle_r01.json")))


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


# ------------------------------------------------------------------------------------------- CPU: oracle == fixture
def test_oracle_orb_matches_fixture(oracle, synth):
    for t in (0, 1):
        k, d = oracle.orb_extract(synth.frame(t))
        assert len(k) == GOLD[f"orb_frame{t}"]["n"]
        assert digest(k, d) == GOLD[f"orb_frame{t}"]["sha256"]
    k, d = oracle.orb_extract(synth.frame(0), oracle.orb_params(score_type=oracle.HARRIS_SCORE))
    assert digest(k, d) == GOLD["orb_frame0_harris"]["sha256"]


def test_oracle_match_window_matches_fixture(oracle, synth):
    k0, d0 = oracle.orb_extract(synth.frame(0))
    k1, d1 = oracle.orb_extract(synth.frame(1))
    m, n, prev = oracle.match_window(k0, d0, k1, d1, None, 20, 1, 0, 8, 0.9)
    assert n == GOLD["match_window_0_1"]["nmatches"]
    assert digest(np.asarray(m, np.int32), prev) == GOLD["match_window_0_1"]["sha256"]


def _corrupted_matches(m, n2):
    rng = np.random.default_rng(7)
    mm = np.asarray(m, np.int32).copy()
    bad = rng.choice(np.flatnonzero(mm >= 0), 60, replace=False)
    mm[bad] = rng.integers(0, n2, 60)
    return mm


def test_oracle_remove_outliers_matches_fixture(oracle, synth):
    k0, d0 = oracle.orb_extract(synth.frame(0))
    k1, d1 = oracle.orb_extract(synth.frame(1))
    m, n, prev = oracle.match_window(k0, d0, k1, d1, None, 20, 1, 0, 8, 0.9)
    kept, ninl = oracle.remove_outliers(k0, k1, _corrupted_matches(m, len(k1)))
    assert ninl == GOLD["remove_outliers_0_1"]["ninliers"]
    assert digest(kept) == GOLD["remove_outliers_0_1"]["sha256"]


def test_oracle_ba_matches_fixture(oracle, synth):
    g = synth.ba_graph(8, 60)
    p, l, st = oracle.ba_optimize(g, 10, 0)
    gold = GOLD["ba_8_60"]
    assert g.E == gold["E"]
    assert list(st["trials_hist"]) == gold["trials_hist"]
    assert np.allclose(st["chi2_hist"], gold["chi2_hist"], rtol=1e-12, atol=0)   # same code, same compiler flags


# ------------------------------------------------------------------------------------------- GPU: HIP path == fixture
@pytest.mark.gpu
def test_hip_orb_and_match_equal_fixture(synth):
    from se2lam_amd.matcher import ORBmatcher
    from se2lam_amd.orb import ORBextractor, HARRIS_SCORE
    ex = ORBextractor()
    out = []
    for t in (0, 1):
        k, d = ex(synth.frame(t))
        assert digest(k, d) == GOLD[f"orb_frame{t}"]["sha256"]       # every key point field and descriptor byte
        out.append((k, d))
    k, d = ORBextractor(scoreType=HARRIS_SCORE)(synth.frame(0))
    assert digest(k, d) == GOLD["orb_frame0_harris"]["sha256"]
    (k0, d0), (k1, d1) = out
    prev = np.ascontiguousarray(np.stack([k0["x"], k0["y"]], axis=1), np.float32)   # Track.cpp:131-132 call arguments
    n, m = ORBmatcher(0.9).MatchByWindow(k0, d0, k1, d1, prev, 20)
    assert n == GOLD["match_window_0_1"]["nmatches"]
    assert digest(np.asarray(m, np.int32), prev) == GOLD["match_window_0_1"]["sha256"]
    from se2lam_amd.track import Track
    mm = _corrupted_matches(m, len(k1))
    ninl = Track().removeOutliers(k0, k1, mm)                                       # Track.cpp:134
    assert ninl == GOLD["remove_outliers_0_1"]["ninliers"]
    assert digest(mm) == GOLD["remove_outliers_0_1"]["sha256"]


@pytest.mark.gpu
@pytest.mark.parametrize("P,L", [(8, 60), (50, 5000)])
def test_hip_ba_equals_fixture(synth, P, L):
    """BA cost history within the north star's 1e-5 relative of the committed oracle run."""
    from se2lam_amd.optimizer import SlamOptimizer
    g = synth.ba_graph(P, L)
    o = SlamOptimizer()
    o.load(g)
    o.initializeOptimization(0)
    assert o.optimize(10) == 10
    gold = GOLD[f"ba_{P}_{L}"]
    assert o.stats["trials_hist"] == gold["trials_hist"]
    assert np.allclose(o.stats["chi2_hist"], gold["chi2_hist"], rtol=1e-5, atol=0)
    poses, _ = o.estimates()
    assert np.allclose(poses[-1], gold["pose_last"], rtol=1e-5, atol=1e-5)
