"""Committed fixtures of THE REFERENCE ITSELF (tests/golden/ref_r05.json + ref_r05.npz, written by tools/gen_golden_ref.py): outputs
of the reference's own source files, compiled unmodified in oracle/_ref, on the seeded synthetic inputs.  They travel with the
repository: the restatement (CPU) and the HIP path (GPU) are held to them here without /root/reference and without oracle/_ref;
where oracle/_ref is present, the first test also checks that the compiled reference still reproduces them."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = json.load(open(os.path.join(HERE, "golden", "ref_r05.json")))
ARR = np.load(os.path.join(HERE, "golden", "ref_r05.npz"))


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def _inputs():
    import gen_golden_ref
    return gen_golden_ref.inputs()


# ------------------------------------------------------------------------------------------------------------------ CPU
def test_compiled_reference_still_reproduces_the_fixtures(capfd):
    from oracle import ref
    if not ref.available():
        pytest.skip("oracle/_ref is not built and /root/reference is not here")
    import gen_golden_ref
    keep = (gen_golden_ref.OUT_JSON, gen_golden_ref.OUT_NPZ)
    try:
        import tempfile
        with tempfile.TemporaryDirectory() as tmp:
            gen_golden_ref.OUT_JSON = os.path.join(tmp, "r.json"); gen_golden_ref.OUT_NPZ = os.path.join(tmp, "r.npz")
            js, arr = gen_golden_ref.build()
    finally:
        gen_golden_ref.OUT_JSON, gen_golden_ref.OUT_NPZ = keep
    capfd.readouterr()
    def same(a, b, where):
        if isinstance(a, dict):
            assert sorted(a) == sorted(b), where
            for k in a:
                same(a[k], b[k], where + "/" + k)
        elif isinstance(a, float):
            # (sums over a std::set ordered by address - Localizer::DoLocalBA's map points - add up in another order from run to run)
            assert np.isclose(a, b, rtol=1e-12, atol=0), where
        else:
            assert a == b, where
    same(json.loads(json.dumps(js, sort_keys=True)), GOLD, "")
    assert sorted(arr) == sorted(ARR.files)
    for name in arr:
        assert np.allclose(arr[name], ARR[name], rtol=1e-12, atol=1e-12), name


def test_restatement_front_end_equals_the_reference_fixtures(oracle, synth):
    feats = {}
    for t in (0, 1):
        k, d = oracle.orb_extract(synth.frame(t))       # arrays as they come: the reference's order is part of the fixture
        feats[t] = (k, d)
        assert len(k) == GOLD[f"orb_frame{t}"]["n"] and digest(k, d) == GOLD[f"orb_frame{t}"]["sha256"]
    k, d = oracle.orb_extract(synth.frame(0), oracle.orb_params(score_type=oracle.HARRIS_SCORE))
    assert digest(k, d) == GOLD["orb_frame0_harris"]["sha256"]
    (k0, d0), (k1, d1) = feats[0], feats[1]
    m, n, prev = oracle.match_window(k0, d0, k1, d1)
    assert n == GOLD["match_window_0_1"]["nmatches"] and digest(m, prev) == GOLD["match_window_0_1"]["sha256"]


def test_restatement_edges_window_priors_and_sparsifier_equal_the_reference_fixtures(oracle, synth):
    inp = _inputs()
    g8 = inp["g8"]
    for i, s in enumerate(inp["states"]):
        e, Jp, Jl = oracle.ba_edge_se2xyz(g8, *s)
        for got, name in ((e, "e"), (Jp, "Jp"), (Jl, "Jl")):
            want = ARR["edge_se2xyz_" + name][i]
            assert np.abs(got - want).max() <= 1e-11 * max(1.0, np.abs(want).max())
    for i, s in enumerate(inp["odo"]):
        e, Ji, Jj = oracle.ba_edge_pre_se2(*s)
        assert np.allclose(e, ARR["edge_pre_se2_e"][i], rtol=1e-13, atol=1e-13) and np.allclose(Ji, ARR["edge_pre_se2_Ji"][i], rtol=1e-13, atol=1e-13)
        assert np.allclose(Jj, ARR["edge_pre_se2_Jj"][i], rtol=1e-13, atol=1e-13)
    assert np.isclose(oracle.ba_chi2(g8), GOLD["window_8_200"]["chi2"], rtol=1e-13, atol=0)
    assert GOLD["window_8_200"]["counts"] == [g8.E + g8.O, int(g8.fixed.sum()), g8.L]
    for i, T in enumerate(inp["poses"]):
        m, w = oracle.plane_motion_prior(T, inp["Tbc"])
        assert np.allclose(m, ARR["prior_expmap_meas"][i], rtol=0, atol=1e-9) and np.allclose(w, ARR["prior_expmap_info"][i], rtol=1e-9, atol=1e-3)
        m, w = oracle.pg_plane_motion_prior(np.linalg.inv(T), inp["Tbc"])
        assert np.allclose(m, ARR["prior_iso3_meas"][i], rtol=0, atol=1e-9) and np.allclose(w, ARR["prior_iso3_info"][i], rtol=1e-9, atol=1e-3)
    z, info, _ = oracle.sparsify(*synth.kf_pair(12, 0, 400.0))
    assert np.array_equal(z, ARR["sparsify_z"]) and np.abs(info - ARR["sparsify_info"]).max() <= 1e-5 * np.abs(info).max()


def test_restatement_triangulation_equals_the_reference_fixture(oracle):
    from test_ref_compiled import _triangulation_scene
    K, Tcr, k1, k2, match, has_obs, P1, P2, Ocam, X = _triangulation_scene(600, 7)
    pos, good, m, ng, nold = oracle.triangulate(k1, k2, match, has_obs, P1, P2, Ocam, 500.0, 8000.0, 2)
    g = GOLD["do_triangulate_600_7"]
    assert (ng, nold) == (g["n_good"], g["n_old"]) and digest(m, good) == g["sha256"]


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_hip_front_end_equals_the_reference_fixtures(synth):
    from se2lam_amd import orb
    from se2lam_amd.matcher import ORBmatcher
    ex = orb.ORBextractor()
    feats = {}
    for t in (0, 1):
        k, d = ex(synth.frame(t))
        feats[t] = (k, d)
        assert digest(k, d) == GOLD[f"orb_frame{t}"]["sha256"]
    k, d = orb.ORBextractor(1000, 1.2, 8, orb.HARRIS_SCORE, 20)(synth.frame(0))
    assert digest(k, d) == GOLD["orb_frame0_harris"]["sha256"]
    (k0, d0), (k1, d1) = feats[0], feats[1]
    prev = np.stack([k0["x"], k0["y"]], axis=1).astype(np.float32).copy()
    nm, m12 = ORBmatcher(0.9).MatchByWindow(k0, d0, k1, d1, prev, 20)
    assert nm == GOLD["match_window_0_1"]["nmatches"] and digest(m12, prev) == GOLD["match_window_0_1"]["sha256"]


@pytest.mark.gpu
def test_hip_window_cost_priors_and_sparsifier_equal_the_reference_fixtures(synth):
    from se2lam_amd import optimizer as op
    from se2lam_amd.localizer import addPlaneMotionSE3Expmap
    from se2lam_amd.sparsifier import DoMarginalizeSE3XYZ_batch
    inp = _inputs()
    o = op.SlamOptimizer()
    o.load(inp["g8"])
    o.initializeOptimization()
    assert np.isclose(o.activeRobustChi2(), GOLD["window_8_200"]["chi2"], rtol=1e-12, atol=0)
    for i, T in enumerate(inp["poses"]):
        m, w = addPlaneMotionSE3Expmap(T, inp["Tbc"])
        assert np.allclose(m, ARR["prior_expmap_meas"][i], rtol=0, atol=1e-9) and np.allclose(w, ARR["prior_expmap_info"][i], rtol=1e-9, atol=1e-3)
    (z, info), = DoMarginalizeSE3XYZ_batch([synth.kf_pair(12, 0, 400.0)])
    assert np.allclose(z, ARR["sparsify_z"], atol=1e-12) and np.abs(info - ARR["sparsify_info"]).max() <= 1e-5 * np.abs(info).max()
