"""BASELINE.json configs[0] - "10 synthetic 640x480 frames + SE(2) odom through Track -> LocalMapper -> optimizer" - run through THE
REFERENCE's own Track / LocalMapper / Map / KeyFrame / MapPoint / Frame sources (compiled unmodified where they lie, oracle/Makefile
target `pipeline`), twice:

  * the CPU build: the reference's own ORBextractor.cpp / ORBmatcher.cpp, g2o's optimize() and cv::findFundamentalMat from the oracle;
  * the drop-in build: ORBextractor.cpp / ORBmatcher.cpp REPLACED by tests/dropin/ORBextractor_binding.cpp / ORBmatcher_binding.cpp (the bindings of
    INTEGRATION.md sections 1-2 as real files, through include/se2lam_amd/{ORBextractor,ORBmatcher,conversions}.h), optimize() and
    findFundamentalMat forwarded to libse2gpu (tests/dropin/g2o_forward.cpp, through include/se2lam_amd/optimizer.h).

"Drops into the existing pipeline unchanged" then means: the same key points and descriptors per frame, the same vnMatches12 out of
MatchByWindow, the same Track::mMatchIdx after the epipolar filter and the depth gate, the same key-frame decisions and map-point
counts, the key-frame poses after every localBA within 1e-5 - frame by frame, on the ten frames of configs[0] and on a longer run.
The CPU tests hold the CPU build to the restatement and to the committed fixture (tests/golden/pipeline_config0.npz, written by
tools/gen_golden_pipeline.py)."""
import os
import sys

import numpy as np
import pytest

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
sys.path.insert(0, os.path.join(ROOT, "tools"))
GOLD = os.path.join(HERE, "golden", "pipeline_config0.npz")
POSE_RTOL = 1e-5        # north_star: "BA cost and pose updates within 1e-5 relative"


def _pipeline():
    from oracle import pipeline
    return pipeline


def _need(kind):
    p = _pipeline()
    if not p.available(kind) and not p.can_build():
        pytest.skip("oracle/_ref/libse2lam_pipeline_%s.so is not built and /root/reference is not here" % kind)
    return p


def _fnv1a(data: bytes) -> int:
    # vectorised FNV-1a is not possible (sequential); 60 KB per frame is fine in pure Python for a handful of frames
    h = 1469598103934665603
    for b in data:
        h = ((h ^ b) * 1099511628211) & 0xFFFFFFFFFFFFFFFF
    return h


def compare_runs(a, b, what):
    """a = the CPU build's run, b = the other one: discrete results identical, poses / positions / costs within POSE_RTOL"""
    assert len(a["frames"]) == len(b["frames"])
    for t, (ra, rb) in enumerate(zip(a["frames"], b["frames"])):
        where = "%s, frame %d" % (what, t)
        for k in ("frame_id", "n_keypoints", "kp_hash", "desc_hash"):
            assert ra[k] == rb[k], (where, k, ra[k], rb[k])                        # the extractor: same key points, same descriptors
        assert np.array_equal(ra["raw_matches"], rb["raw_matches"]), where + ": vnMatches12 of MatchByWindow"
        assert ra["n_raw_matches"] == rb["n_raw_matches"], where
        assert np.array_equal(ra["match_idx"], rb["match_idx"]), where + ": Track::mMatchIdx (epipolar filter, depth gate)"
        for k in ("n_matches", "new_kf", "local_ba", "n_kfs", "n_mps", "n_good_prl", "n_local_kfs", "n_local_mps", "n_ref_kfs"):
            assert ra[k] == rb[k], (where, k, ra[k], rb[k])                        # key-frame decisions, map sizes
        assert np.allclose(ra["Twb"], rb["Twb"], rtol=POSE_RTOL, atol=1e-6), where
        if ra["local_ba"]:
            assert np.array_equal(ra["ba"][:4], rb["ba"][:4]), (where, "window sizes P, L, E, O", ra["ba"][:4], rb["ba"][:4])
            assert np.allclose(ra["ba"][4:6], rb["ba"][4:6], rtol=POSE_RTOL), (where, "cost at the start / end", ra["ba"][4:6], rb["ba"][4:6])
    assert len(a["after_ba"]) == len(b["after_ba"])
    for sa, sb in zip(a["after_ba"], b["after_ba"]):                              # the map after EACH localBA
        where = "%s, after the local BA of frame %d" % (what, sa["frame"])
        assert sa["frame"] == sb["frame"]
        assert np.array_equal(sa["kfs"]["id"], sb["kfs"]["id"]) and np.array_equal(sa["kfs"]["n_obs"], sb["kfs"]["n_obs"]), where
        scale = max(1.0, float(np.abs(sa["kfs"]["Twb"][:, :2]).max()))
        assert np.abs(sa["kfs"]["Twb"][:, :2] - sb["kfs"]["Twb"][:, :2]).max() <= POSE_RTOL * scale, where + ": key-frame positions"
        assert np.abs(sa["kfs"]["Twb"][:, 2] - sb["kfs"]["Twb"][:, 2]).max() <= POSE_RTOL, where + ": key-frame headings"
        assert np.array_equal(sa["mps"]["id"], sb["mps"]["id"]) and np.array_equal(sa["mps"]["n_obs"], sb["mps"]["n_obs"]), where
        assert np.array_equal(sa["mps"]["good_prl"], sb["mps"]["good_prl"]), where
        assert np.allclose(sa["mps"]["pos"], sb["mps"]["pos"], rtol=POSE_RTOL, atol=POSE_RTOL * Z_SCALE), where + ": map points"


Z_SCALE = 3000.0   # the scene's depth (mm): absolute tolerance of a coordinate near zero


# ------------------------------------------------------------------------------------------------------------------ CPU
def test_cpu_pipeline_is_the_reference_and_matches_the_restatement(oracle, synth, capfd):
    """configs[0] through the reference's own sources: the frames' key points are the restatement's, MatchByWindow's output is the
    restatement's, one key frame is inserted (frame 9: Track::nMinFrames = 8), its local BA runs and lowers the cost"""
    p = _need("cpu")
    n = 10
    frames, odo = synth.frames(n), p.odometry(n)
    res = p.run("cpu", frames, odo)
    capfd.readouterr()
    assert res["kind"] == "reference-cpu"
    calls = res["shim_calls"]
    assert calls["FAST"] > 0 and calls["resize"] == 7 * n and calls["GaussianBlur"] == 8 * n and calls["findFundamentalMat"] == n - 1
    k_prev = d_prev = None
    for t, r in enumerate(res["frames"]):
        k, d = oracle.orb_extract(frames[t])
        assert r["n_keypoints"] == len(k) and r["kp_hash"] == _fnv1a(k.tobytes()) and r["desc_hash"] == _fnv1a(d.tobytes()), t
        if t == 1:   # the reference frame is frame 0 and vbPrevMatched its key-point positions (Track::resetLocalTrack, src/Track.cpp:194)
            m, nm, _ = oracle.match_window(k_prev, d_prev, k, d)
            assert nm == r["n_raw_matches"] and np.array_equal(m, r["raw_matches"])
            want = np.ascontiguousarray(m, np.int32).copy()
            ninl = oracle.remove_outliers(k_prev, k, want)[1]
            assert np.array_equal(oracle.remove_outliers(k_prev, k, m)[0], r["match_idx"]) and ninl == r["n_matches"]
        if t == 0:
            k_prev, d_prev = k, d
    kf = [r["new_kf"] for r in res["frames"]]
    assert kf == [1, 0, 0, 0, 0, 0, 0, 0, 0, 1] and [r["local_ba"] for r in res["frames"]] == [0] * 9 + [1]
    last = res["frames"][-1]
    assert last["n_kfs"] == 2 and last["n_mps"] > 400 and last["ba"][0] == 2 and last["ba"][2] == 2 * last["ba"][1]
    assert last["ba"][5] < last["ba"][4]                                          # the local BA lowered the robust cost
    true = p.true_pose(9)
    assert np.allclose(res["kfs"]["Twb"][1, :2], true[:2], atol=8.0) and abs(res["kfs"]["Twb"][1, 2]) < 2e-3   # mm / rad: near the truth
    depth = res["mps"]["pos"][:, 2] - p.TBC[2]
    assert abs(np.median(depth) - p.Z0) < 0.05 * p.Z0                              # the plane the texture lies on, 3 m above the camera


def test_cpu_pipeline_reproduces_the_committed_fixture(synth, capfd):
    p = _need("cpu")
    import gen_golden_pipeline as g
    got = g.pack(p.run("cpu", synth.frames(g.NFRAMES), p.odometry(g.NFRAMES)))
    capfd.readouterr()
    gold = np.load(GOLD)
    assert sorted(got) == sorted(gold.files)
    for name in gold.files:
        if gold[name].dtype.kind == "f":
            assert np.allclose(got[name], gold[name], rtol=1e-6, atol=1e-6), name
        else:
            assert np.array_equal(got[name], gold[name]), name


def test_dropin_library_links_libse2gpu_and_no_oracle_code():
    """the drop-in build calls the C ABI (undefined se2gpu_* symbols resolved by libse2gpu.so) and contains none of the oracle's
    restatements; its ORBextractor / ORBmatcher members are the bindings' (the reference's two files are not in the link)"""
    p = _need("dropin")
    import subprocess
    p.build()
    syms = subprocess.run(["nm", "-D", p.LIBS["dropin"]], capture_output=True, text=True, check=True).stdout
    und = {l.split()[-1] for l in syms.splitlines() if " U " in l}
    for need in ("se2gpu_orb_extract", "se2gpu_match_window", "se2gpu_match_projection", "se2gpu_search_by_bow", "se2gpu_ba_optimize",
                 "se2gpu_ba_add_edge_se2xyz", "se2gpu_track_fundamental_mask"):
        assert need in und, need
    assert not [s for s in syms.split() if s.startswith(("ba_ref_", "match_ref_", "orb_ref_"))]
    cpu = subprocess.run(["nm", "-D", p.LIBS["cpu"]], capture_output=True, text=True, check=True).stdout
    assert "ba_ref_optimize" in cpu and not [l for l in cpu.splitlines() if " U se2gpu_" in l]


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
def test_config0_reference_pipeline_over_libse2gpu_equals_the_cpu_reference(synth, capfd):
    """THE drop-in test: configs[0] through the reference's own Track / LocalMapper with libse2gpu underneath, against the same
    sources on the CPU - and against the committed fixture of the CPU run"""
    p = _pipeline()
    n = 10
    frames, odo = synth.frames(n), p.odometry(n)
    gpu = p.run("dropin", frames, odo)
    assert gpu["kind"] == "dropin-gpu"
    calls = gpu["shim_calls"]     # the stand-in's image functions never ran: every pyramid / FAST / blur came from the device
    assert calls["FAST"] == 0 and calls["resize"] == 0 and calls["GaussianBlur"] == 0 and calls["findFundamentalMat"] == n - 1, calls
    import gen_golden_pipeline as g
    got, gold = g.pack(gpu), np.load(GOLD)
    for name in gold.files:
        if name.startswith(("kp_hash", "desc_hash", "match_idx_", "raw_matches_")) or name in g.SCALARS + ("kf_id", "kf_n_obs", "mp_id", "mp_n_obs", "mp_good"):
            assert np.array_equal(got[name], gold[name]), name
    assert np.allclose(got["kf_Twb"], gold["kf_Twb"], rtol=POSE_RTOL, atol=1e-5)
    assert np.allclose(got["mp_pos"], gold["mp_pos"], rtol=POSE_RTOL, atol=POSE_RTOL * Z_SCALE)
    assert np.allclose(got["ba"][-1][4:6], gold["ba"][-1][4:6], rtol=POSE_RTOL)
    if p.available("cpu"):
        cpu = p.run("cpu", frames, odo)
        compare_runs(cpu, gpu, "configs[0]")
    capfd.readouterr()


@pytest.mark.gpu
def test_longer_run_with_more_key_frames_equals_the_cpu_reference(synth, capfd):
    """120 frames with a key frame every 11 (Config::FPS = 10 -> Track::nMaxFrames): eleven local BAs, MatchByProjection against the
    local map, covisibility, Map::pruneRedundantKF removing key frames in between - the same decisions and the same map on both builds"""
    p = _pipeline()
    if not p.available("cpu"):
        pytest.skip("the CPU build of the pipeline did not travel")
    n = 120
    frames, odo = synth.frames(n), p.odometry(n)
    cfg = p.default_config()
    cfg.fps = 10
    cpu, gpu = p.run("cpu", frames, odo, cfg), p.run("dropin", frames, odo, cfg)
    capfd.readouterr()
    assert sum(r["local_ba"] for r in cpu["frames"]) >= 10 and cpu["frames"][-1]["n_kfs"] >= 4
    assert max(r["ba"][0] for r in cpu["frames"] if r["local_ba"]) >= 4                # windows of several key frames
    compare_runs(cpu, gpu, "120 frames")
