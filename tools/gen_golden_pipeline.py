"""Writes tests/golden/pipeline_config0.npz: what THE REFERENCE's own pipeline (oracle/_ref/libse2lam_pipeline_cpu.so - every source file
of /root/reference compiled where it lies, Track -> LocalMapper -> optimizer fed frame by frame) does on BASELINE.json configs[0],
ten synthetic 640x480 frames + SE(2) odometry.  A fixture is data: per frame the key-point / descriptor digests, MatchByWindow's
vnMatches12, Track::mMatchIdx after the epipolar filter and the depth gate, the key-frame decision, map sizes; after the local BA the
key-frame poses and map points.  Run where /root/reference exists:  python tools/gen_golden_pipeline.py"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
OUT = os.path.join(ROOT, "tests", "golden", "pipeline_config0.npz")
NFRAMES = 10
SCALARS = ("frame_id", "n_keypoints", "n_raw_matches", "n_matches", "new_kf", "local_ba", "n_kfs", "n_mps", "n_good_prl",
           "n_local_kfs", "n_local_mps", "n_ref_kfs")


def pack(res) -> dict:
    """the comparable content of oracle.pipeline.run(...) as flat arrays"""
    fr = res["frames"]
    out = {k: np.array([r[k] for r in fr], np.int64) for k in SCALARS}
    out["kp_hash"] = np.array([r["kp_hash"] for r in fr], np.uint64)
    out["desc_hash"] = np.array([r["desc_hash"] for r in fr], np.uint64)
    for t, r in enumerate(fr):
        out["match_idx_%d" % t] = r["match_idx"]
        out["raw_matches_%d" % t] = r["raw_matches"]
    out["frame_Twb"] = np.stack([r["Twb"] for r in fr])
    out["ba"] = np.stack([r["ba"] for r in fr])
    out["kf_id"], out["kf_Twb"], out["kf_Tcw"], out["kf_n_obs"] = res["kfs"]["id"], res["kfs"]["Twb"], res["kfs"]["Tcw"], res["kfs"]["n_obs"]
    out["mp_id"], out["mp_pos"], out["mp_n_obs"], out["mp_good"] = res["mps"]["id"], res["mps"]["pos"], res["mps"]["n_obs"], res["mps"]["good_prl"]
    return out


def build():
    from oracle import pipeline
    from se2lam_amd import synth
    return pack(pipeline.run("cpu", synth.frames(NFRAMES), pipeline.odometry(NFRAMES)))


if __name__ == "__main__":
    arr = build()
    np.savez_compressed(OUT, **arr)
    print("wrote", OUT, os.path.getsize(OUT), "bytes;", int(arr["n_kfs"][-1]), "key frames,", int(arr["n_mps"][-1]), "map points")
