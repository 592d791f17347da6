#!/usr/bin/env python3
"""Writes tests/golden/oracle_r01.json: digests and a few raw numbers of the CPU oracle's outputs on the seeded synthetic
inputs.  The reference ships no golden vectors and cannot be built here, so these are NOT
reference outputs: they freeze the oracle (test infrastructure) so that an accidental change of the restatement - or a
GPU box whose oracle build differs - is caught, and they give the GPU tests a committed target next to the live oracle.

    python tools/gen_golden.py          (needs oracle/_build/liboracle.so: make -C oracle)
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import oracle  # noqa: E402
from se2lam_amd import synth  # noqa: E402


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def build():
    out = {"_about": "oracle outputs on se2lam_amd.synth inputs (seed 20190520 / 424242); NOT reference outputs"}
    # ORB: frames 0 and 1, FAST and Harris score
    for t in (0, 1):
        img = synth.frame(t)
        k, d = oracle.orb_extract(img)
        out[f"orb_frame{t}"] = {"n": int(len(k)), "sha256": digest(k, d), "first_kp": [float(v) for v in k[0].tolist()[:5]]}
    kh, dh = oracle.orb_extract(synth.frame(0), oracle.orb_params(score_type=oracle.HARRIS_SCORE))
    out["orb_frame0_harris"] = {"n": int(len(kh)), "sha256": digest(kh, dh), "first_response": float(kh["response"][0])}
    # MatchByWindow frame 0 -> 1 with the reference's call arguments (Track.cpp:131-132)
    k0, d0 = oracle.orb_extract(synth.frame(0))
    k1, d1 = oracle.orb_extract(synth.frame(1))
    m, n, prev = oracle.match_window(k0, d0, k1, d1, None, 20, 1, 0, 8, 0.9)
    out["match_window_0_1"] = {"nmatches": int(n), "sha256": digest(np.asarray(m, np.int32), prev)}
    # Track::removeOutliers on those matches (Track.cpp:134) with 60 of them corrupted
    rng = np.random.default_rng(7)
    mm = np.asarray(m, np.int32).copy()
    bad = rng.choice(np.flatnonzero(mm >= 0), 60, replace=False)
    mm[bad] = rng.integers(0, len(k1), 60)
    kept, ninl = oracle.remove_outliers(k0, k1, mm)
    out["remove_outliers_0_1"] = {"ninliers": int(ninl), "sha256": digest(kept)}
    # BA: config-3-shaped and tiny graphs, 10 LM iterations
    for P, L in ((8, 60), (50, 5000)):
        g = synth.ba_graph(P, L)
        p, l, st = oracle.ba_optimize(g, 10, 0)
        out[f"ba_{P}_{L}"] = {"E": int(g.E), "chi2_hist": [float(v) for v in st["chi2_hist"]],
                              "trials_hist": [int(v) for v in st["trials_hist"]],
                              "pose_last": [float(v) for v in p[-1]]}
    return out


if __name__ == "__main__":
    path = os.path.join(ROOT, "tests", "golden", "oracle_r01.json")
    json.dump(build(), open(path, "w"), indent=1)
    print("wrote", path)
