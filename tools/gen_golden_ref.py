#!/usr/bin/env python3
"""Writes tests/golden/ref_r05.json and tests/golden/ref_r05.npz: outputs of THE REFERENCE ITSELF - its own source files compiled
unmodified in oracle/_ref (oracle/Makefile, target `ref`) - on the seeded synthetic inputs of se2lam_amd.synth.  They travel
with the repository, so that the restatement (CPU) and the HIP path (GPU) can be held to reference-derived numbers on a
machine where /root/reference and oracle/_ref are absent.  tests/test_golden_ref.py reads them; where oracle/_ref is
present it also checks that the compiled reference still reproduces them.

    python tools/gen_golden_ref.py          (needs /root/reference: builds oracle/_ref first)

Exact quantities (key points, descriptors, match lists, counters) are stored as SHA-256 digests, floating-point ones as arrays.
"""
import hashlib
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle, ref  # noqa: E402
from se2lam_amd import synth  # noqa: E402

OUT_JSON = os.path.join(ROOT, "tests", "golden", "ref_r05.json")
OUT_NPZ = os.path.join(ROOT, "tests", "golden", "ref_r05.npz")


def digest(*arrays):
    h = hashlib.sha256()
    for a in arrays:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def inputs():
    """Everything the fixtures are computed from, rebuilt identically by the tests."""
    rng = np.random.default_rng(20260926)
    g8 = synth.ba_graph(8, 200, seed=5)
    states = []
    for _ in range(5):
        k = int(rng.integers(0, g8.E))
        kf, lm = int(g8.e_kf[k]), int(g8.e_lm[k])
        states.append((g8.poses[kf] + rng.normal(0, [5.0, 5.0, 0.01]), g8.lms[lm] + rng.normal(0, 10.0, 3), g8.e_uv[k]))
    odo = [(g8.poses[int(g8.o_i[k])] + rng.normal(0, [3.0, 3.0, 0.01]), g8.poses[int(g8.o_j[k])], g8.o_meas[k]) for k in range(5)]
    Tbc = np.eye(4); Tbc[:3, :3] = synth.RBC; Tbc[:3, 3] = synth.TBC
    poses = []
    for _ in range(3):
        Tcw = synth.se2_to_Tcw(np.array([rng.uniform(-3000, 3000), rng.uniform(-3000, 3000), rng.uniform(-3.1, 3.1)]))
        poses.append(synth.se3_exp_np(np.concatenate([rng.normal(0, 0.02, 3), rng.normal(0, 8.0, 3)])) @ Tcw)
    return dict(g8=g8, states=states, odo=odo, Tbc=Tbc, poses=poses)


def build():
    js = {"_about": "outputs of the reference's own sources compiled in oracle/_ref on se2lam_amd.synth inputs (tools/gen_golden_ref.py)"}
    arr = {}
    # ---- front end (exact)
    feats = {}
    for t in (0, 1):
        k, d = ref.orb_extract(synth.frame(t))   # in the reference's own order (round 5: no sorting before the digest)
        feats[t] = (k, d)
        js[f"orb_frame{t}"] = {"n": int(len(k)), "sha256": digest(k, d)}
    k, d = ref.orb_extract(synth.frame(0), oracle.orb_params(score_type=oracle.HARRIS_SCORE))
    js["orb_frame0_harris"] = {"n": int(len(k)), "sha256": digest(k, d)}
    (k0, d0), (k1, d1) = feats[0], feats[1]
    m, n, prev = ref.match_window(k0, d0, k1, d1)
    js["match_window_0_1"] = {"nmatches": int(n), "sha256": digest(m, prev)}
    # ---- the SE(2)-XYZ edges and a whole window
    inp = inputs()
    g8 = inp["g8"]
    e = [ref.edge_se2xyz(g8, *s) for s in inp["states"]]
    arr["edge_se2xyz_e"] = np.stack([x[0] for x in e]); arr["edge_se2xyz_Jp"] = np.stack([x[1] for x in e]); arr["edge_se2xyz_Jl"] = np.stack([x[2] for x in e])
    o = [ref.edge_pre_se2(*s) for s in inp["odo"]]
    arr["edge_pre_se2_e"] = np.stack([x[0] for x in o]); arr["edge_pre_se2_Ji"] = np.stack([x[1] for x in o]); arr["edge_pre_se2_Jj"] = np.stack([x[2] for x in o])
    total, chi_e, chi_o, counts = ref.window_chi2(g8)
    js["window_8_200"] = {"chi2": total, "counts": list(counts)}
    arr["window_chi_e"] = chi_e[:40].copy(); arr["window_chi_o"] = chi_o.copy()
    # ---- plane-motion priors
    pm = [ref.plane_motion_prior(T, inp["Tbc"])[:2] for T in inp["poses"]]
    pg = [ref.pg_plane_motion_prior(np.linalg.inv(T), inp["Tbc"])[:2] for T in inp["poses"]]
    arr["prior_expmap_meas"] = np.stack([x[0] for x in pm]); arr["prior_expmap_info"] = np.stack([x[1] for x in pm])
    arr["prior_iso3_meas"] = np.stack([x[0] for x in pg]); arr["prior_iso3_info"] = np.stack([x[1] for x in pg])
    # ---- sparsifier
    z, info = ref.sparsify(*synth.kf_pair(12, 0, 400.0))
    arr["sparsify_z"] = z; arr["sparsify_info"] = info
    # ---- Map::loadLocalGraph (both variants), GlobalBA, DoLocalBA, doTriangulate, the vocabulary
    import test_ref_compiled as T
    for n_ref in (0, 3):
        m_, w = T._reference_window(synth, n_ref)
        m_.update_local_graph(0)
        out = m_.load_local_graph()
        js[f"load_local_graph_ref{n_ref}"] = {"chi2": out["chi2"], "vertices": int(len(out["v_id"])), "edges": int(len(out["e_ids"])),
                                              "fixed": [int(i) for i in np.nonzero(out["v_fixed"])[0]]}
        order = np.lexsort((out["e_ids"][:, 0], out["e_ids"][:, 1]))
        arr[f"load_local_graph_ref{n_ref}_info"] = out["e_info"][order][:60].copy()
        arr[f"load_local_graph_ref{n_ref}_ids"] = out["e_ids"][order][:60].copy()
        _, _, out3, g3 = T._se3_window(synth, n_ref)
        js[f"se3_local_graph_ref{n_ref}"] = {"chi2": out3["chi2"], "priors": int(len(out3["p_id"])), "edges": int(len(out3["e_ids"]))}
    mg, wg = T._global_map(synth)
    og = mg.global_ba()
    js["global_ba_14"] = {"chi2": og["chi2"], "edges": [[int(a), int(b)] for a, b in og["e_ids"].tolist()]}
    arr["global_ba_e_chi2"] = og["e_chi2"].copy()
    Tcw0, Xw, kps, TBC, F, CX, CY, DELTA = T._pose_only_case(0, 400)
    K = np.array([[F, 0, CX], [0, F, CY], [0, 0, 1]], np.float32)
    good = np.ones(400, np.uint8); good[::9] = 0
    ol = ref.localizer_do_local_ba(K, TBC, np.float32(DELTA), Tcw0, kps, Xw, good)
    js["do_local_ba_0_400"] = {"chi2": ol["chi2"], "edges": ol["n_edges"], "fixed": ol["n_fixed"]}
    Kt, Tcr, k1t, k2t, match, has_obs, P1, P2, Ocam, X = T._triangulation_scene(600, 7)
    pos, goodt, mt, ng, nold = ref.track_triangulate(Kt, k1t, k2t, match, has_obs, X, Tcr, 500.0, 8000.0)
    js["do_triangulate_600_7"] = {"n_good": ng, "n_old": nold, "sha256": digest(mt, goodt)}
    arr["do_triangulate_pos"] = pos[:50].copy()
    with open(OUT_JSON, "w") as f:
        json.dump(js, f, indent=1, sort_keys=True)
    np.savez_compressed(OUT_NPZ, **arr)
    return js, arr


if __name__ == "__main__":
    js, arr = build()
    print("wrote", OUT_JSON, os.path.getsize(OUT_JSON), "bytes;", OUT_NPZ, os.path.getsize(OUT_NPZ), "bytes;", len(js) - 1, "records,", len(arr), "arrays")
