"""Randomised sweep, CPU only: the compiled reference's BACK END (oracle/_ref: optimizer.cpp, sparsifier.cpp, Map.cpp / KeyFrame.cpp /
MapPoint.cpp) against the restatement and the host-side product code, on random inputs.
  map      Map::updateLocalGraph on random maps                      vs se2gpu_map_update_local_graph (CSR view, host code)
  graph    Map::loadLocalGraph on random windows (with / without reference key frames, random odometry gaps, random
           extrinsic, random plane-motion informations)              vs oracle.ba_edge_information / ba_chi2
  prior    addPlaneMotionSE3Expmap / addVertexSE3PlaneMotion          vs oracle.plane_motion_prior / pg_plane_motion_prior
  sparsify Sparsifier::DoMarginalizeSE3XYZ                            vs oracle.sparsify (relative pose exact, InfoSE3 on the same H)
  tri      Track::doTriangulate on random two-view scenes             vs oracle.triangulate (matches, flags, counters exact)
  poseba   Localizer::DoLocalBA's graph at optimize()                 vs oracle.pose_only_ba (cost at the start)
  pg       GlobalMapper::GlobalBA's graph at optimize()               vs oracle.pg_chi2 (cost and per-edge chi2)
usage: python tools/fuzz_ref_backend.py [seconds]"""
import dataclasses
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from oracle import oracle, ref  # noqa: E402
from se2lam_amd import synth  # noqa: E402
from se2lam_amd.mapview import updateLocalGraph  # noqa: E402
from test_mapview import _random_map  # noqa: E402


def fuzz_map(rng):
    K = int(rng.integers(2, 70)); M = int(rng.integers(0, 700)); reach = int(rng.integers(1, 6))
    kf_id, covisible, kf_obs, mp_id, mp_obs = _random_map(rng, K, M, reach)
    m = ref.RefMap(np.eye(3), np.eye(4), 2.0)
    for a in range(K):
        m.add_kf(int(kf_id[a]), a, [0.0, 0.0, 0.0], np.zeros((len(kf_obs[a]), 2)))
    for j in range(M):
        m.add_mp(int(mp_id[j]), [0.0, 0.0, 1000.0])
    for a in range(K):
        for f, j in enumerate(kf_obs[a]):
            m.observe(a, j, f)
    for a in range(K):
        for b in covisible[a]:
            m.covisible(a, b)
    for cur in rng.integers(0, K, 3):
        lk_r, rk_r, lm_r = m.update_local_graph(int(cur))
        lk, rk, lm = updateLocalGraph(kf_id, covisible, kf_obs, mp_id, mp_obs, int(cur), 3)
        assert kf_id[lk].tolist() == lk_r.tolist() and kf_id[rk].tolist() == rk_r.tolist() and mp_id[lm].tolist() == lm_r.tolist(), ("map", K, M, reach, int(cur))


def _rot(rng, sigma):
    return synth.se3_exp_np(np.concatenate([rng.normal(0, sigma, 3), np.zeros(3)]))[:3, :3]


def fuzz_graph(rng):
    P = int(rng.integers(4, 16)); L = int(rng.integers(20, 260))
    g = synth.ba_graph(P, L, seed=int(rng.integers(0, 1 << 30)))
    n_ref = int(rng.integers(0, max(1, P // 3)))
    nL = P - n_ref
    xrot, yrot, zinf = float(10 ** rng.uniform(4, 7)), float(10 ** rng.uniform(4, 7)), float(10 ** rng.uniform(-1, 1))
    # a random extrinsic close to the generator's, exactly representable in float32 (Config::bTc is CV_32F)
    bTc = np.eye(4); bTc[:3, :3] = g.Rbc @ _rot(rng, 0.05); bTc[:3, 3] = g.tbc + rng.normal(0, 20.0, 3)
    bTc = bTc.astype(np.float32).astype(np.float64)
    K = np.array([[g.fx, 0, g.cx], [0, g.fx, g.cy], [0, 0, 1]], np.float32)
    huber = np.float32(rng.uniform(1.5, 4.0))
    m = ref.RefMap(K, bTc, huber, xrot, yrot, zinf)
    twb = g.poses.astype(np.float32)
    level = rng.integers(0, 8, g.E)
    sf = np.ones(8, np.float32)
    for i in range(1, 8):
        sf[i] = sf[i - 1] * np.float32(1.2)
    sigma2 = (sf * sf)[level]
    uv32 = g.e_uv.astype(np.float32)
    frame_id = rng.permutation(np.arange(2, 2 + 3 * P))[:P]            # Frame::id (1 would be fixed by the second clause of the rule)
    if rng.random() < 0.3:
        frame_id[int(rng.integers(0, nL))] = 1
    ftr = np.zeros(g.E, int); cnt = np.zeros(P, int)
    for k in range(g.E):
        ftr[k] = cnt[g.e_kf[k]]; cnt[g.e_kf[k]] += 1
    # camera-frame points as KeyFrame::mViewMPs holds them: from the reference's own float pose, computed after the key frame exists
    for a in range(P):
        sel = np.nonzero(g.e_kf == a)[0]
        m.add_kf(10 + a, int(frame_id[a]), twb[a], uv32[sel], level[sel], np.tile([0.0, 0.0, 1.0], (len(sel), 1)))
    Tcw = np.stack([m.kf_pose(a) for a in range(P)]).astype(np.float64)
    lw32 = g.lms.astype(np.float32)
    lc = np.einsum("eij,ej->ei", Tcw[g.e_kf][:, :3, :3], lw32[g.e_lm].astype(np.float64)) + Tcw[g.e_kf][:, :3, 3]
    lc32 = lc.astype(np.float32)
    m2 = ref.RefMap(K, bTc, huber, xrot, yrot, zinf)                    # again, now with the camera-frame points in place
    for a in range(P):
        sel = np.nonzero(g.e_kf == a)[0]
        m2.add_kf(10 + a, int(frame_id[a]), twb[a], uv32[sel], level[sel], lc32[sel])
    for l in range(g.L):
        m2.add_mp(1000 + l, lw32[l])
    for k in range(g.E):
        m2.observe(int(g.e_kf[k]), int(g.e_lm[k]), int(ftr[k]))
    for a in range(1, nL):
        m2.covisible(0, a); m2.covisible(a, 0)
    odo = {}
    for k in range(g.O):
        if rng.random() < 0.15:
            continue                                                     # a gap in the odometry chain
        i, j = int(g.o_i[k]), int(g.o_j[k])
        odo[i] = (j, g.o_meas[k], np.linalg.inv(g.o_info[k].reshape(3, 3)))
        m2.set_odo(i, j, odo[i][1], odo[i][2])
    lk, rk, lm = m2.update_local_graph(0)
    out = m2.load_local_graph()
    local_mp = sorted(set(int(l) for k, l in zip(g.e_kf, g.e_lm) if k < nL))
    observers_outside = sorted(set(int(k) for k, l in zip(g.e_kf, g.e_lm) if k >= nL and int(l) in set(local_mp)))
    assert lk.tolist() == [10 + a for a in range(nL)] and rk.tolist() == [10 + a for a in observers_outside], ("graph lists", P, n_ref)
    kf_list = list(range(nL)) + observers_outside                         # vertex id -> key frame
    nK = len(kf_list)
    maxKFid = nK + 1
    vtx_of_kf = {a: i for i, a in enumerate(kf_list)}
    assert out["v_id"].tolist() == list(range(nK)) + [maxKFid + i for i in range(len(local_mp))]
    fixed = np.zeros(nK, bool)
    fid = frame_id[kf_list]
    if not observers_outside:
        fixed[:nL] |= fid[:nL] == fid[:nL].min()
    fixed[:nL] |= fid[:nL] == 1
    fixed[nL:] = True
    assert np.array_equal(out["v_fixed"][:nK], fixed), ("fixed rule", fid.tolist(), out["v_fixed"][:nK].tolist())
    assert np.array_equal(out["v_est"][:nK], twb[kf_list].astype(np.float64))
    want_odo = [(i, j) for i, (j, _, _) in sorted(odo.items()) if i < nL and j < nL]
    assert [tuple(x) for x in out["o_ids"].tolist()] == want_odo
    for (i, j), info in zip(want_odo, out["o_info"]):
        assert np.allclose(info, np.linalg.inv(odo[i][2]), rtol=1e-10, atol=0)
    # informations: the restatement on the same float inputs (Rcw from the reference's own poses, body frame = its Twb)
    Rcw = Tcw[:, :3, :3].astype(np.float32).reshape(P, 9)
    info = oracle.ba_edge_information(lc=lc32, lw=lw32[g.e_lm], e_kf=g.e_kf, sigma2=sigma2, Rcw=Rcw, twb_xy=twb[:, :2], fx=np.float32(g.fx),
                                      xrot_info=np.float32(xrot), z_info=np.float32(zinf))
    pos_of = {l: i for i, l in enumerate(local_mp)}
    in_graph = np.array([int(l) in pos_of and int(k) in vtx_of_kf for k, l in zip(g.e_kf, g.e_lm)])
    edge_of = {(vtx_of_kf[int(k)], maxKFid + pos_of[int(l)]): e for e, (k, l) in enumerate(zip(g.e_kf, g.e_lm)) if in_graph[e]}
    assert len(out["e_ids"]) == len(edge_of)
    for ids, uv, W, delta in zip(out["e_ids"].tolist(), out["e_uv"], out["e_info"], out["e_delta"]):
        e = edge_of[tuple(ids)]
        assert np.array_equal(uv, uv32[e].astype(np.float64)) and delta == float(huber)
        assert np.allclose(W, info[e], rtol=1e-9, atol=0), ("information", e, W, info[e])
    # the robust cost through the restatement (its camera: Tcb = bTc^-1 in double of the float extrinsic)
    remap_kf = np.full(P, -1); remap_kf[kf_list] = np.arange(nK)
    remap_lm = np.full(g.L, -1); remap_lm[local_mp] = np.arange(len(local_mp))
    g2 = dataclasses.replace(
        g, poses=twb[kf_list].astype(np.float64), fixed=fixed.astype(np.uint8), lms=lw32[local_mp].astype(np.float64),
        e_kf=remap_kf[g.e_kf[in_graph]].astype(np.int32), e_lm=remap_lm[g.e_lm[in_graph]].astype(np.int32), e_uv=uv32[in_graph].astype(np.float64),
        e_info=np.stack([info[in_graph][:, 0, 0], info[in_graph][:, 0, 1], info[in_graph][:, 1, 1]], axis=1),
        o_i=np.array([i for i, _ in want_odo], np.int32), o_j=np.array([j for _, j in want_odo], np.int32),
        o_meas=np.array([odo[i][1] for i, _ in want_odo]).reshape(-1, 3), o_info=np.array([np.linalg.inv(odo[i][2]).reshape(-1) for i, _ in want_odo]).reshape(-1, 9),
        Rbc=_requat_rot(bTc[:3, :3]), tbc=bTc[:3, 3].copy(), huber=float(huber), poses_true=None, lms_true=None)
    assert np.isclose(out["chi2"], oracle.ba_chi2(g2), rtol=1e-9, atol=0), ("chi2", out["chi2"], oracle.ba_chi2(g2))


def _requat_rot(R):
    """what toSE3Quat(cv::Mat) makes of a float32-rounded rotation: the matrix of its normalised quaternion"""
    from scipy.spatial.transform import Rotation
    t = np.trace(R)
    if t > 0:
        s = np.sqrt(t + 1.0); w = 0.5 * s; s = 0.5 / s
        q = [(R[2, 1] - R[1, 2]) * s, (R[0, 2] - R[2, 0]) * s, (R[1, 0] - R[0, 1]) * s, w]
    else:
        i = int(np.argmax(np.diag(R))); j = (i + 1) % 3; k = (j + 1) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        v = np.zeros(3); v[i] = 0.5 * s; s = 0.5 / s
        w = (R[k, j] - R[j, k]) * s; v[j] = (R[j, i] + R[i, j]) * s; v[k] = (R[k, i] + R[i, k]) * s
        q = [v[0], v[1], v[2], w]
    return Rotation.from_quat(np.array(q)).as_matrix()


def fuzz_prior(rng):
    Tbc = np.eye(4); Tbc[:3, :3] = synth.RBC @ _rot(rng, 0.1); Tbc[:3, 3] = synth.TBC + rng.normal(0, 30.0, 3)
    Tbc = Tbc.astype(np.float32).astype(np.float64)
    Tq = Tbc.copy(); Tq[:3, :3] = _requat_rot(Tbc[:3, :3])
    Tcw = synth.se2_to_Tcw(np.array([rng.uniform(-5000, 5000), rng.uniform(-5000, 5000), rng.uniform(-3.1, 3.1)]))
    T = synth.se3_exp_np(np.concatenate([rng.normal(0, 0.03, 3), rng.normal(0, 10.0, 3)])) @ Tcw
    xr, yr, z = float(10 ** rng.uniform(3, 7)), float(10 ** rng.uniform(3, 7)), float(10 ** rng.uniform(-2, 2))
    xr, yr, z = float(np.float32(xr)), float(np.float32(yr)), float(np.float32(z))     # Config holds floats
    m, w, _ = ref.plane_motion_prior(T, Tbc, xr, yr, z)
    mo, wo = oracle.plane_motion_prior(T, Tq, xr, yr, z)
    assert np.allclose(m, mo, rtol=0, atol=1e-8) and np.allclose(w, wo, rtol=1e-8, atol=1e-8 * np.abs(wo).max()), "prior expmap"
    Twc = np.linalg.inv(T)
    m, w, _ = ref.pg_plane_motion_prior(Twc, Tbc, xr, yr, z)
    mo, wo = oracle.pg_plane_motion_prior(Twc, Tq, xr, yr, z)
    assert np.allclose(m, mo, rtol=0, atol=1e-8) and np.allclose(w, wo, rtol=1e-8, atol=1e-8 * np.abs(wo).max()), "prior iso3"


def fuzz_sparsify(rng):
    N = int(rng.integers(6, 120))
    kf, mp, m_kf, m_mp, m_info = synth.kf_pair(N, int(rng.integers(0, 1 << 30)), float(rng.uniform(80, 900)))
    z, info = ref.sparsify(kf, mp, m_kf, m_mp, m_info)
    zo, io, Hm = oracle.sparsify(kf, mp, m_kf, m_mp, m_info)
    assert np.array_equal(z, zo), "sparsify pose"
    assert np.abs(ref.sparsify_info_se3(kf, Hm) - io).max() <= 1e-11 * np.abs(io).max(), "InfoSE3"
    rel = np.abs(info - io).max() / np.abs(io).max()
    return rel


KP = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])


def fuzz_tri(rng):
    n = int(rng.integers(1, 400))
    K = np.array([[400.0, 0, 320.0], [0, 400.0, 240.0], [0, 0, 1]], np.float32)
    Tcr = synth.se3_exp_np(np.concatenate([rng.normal(0, 0.03, 3), rng.normal(0, 120.0, 3)])).astype(np.float32)
    P1 = (K @ np.eye(3, 4, dtype=np.float32)).astype(np.float32); P2 = (K @ Tcr[:3]).astype(np.float32)
    X = np.stack([rng.uniform(-1500, 1500, n), rng.uniform(-800, 800, n), rng.uniform(200, 14000, n)], 1).astype(np.float32)
    Xh = np.concatenate([X, np.ones((n, 1), np.float32)], 1)
    u1 = (P1 @ Xh.T).T; u1 = u1[:, :2] / u1[:, 2:]
    u2 = (P2 @ Xh.T).T; u2 = u2[:, :2] / u2[:, 2:]
    k1 = np.zeros(n, KP); k2 = np.zeros(n, KP)
    k1["x"], k1["y"] = u1[:, 0], u1[:, 1]
    perm = rng.permutation(n)
    k2["x"][perm], k2["y"][perm] = u2[:, 0] + rng.normal(0, 0.3, n), u2[:, 1] + rng.normal(0, 0.3, n)
    match = perm.astype(np.int32); match[rng.random(n) < 0.1] = -1
    has_obs = (rng.random(n) < 0.15).astype(np.uint8)
    Ocam = np.linalg.inv(Tcr.astype(np.float64))[:3, 3].astype(np.float32)
    lower, upper = float(rng.uniform(200, 800)), float(rng.uniform(4000, 12000))
    pos, good, m, ng, nold = ref.track_triangulate(K, k1, k2, match, has_obs, X, Tcr, lower, upper)
    po, go, mo, ngo, noldo = oracle.triangulate(k1, k2, match, has_obs, P1, P2, Ocam, lower, upper, 2)
    # a point whose depth or parallax sits on a gate may fall either way between two SVD routines: compare away from the gates
    acc = (match >= 0) & (has_obs == 0) & (m >= 0) & (mo >= 0)
    diff = (m != mo) | (good != go)
    if diff.any():
        z = np.where(mo >= 0, po[:, 2], pos[:, 2])
        p1 = np.where((mo >= 0)[:, None], po, pos).astype(np.float64); p2 = p1 - Ocam
        cosp = np.abs((p1 * p2).sum(1)) / (np.linalg.norm(p1, axis=1) * np.linalg.norm(p2, axis=1) + 1e-30)
        near = (np.abs(z - lower) < 1e-3 * lower) | (np.abs(z - upper) < 1e-3 * upper) | (np.abs(cosp - 0.9994) < 1e-6)      # cvu::checkParallax: cos < 0.9994 for 2 degrees
        if not near[diff].all():
            i = int(np.nonzero(diff & ~near)[0][0])
            raise AssertionError(("tri", n, i, "ref", pos[i].tolist(), int(m[i]), int(good[i]), "oracle", po[i].tolist(), int(mo[i]), int(go[i]),
                                  "depth gates", lower, upper, "cos parallax", float(cosp[i]), "true", X[i].tolist(), "has_obs", int(has_obs[i])))
    assert nold == noldo
    # positions: the DLT of a point without parallax is ill-conditioned, and the smallest singular vector of two SVD routines (the
    # stand-in's reading of cv::SVD in float, the restatement's FP64 Jacobi) then differs visibly; compare the well-posed ones
    well = acc & (good == 1) & (go == 1)
    if well.any():
        rel = np.abs(pos[well] - po[well]).max(axis=1) / np.abs(po[well]).max(axis=1)
        assert rel.max() <= 2e-3, ("tri positions", float(rel.max()))


def fuzz_poseba(rng):
    n = int(rng.integers(4, 300))
    F_, CX_, CY_ = 400.0, 320.0, 240.0
    K = np.array([[F_, 0, CX_], [0, F_, CY_], [0, 0, 1]], np.float32)
    TBC = np.eye(4); TBC[:3, :3] = synth.RBC; TBC[:3, 3] = synth.TBC
    pose = np.array([rng.uniform(-4000, 4000), rng.uniform(-4000, 4000), rng.uniform(-3.14, 3.14)])
    Tcw = (synth.se3_exp_np(np.concatenate([rng.normal(0, 0.01, 3), rng.normal(0, 5.0, 3)])) @ synth.se2_to_Tcw(pose)).astype(np.float32)
    Xc = np.stack([rng.uniform(-2000, 2000, n), rng.uniform(-1500, 1500, n), rng.uniform(1500, 8000, n)], 1)
    Xw = ((np.linalg.inv(Tcw.astype(np.float64)) @ np.c_[Xc, np.ones(n)].T).T[:, :3]).astype(np.float32)
    kps = np.zeros(n, KP)
    uv = F_ * Xc[:, :2] / Xc[:, 2:] + [CX_, CY_] + rng.normal(0, 2.0, (n, 2))
    kps["x"], kps["y"], kps["octave"] = uv[:, 0], uv[:, 1], rng.integers(0, 8, n)
    good = (rng.random(n) < 0.85).astype(np.uint8)
    delta = np.float32(rng.uniform(1.5, 4.0))
    out = ref.localizer_do_local_ba(K, TBC, delta, Tcw, kps, Xw, good)
    sel = out["e_point"]
    assert sorted(sel.tolist()) == np.nonzero(good)[0].tolist(), "poseba points"
    T0 = Tcw.astype(np.float64).copy(); T0[:3, :3] = _requat_rot(T0[:3, :3])
    Tq = TBC.copy(); Tq[:3, :3] = _requat_rot(TBC[:3, :3])
    meas, info = oracle.plane_motion_prior(T0, Tq)
    _, st = oracle.pose_only_ba(T0, meas, info, Xw[sel].astype(np.float64), out["e_uv"], out["e_w"], F_, CX_, CY_, float(delta), 1)
    assert np.isclose(out["chi2"], st["chi2_init"], rtol=1e-8, atol=0), ("poseba", out["chi2"], st["chi2_init"])


def fuzz_pg(rng):
    import test_ref_compiled as T
    P = int(rng.integers(3, 30))
    m, w = T._global_map(synth, P, int(rng.integers(0, 1 << 30)))
    out = m.global_ba(global_iter=int(rng.integers(1, 40)))
    assert out["v_id"].tolist() == list(range(P)) and out["v_fixed"].tolist() == [True] + [False] * (P - 1), "pg vertices"
    assert len(out["e_ids"]) == len(w["odo"]) + len(w["ftr"]), "pg edges"
    pg = synth.PoseGraph(poses=out["v_est"], fixed=out["v_fixed"].astype(np.uint8), has_prior=np.ones(P, np.uint8), prior_meas=out["p_meas"],
                         prior_info=out["p_info"], o_i=out["e_ids"][:, 0].astype(np.int32), o_j=out["e_ids"][:, 1].astype(np.int32),
                         o_meas=out["e_meas"], o_info=out["e_info"])
    total, chi = oracle.pg_chi2(pg)
    assert np.isclose(out["chi2"], total, rtol=1e-8, atol=0) and np.allclose(out["e_chi2"], chi, rtol=1e-7, atol=1e-10), ("pg", out["chi2"], total)


def main(budget):
    rng = np.random.default_rng(int(os.environ.get("SEED", "20260926")))
    t0 = time.time()
    n = dict(map=0, graph=0, prior=0, sparsify=0, tri=0, poseba=0, pg=0)
    worst = 0.0
    while time.time() - t0 < budget:
        kind = ("map", "graph", "prior", "sparsify", "tri", "poseba", "pg")[int(rng.integers(0, 7))]
        if kind == "map":
            fuzz_map(rng)
        elif kind == "graph":
            fuzz_graph(rng)
        elif kind == "prior":
            for _ in range(20):
                fuzz_prior(rng)
        elif kind == "tri":
            fuzz_tri(rng)
        elif kind == "poseba":
            fuzz_poseba(rng)
        elif kind == "pg":
            fuzz_pg(rng)
        else:
            worst = max(worst, fuzz_sparsify(rng))
        n[kind] += 1
    print("fuzz_ref_backend: %.0f s, cases %s (prior: x20 each), no mismatch; sparsify end-to-end worst relative difference %.2e (ill-conditioned by construction)"
          % (time.time() - t0, n, worst))


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
