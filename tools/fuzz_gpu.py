#!/usr/bin/env python3
"""Randomised parity sweep on the GPU (not part of the test suite: it runs for as long as it is given).
   python tools/fuzz_gpu.py [seconds] [seed] [kinds, comma separated]
ORB: random image sizes / feature counts / level counts / score types, single frames and batches against the oracle.
BA : random SE(2) windows (sizes, fixed patterns, kidnapped starts) - LM histories against the oracle.
Prints one line per case and exits non-zero at the first mismatch."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from oracle import oracle  # noqa: E402  (checker only)
from se2lam_amd import synth  # noqa: E402
from se2lam_amd.optimizer import SlamOptimizer  # noqa: E402
from se2lam_amd.orb import ORBextractor  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
KINDS = [int(k) for k in sys.argv[3].split(',')] if len(sys.argv) > 3 else list(range(12))   # e.g. 0,4,5: the three BA models only; 11 = the one-workgroup-per-window batch path
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
tex = synth.texture()
ncase = 0
from se2lam_amd import optimizer as op  # noqa: E402
from se2lam_amd.matcher import ORBmatcher  # noqa: E402
from se2lam_amd.track import Track  # noqa: E402

feat_cache = {}
track = Track()


def feats(t):
    if t not in feat_cache:
        feat_cache[t] = oracle.orb_extract(synth.frame(t))
    return feat_cache[t]


def fail(msg):
    print(msg + " MISMATCH")
    sys.exit(1)


while time.time() < t_end:
    ncase += 1
    kind = KINDS[ncase % len(KINDS)]
    if kind == 2:   # MatchByWindow on feature subsets / windows / ratios, chained vbPrevMatched
        a, b = int(rng.integers(0, 30)), int(rng.integers(0, 30))
        (k1, d1), (k2, d2) = feats(a), feats(b)
        n1, n2 = int(rng.integers(0, len(k1) + 1)), int(rng.integers(0, len(k2) + 1))
        s1, s2 = np.sort(rng.choice(len(k1), n1, replace=False)), np.sort(rng.choice(len(k2), n2, replace=False))
        k1, d1, k2, d2 = k1[s1], np.ascontiguousarray(d1[s1]), k2[s2], np.ascontiguousarray(d2[s2])
        win, lo = int(rng.choice([5, 12, 20, 45, 90])), int(rng.integers(0, 3))
        mn = int(rng.integers(0, 3)); mx = int(rng.integers(mn + 1, 9)); ratio = float(rng.choice([0.6, 0.75, 0.9, 1.0]))
        prev = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], 1), np.float32)
        # (no capacity left to refuse on: a window with more than 128 candidates takes the exact spill scan)
        nm, m12 = ORBmatcher(ratio).MatchByWindow(k1, d1, k2, d2, prev, win, lo, mn, mx)
        m_ref, nm_ref, p_ref = oracle.match_window(k1, d1, k2, d2, None, win, lo, mn, mx, ratio)
        ok = nm == nm_ref and np.array_equal(m12, m_ref) and np.array_equal(prev, p_ref)
        print(f"match frames {a}->{b} n {n1}x{n2} win {win} levels {mn}..{mx} ratio {ratio}: {nm} matches {'ok' if ok else ''}")
        if not ok:
            fail("match")
        continue
    if kind == 3:   # findFundamentalMat masks
        n = int(rng.choice([0, 5, 7, 8, 12, 15, 16, 40, 300, 1000]))
        X = np.stack([rng.uniform(-3000, 3000, n), rng.uniform(-2000, 2000, n), rng.uniform(3000, 9000, n)], 1)
        K = np.array([[400, 0, 320], [0, 400, 240], [0, 0, 1.0]])
        th = rng.uniform(-0.08, 0.08)
        R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        tt = rng.uniform(-300, 300, 3)
        uv1 = X @ K.T; uv2 = (X @ R.T + tt) @ K.T
        p1 = (uv1[:, :2] / uv1[:, 2:] + rng.normal(0, 0.5, (n, 2))).astype(np.float32) if n else np.zeros((0, 2), np.float32)
        p2 = (uv2[:, :2] / uv2[:, 2:] + rng.normal(0, 0.5, (n, 2))).astype(np.float32) if n else np.zeros((0, 2), np.float32)
        out = rng.random(n) < rng.uniform(0, 0.7)
        p2[out] += rng.uniform(-80, 80, (int(out.sum()), 2)).astype(np.float32)
        mask, ni = track.findFundamentalMat(p1, p2)
        mask_ref, ni_ref = oracle.fundamental_mask(p1, p2)
        ok = ni == ni_ref and np.array_equal(mask, mask_ref)
        print(f"ransac n {n}: {ni} inliers {'ok' if ok else ''}")
        if not ok:
            fail("ransac")
        continue
    if kind == 4:   # marginalising SE3-expmap window
        P = int(rng.integers(3, 40)); L = int(rng.integers(2 * P, 30 * P)); nref = int(rng.integers(0, min(4, P - 2) + 1))
        g = synth.ba3_graph(P, L, nref, seed=int(rng.integers(1, 10**6)))
        o = op.SlamOptimizer(); op.load_se3_graph(o, g); o.initializeOptimization(0)
        iters = int(rng.integers(1, 11))
        o.optimize(iters)
        st = oracle.ba3_optimize(g, iters)[3]
        s_ = o.stats
        ok = s_["trials_hist"] == st["trials_hist"] and np.allclose(s_["chi2_hist"], st["chi2_hist"], rtol=1e-5, atol=0)
        print(f"ba3  P {P} L {L} ref {nref} E {g.E} iters {iters}: trials {s_['trials_hist']} {'ok' if ok else ''}")
        if not ok:
            fail("ba3")
        continue
    if kind == 5:   # pose graph
        P = int(rng.integers(3, 120))
        g = synth.pose_graph(P, seed=int(rng.integers(1, 10**6)))
        o = op.SlamOptimizer(); op.load_pose_graph(o, g); o.initializeOptimization(0)
        iters = int(rng.integers(1, 11))
        o.optimize(iters)
        st = oracle.pg_optimize(g, iters)[2]
        s_ = o.stats
        ok = s_["trials_hist"] == st["trials_hist"] and np.allclose(s_["chi2_hist"], st["chi2_hist"], rtol=1e-5, atol=0)
        print(f"pg   P {P} edges {g.O} iters {iters}: trials {s_['trials_hist']} {'ok' if ok else ''}")
        if not ok:
            fail("pg")
        continue
    if kind == 6:   # MatchByProjection: map points back-projected from one frame's key points, seen from a moved key frame
        a, b = int(rng.integers(0, 30)), int(rng.integers(0, 30))
        (k0, d0), (k1, d1) = feats(a), feats(b)
        m = int(rng.choice([0, 1, 40, 700, 1500, 3000]))
        src = rng.integers(0, len(k0), m)
        depth = rng.uniform(800, 6000, m).astype(np.float32)
        Xc = np.stack([(k0["x"][src] - 320.0) / 400.0 * depth, (k0["y"][src] - 240.0) / 400.0 * depth, depth], 1).astype(np.float32)
        th = float(rng.uniform(-0.03, 0.03))
        R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], np.float32)
        tt = rng.uniform(-30, 30, 3).astype(np.float32)
        Tcw = np.concatenate([R, tt[:, None]], 1).astype(np.float32)
        mp_pos = ((Xc - tt) @ R).astype(np.float32)
        mp_desc = np.ascontiguousarray(d0[src].copy())
        flip = rng.integers(0, 256, (m, 32)).astype(np.uint8) & ((rng.random((m, 32)) < 0.03).astype(np.uint8) * 255)
        mp_desc ^= flip.astype(np.uint8)
        args = (mp_pos, mp_desc, k0["octave"][src].astype(np.int32), (rng.random(m) < 0.1).astype(np.uint8), Tcw,
                (400.0, 400.0, 320.0, 240.0), k1, d1, (rng.random(len(k1)) < 0.2).astype(np.uint8))
        win, lo = int(rng.choice([8, 15, 25])), int(rng.integers(0, 4))
        nm, idx = ORBmatcher().MatchByProjection(*args, win, lo)
        idx_ref, nm_ref = oracle.match_projection(*args, win, lo, 0.6)
        ok = nm == nm_ref and np.array_equal(idx, idx_ref)
        print(f"proj frames {a}->{b} map points {m} win {win} level offset {lo}: {nm} matches {'ok' if ok else ''}")
        if not ok:
            fail("proj")
        continue
    if kind == 7:   # doTriangulate
        n = int(rng.choice([0, 1, 30, 600, 1000]))
        K = np.array([[400.0, 0, 320.0], [0, 400.0, 240.0], [0, 0, 1]], np.float32)
        th = float(rng.uniform(-0.05, 0.05))
        R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]], np.float32)
        tt = np.array([rng.uniform(-300, -50), rng.uniform(-20, 20), rng.uniform(-50, 50)], np.float32)
        Tcr = np.eye(4, dtype=np.float32); Tcr[:3, :3] = R; Tcr[:3, 3] = tt
        P1 = (K @ np.eye(3, 4, dtype=np.float32)).astype(np.float32); P2 = (K @ Tcr[:3]).astype(np.float32)
        X = np.stack([rng.uniform(-1500, 1500, n), rng.uniform(-800, 800, n), rng.uniform(200, 14000, n)], 1).astype(np.float32)
        Xh = np.concatenate([X, np.ones((n, 1), np.float32)], 1)
        u1 = (P1 @ Xh.T).T; u2 = (P2 @ Xh.T).T
        KPd = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"), ("octave", "<i4"), ("class_id", "<i4")])
        k1 = np.zeros(n, KPd); k2 = np.zeros(n, KPd)
        if n:
            k1["x"], k1["y"] = (u1[:, 0] / u1[:, 2]), (u1[:, 1] / u1[:, 2])
            perm = rng.permutation(n)
            k2["x"][perm], k2["y"][perm] = (u2[:, 0] / u2[:, 2]) + rng.normal(0, 0.3, n), (u2[:, 1] / u2[:, 2]) + rng.normal(0, 0.3, n)
            match = perm.astype(np.int32); match[rng.random(n) < 0.1] = -1
        else:
            match = np.zeros(0, np.int32)
        has_obs = (rng.random(n) < 0.15).astype(np.uint8)
        Ocam = np.linalg.inv(Tcr)[:3, 3].astype(np.float32)
        mind = int(rng.integers(1, 5))
        ref = oracle.triangulate(k1, k2, match, has_obs, P1, P2, Ocam, 500.0, 8000.0, mind)
        got = track.doTriangulate(k1, k2, match, has_obs, P1, P2, Ocam, 500.0, 8000.0, mind)
        ok = np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1]) and np.array_equal(got[2], ref[2]) and got[3:] == ref[3:]
        print(f"tri  n {n} minDegree {mind}: {got[3]} good {'ok' if ok else ''}")
        if not ok:
            fail("tri")
        continue
    if kind == 8:   # Sparsifier::DoMarginalizeSE3XYZ, a batch of key-frame pairs
        from se2lam_amd.sparsifier import DoMarginalizeSE3XYZ_batch
        spec = [(int(rng.integers(8, 250)), int(rng.integers(0, 10**6)), float(rng.uniform(80, 900))) for _ in range(int(rng.integers(1, 9)))]
        pairs = [synth.kf_pair(*sp) for sp in spec]
        got = DoMarginalizeSE3XYZ_batch(pairs)
        ok = True
        for sp, (kf, mp, m_kf, m_mp, m_info), (z, info) in zip(spec, pairs, got):
            zr, ir, _ = oracle.sparsify(kf, mp, m_kf, m_mp, m_info)
            good = np.allclose(z, zr, atol=1e-12) and np.abs(info - ir).max() <= 1e-5 * np.abs(ir).max()
            if not good:
                w = np.linalg.eigvalsh(0.5 * (ir + ir.T))
                print(f"spars pair {sp}: rel diff {np.abs(info - ir).max() / np.abs(ir).max():.3e}, z diff {np.abs(z - zr).max():.3e}, "
                      f"eig(info_ref) {w.min():.3e} .. {w.max():.3e}")
            ok = ok and good
        print(f"spars {len(pairs)} pairs: {'ok' if ok else ''}")
        if not ok:
            fail("spars")
        continue
    if kind == 9:   # Localizer::DoLocalBA (pose-only BA with the plane-motion prior)
        from se2lam_amd.localizer import Localizer
        RBC = np.array([[0, 0, 1], [-1, 0, 0], [0, -1, 0.0]])
        TBC = np.eye(4); TBC[:3, :3] = RBC; TBC[:3, 3] = [100.0, 0.0, 300.0]

        def Twb(x, y, th):
            T = np.eye(4); c_, s_ = np.cos(th), np.sin(th)
            T[:3, :3] = [[c_, -s_, 0], [s_, c_, 0], [0, 0, 1.0]]; T[:3, 3] = [x, y, 0]
            return T
        n = int(rng.choice([0, 3, 6, 40, 400, 1500]))
        pose = (rng.uniform(-2000, 2000), rng.uniform(-2000, 2000), rng.uniform(-3, 3))
        Tcw_true = np.linalg.inv(Twb(*pose) @ TBC)
        Xc = np.stack([rng.uniform(-2000, 2000, n), rng.uniform(-1500, 1500, n), rng.uniform(1500, 8000, n)], 1)
        Xw = (np.linalg.inv(Tcw_true) @ np.c_[Xc, np.ones(n)].T).T[:, :3]
        uv = 400.0 * Xc[:, :2] / Xc[:, 2:] + [320.0, 240.0] + rng.normal(0, 0.7, (n, 2))
        out = rng.random(n) < rng.uniform(0, 0.3)
        uv[out] += rng.uniform(-50, 50, (int(out.sum()), 2))
        w = 1.0 / 1.2 ** (2 * rng.integers(0, 8, n))
        Tcw0 = np.linalg.inv(Twb(pose[0] + rng.normal(0, 40), pose[1] + rng.normal(0, 40), pose[2] + rng.normal(0, 0.03)) @ TBC)
        delta = float(np.sqrt(5.991))
        loc = Localizer()
        T = loc.DoLocalBA(Tcw0, TBC, Xw, uv, w, 400.0, 320.0, 240.0, delta, 30)
        st = loc.stats
        meas, info = oracle.plane_motion_prior(Tcw0, TBC)
        To, so = oracle.pose_only_ba(Tcw0, meas, info, Xw, uv, w, 400.0, 320.0, 240.0, delta, 30)
        # identical trial counts and costs while the steps still change the cost (tests/test_pose_ba.py::_same_run)
        h = np.asarray(so["chi2_hist"]); prev = np.concatenate([[so["chi2_init"]], h[:-1]])
        flat = (prev - h) < 1e-9 * np.maximum(h, 1e-300)
        live = min(int(np.argmax(flat)) if flat.any() else len(h), st["iterations"])
        nn = min(st["iterations"], so["iterations"])
        noise = so["chi2_init"] < 1e-18      # only the prior edge, already at its minimum: the cost is rounding noise
        ok = (noise or np.isclose(st["chi2_init"], so["chi2_init"], rtol=1e-12) and list(st["trials_hist"][:live]) == list(so["trials_hist"][:live])
              and np.allclose(st["chi2_hist"][:nn], so["chi2_hist"][:nn], rtol=1e-5, atol=1e-12)) and (
              np.allclose(T[:3, :3], To[:3, :3], atol=1e-6) and np.allclose(T[:3, 3], To[:3, 3], rtol=1e-5, atol=1e-2))
        print(f"pose n {n}: {st['iterations']} iterations {'ok' if ok else ''}")
        if not ok:
            print(st, so)
            fail("pose")
        continue
    if kind == 10:   # SearchByBoW with a stand-in vocabulary (node = the first bits of the descriptor)
        a, b = int(rng.integers(0, 30)), int(rng.integers(0, 30))
        (k1, d1), (k2, d2) = feats(a), feats(b)
        nbits = int(rng.integers(1, 9))

        def fvec(desc, keep_p):
            node = (desc[:, 0].astype(np.int32) | (desc[:, 1].astype(np.int32) << 8)) & ((1 << nbits) - 1)
            order = np.argsort(node, kind="stable")
            nodes, counts = np.unique(node[order], return_counts=True)
            ptr = np.concatenate([[0], np.cumsum(counts)])
            keep = rng.random(len(nodes)) < keep_p
            segs = [order[ptr[i]:ptr[i + 1]] for i in range(len(nodes)) if keep[i]]
            return (nodes[keep].astype(np.int32), np.concatenate([[0], np.cumsum([len(x) for x in segs])]).astype(np.int32),
                    (np.concatenate(segs) if segs else np.zeros(0, np.int64)).astype(np.int32))
        fv1, fv2 = fvec(d1, rng.uniform(0.5, 1.0)), fvec(d2, rng.uniform(0.5, 1.0))
        h1 = (rng.random(len(k1)) < 0.7).astype(np.uint8); h2 = (rng.random(len(k2)) < 0.7).astype(np.uint8)
        ratio = float(rng.choice([0.6, 0.75, 0.9])); mp_only = bool(rng.integers(0, 2)); ori = bool(rng.integers(0, 2))
        nm, m12 = ORBmatcher(ratio).SearchByBoW(k1, d1, fv1, h1, k2, d2, fv2, h2, bIfMPOnly=mp_only, checkOri=ori)
        m_ref, nm_ref = oracle.search_by_bow(k1, d1, fv1, h1, k2, d2, fv2, h2, mp_only, ratio, ori)
        ok = nm == nm_ref and np.array_equal(m12, m_ref)
        print(f"bow  frames {a},{b} bits {nbits} mpOnly {mp_only} ori {ori} ratio {ratio}: {nm} matches {'ok' if ok else ''}")
        if not ok:
            fail("bow")
        continue
    if kind == 11:   # a batch of random windows through the one-workgroup-per-window kernel (csrc/ba_window.hip), each against the oracle
        nb = int(rng.integers(1, 7))
        gs, its = [], int(rng.integers(1, 11))
        for _ in range(nb):
            P = int(rng.integers(2, 61))
            L = int(rng.integers(max(8, P), 30 * P))
            g = synth.ba_graph(P, L, obs_per_lm=float(rng.uniform(2.5, 12.0)), seed=int(rng.integers(1, 10**6)))
            if rng.random() < 0.4:
                k = int(rng.integers(1, P))
                g.poses[k, :2] += rng.normal(0, 400, 2)
            gs.append(g)
        os.environ["SE2GPU_BA_RESIDENT"] = "1"
        opts = []
        for g in gs:
            o = SlamOptimizer()
            o.load(g)
            o.initializeOptimization(0)
            opts.append(o)
        op.optimize_batch(opts, its)
        os.environ.pop("SE2GPU_BA_RESIDENT")
        from se2lam_amd import capi as _capi
        path = int(_capi.lib().se2gpu_ba_last_batch_path())
        for g, o in zip(gs, opts):
            ref = oracle.ba_optimize(g, its)[2]
            n = ref["iterations"]
            got, want = np.array(o.stats["chi2_hist"][:o.stats["iterations"]]), np.array(ref["chi2_hist"][:n])
            # a rejected / accepted trial whose gain ratio is within rounding of zero may fall either way when the sums are unordered:
            # the histories are compared while the decisions agree, and they must agree wherever |rho| is not tiny
            same = o.stats["iterations"] == n and o.stats["trials_hist"][:n] == ref["trials_hist"][:n]
            ok = same and np.allclose(got, want, rtol=1e-5)
            if not same and np.abs(np.array(ref["rho_log"])).min() < 1e-6:
                ok = True
            print(f"baw  path {path} P {g.P} L {g.L} E {g.E} iters {its}: trials {o.stats['trials_hist'][:o.stats['iterations']]} {'ok' if ok else 'MISMATCH'}")
            if not ok or path != 2:
                print(got, want, ref["trials_hist"][:n])
                sys.exit(1)
        continue
    if kind == 1:
        W, H = int(rng.integers(160, 900)), int(rng.integers(120, 700))
        nf = int(rng.choice([150, 500, 1000, 2000]))
        nl = int(rng.integers(1, 9))
        st = int(rng.integers(0, 2))
        B = int(rng.choice([1, 1, 3, 16, 21]))
        # the smallest level must keep a scan area (the library refuses otherwise, like the reference would crash)
        if min(W, H) / 1.2 ** (nl - 1) < 2 * 16 + 8:
            continue
        imgs = []
        for b in range(B):
            ox, oy = int(rng.integers(0, 1280 - W)), int(rng.integers(0, 960 - H))
            imgs.append(np.ascontiguousarray(tex[oy:oy + H, ox:ox + W]))
        try:
            ex = ORBextractor(nfeatures=nf, nlevels=nl, scoreType=st, max_rows=H, max_cols=W, max_batch=B)
        except Exception as e:   # geometry the library refuses: since round 5 only where the reference itself raises (a cell
            # window outside its level, cv::Mat::colRange) - checked against the compiled reference where it is present - or
            # where a level has more than 1024 cells
            msg = str(e)
            verdict = ""
            if "the reference raises" in msg:
                try:
                    from oracle import ref
                    if ref.available():
                        try:
                            ref.orb_extract(imgs[0], oracle.orb_params(nfeatures=nf, nlevels=nl, score_type=st), cap=4 * nf + 64)
                            print(f"orb  {W}x{H} nf {nf} levels {nl}: refused, but the compiled reference runs it MISMATCH")
                            sys.exit(1)
                        except ValueError:
                            verdict = " - and the compiled reference raises"
                except ImportError:
                    pass
            print(f"orb  {W}x{H} nf {nf} levels {nl}: refused ({msg[:90]}){verdict}")
            continue
        try:
            out = ex.extract_batch(np.stack(imgs)) if B > 1 else [ex(imgs[0])]
        except Exception as e:
            # (HARRIS_SCORE cells with more than 4096 corners are selected in bands of rows now: no capacity to refuse on)
            print(f"orb  {W}x{H} nf {nf} levels {nl} score {st} batch {B}: ERROR {str(e)[:100]}")
            sys.exit(1)
        p = oracle.orb_params(nfeatures=nf, nlevels=nl, score_type=st)
        for b in sorted({0, B - 1}):
            ko, do = oracle.orb_extract(imgs[b], p, cap=4 * nf + 64)
            ok = np.array_equal(out[b][0], ko) and np.array_equal(out[b][1], do)
            print(f"orb  {W}x{H} nf {nf} levels {nl} score {st} batch {B} frame {b}: {len(ko)} kp {'ok' if ok else 'MISMATCH'}")
            if not ok:
                sys.exit(1)
    else:
        big = rng.random() < 0.25    # loops long enough for the nested-dissection orders of the pose solve (several levels)
        P = int(rng.integers(70, 190)) if big else int(rng.integers(2, 70))
        L = int(rng.integers(max(8, P), (12 if big else 40) * P))
        g = synth.ba_graph(P, L, obs_per_lm=float(rng.uniform(2.5, 9.0)), seed=int(rng.integers(1, 10**6)))
        if rng.random() < 0.4:   # a start far from the optimum: rejected trials
            k = int(rng.integers(1, P))
            g.poses[k, :2] += rng.normal(0, 400, 2)
        iters = int(rng.integers(1, 12))
        o = SlamOptimizer()
        o.load(g)
        o.initializeOptimization(0)
        o.optimize(iters)
        ref = oracle.ba_optimize(g, iters)[2]
        got, want = np.array(o.stats["chi2_hist"][:o.stats["iterations"]]), np.array(ref["chi2_hist"][:ref["iterations"]])
        ok = (o.stats["iterations"] == ref["iterations"] and o.stats["trials_hist"][:ref["iterations"]] == ref["trials_hist"][:ref["iterations"]]
              and np.allclose(got, want, rtol=1e-5))
        print(f"ba   P {P} L {L} E {g.E} iters {iters}: trials {o.stats['trials_hist'][:o.stats['iterations']]} {'ok' if ok else 'MISMATCH'}")
        if not ok:
            print(got, want, ref["trials_hist"][:ref["iterations"]])
            sys.exit(1)
print(f"{ncase} cases, no mismatch")
