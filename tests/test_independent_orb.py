"""Independent re-derivations of the two per-key-point stages of the extractor, written from the ORB definitions and not
from oracle/orb_ref.cpp, evaluated with numpy on the oracle's own pyramid levels:

* IC_Angle (/root/reference/src/ORBextractor.cpp:130-157): the angle of the intensity centroid of the radius-15 disc -
  here as float atan2 of brute-force moments; cv::fastAtan2 is a polynomial with 0.3 degrees of stated accuracy;
* computeOrbDescriptor (:161-200): steered BRIEF - the 256 point pairs of the learned pattern rotated by the key point's
  angle, coordinates rounded to the nearest pixel, compared on the BLURRED level.

They pin what the oracle's transcription could have got wrong silently: which level image each stage reads (blurred or
not), the level coordinates of a key point, the disc's row extents, the sign and order of the rotation, row / column
order of the pattern table, the bit order inside a descriptor byte.  CPU only."""
import numpy as np


def _level_xy(kp, scale):
    # key points are reported in level-0 coordinates: pt = level coordinates * mvScaleFactor[level] (float)
    x = np.rint(kp["x"].astype(np.float64) / scale).astype(int)
    y = np.rint(kp["y"].astype(np.float64) / scale).astype(int)
    return x, y


def test_orientation_is_the_angle_of_the_intensity_centroid(oracle, synth):
    t = oracle.orb_tables()
    umax = t["umax"]
    worst = 0.0
    for frame in (2, 17):
        img = synth.frame(frame)
        kps, _ = oracle.orb_extract(img)
        levels = [oracle.orb_level(img, lv, bordered=True).astype(np.int64) for lv in range(8)]   # un-blurred, 16 px frame
        for lv in range(8):
            sel = kps[kps["octave"] == lv]
            assert len(sel) > 0
            xs, ys = _level_xy(sel, float(t["scale"][lv]))
            # the level coordinates are integers: the reported ones are their float products with the scale
            assert np.array_equal((xs * t["scale"][lv]).astype(np.float32), sel["x"])
            assert np.array_equal((ys * t["scale"][lv]).astype(np.float32), sel["y"])
            I = levels[lv]
            for k in range(0, len(sel), 3):
                cx, cy = xs[k] + 16, ys[k] + 16
                m10 = m01 = 0
                for v in range(-15, 16):
                    d = int(umax[abs(v)])
                    row = I[cy + v, cx - d:cx + d + 1]
                    m10 += int((np.arange(-d, d + 1) * row).sum())
                    m01 += v * int(row.sum())
                ref = np.degrees(np.arctan2(float(m01), float(m10))) % 360.0
                err = abs((float(sel["angle"][k]) - ref + 180.0) % 360.0 - 180.0)
                worst = max(worst, err)
    assert worst < 0.3, worst          # cv::fastAtan2's accuracy


def test_descriptor_is_steered_brief_on_the_blurred_level(oracle, synth):
    t = oracle.orb_tables()
    pat = oracle.orb_pattern().reshape(256, 4).astype(np.float32)       # x0, y0, x1, y1 per comparison
    total = wrong_unblurred = 0
    for frame in (2, 17):
        img = synth.frame(frame)
        kps, desc = oracle.orb_extract(img)
        blur = [oracle.orb_level(img, lv, blurred=True, bordered=True) for lv in range(8)]
        plain = [oracle.orb_level(img, lv, bordered=True) for lv in range(8)]
        for lv in range(8):
            idx = np.nonzero(kps["octave"] == lv)[0][::2]
            xs, ys = _level_xy(kps[idx], float(t["scale"][lv]))
            for n, k in enumerate(idx):
                ang = np.float32(kps["angle"][k]) * np.float32(np.pi / np.float32(180.0))
                a, b = np.float32(np.cos(np.float64(ang))), np.float32(np.sin(np.float64(ang)))
                # rotated sample positions, rounded half to even like cvRound
                r0 = np.rint(pat[:, 0] * b + pat[:, 1] * a).astype(int); c0 = np.rint(pat[:, 0] * a - pat[:, 1] * b).astype(int)
                r1 = np.rint(pat[:, 2] * b + pat[:, 3] * a).astype(int); c1 = np.rint(pat[:, 2] * a - pat[:, 3] * b).astype(int)
                cy, cx = ys[n] + 16, xs[n] + 16
                bits = blur[lv][cy + r0, cx + c0] < blur[lv][cy + r1, cx + c1]
                got = np.packbits(bits.reshape(32, 8), axis=1, bitorder="little").reshape(32)
                assert np.array_equal(got, desc[k]), (frame, lv, int(k))
                total += 1
                bits_u = plain[lv][cy + r0, cx + c0] < plain[lv][cy + r1, cx + c1]
                wrong_unblurred += int(not np.array_equal(bits_u, bits))
    assert total > 900
    assert wrong_unblurred > 0.9 * total      # (the test can tell the blurred level from the un-blurred one)


def test_key_points_are_the_best_cell_local_fast_maxima(oracle, synth):
    """ComputeKeyPoints (:531-716) as properties that do not depend on the order of its loops: with S the FAST score map of
    a level (checked against a brute-force segment test in test_independent_pin.py),
    * every key point is a corner (S > 20, or S > 7 in a cell without any S > 20) and a strict maximum of S among its
      eight neighbours inside its own cell (cv::FAST runs per cell with non-max suppression), response = S - 1;
    * no candidate of a cell that was dropped beats a kept one of the same cell (both retainBest cuts - per cell and per
      level - keep the larger responses), and a cell uses the low threshold only if it has no corner at the high one."""
    t = oracle.orb_tables()
    img = synth.frame(5)
    kps, _ = oracle.orb_extract(img)
    geo = oracle.orb_geometry(480, 640)
    for lv in range(8):
        S = oracle.orb_score(img, lv).astype(np.int32)
        w, h, gcols, grows, cellW, cellH = (int(v) for v in geo[lv][:6])
        sel = kps[kps["octave"] == lv]
        xs, ys = _level_xy(sel, float(t["scale"][lv]))
        kept = set(zip(ys.tolist(), xs.tolist()))
        assert len(kept) == len(sel)
        assert np.array_equal(sel["response"], (S[ys, xs] - 1).astype(np.float32))
        # candidates per cell: strict 8-neighbour maxima inside the cell's window of the scan area [16, w-16) x [16, h-16)
        cand = {}
        yy, xx = np.nonzero(S[16:h - 16, 16:w - 16] > 7)
        for y, x in zip((yy + 16).tolist(), (xx + 16).tolist()):
            cj, ci = min((x - 16) // cellW, gcols - 1), min((y - 16) // cellH, grows - 1)
            xa, ya = 16 + cj * cellW, 16 + ci * cellH
            xb = w - 16 if cj == gcols - 1 else xa + cellW
            yb = h - 16 if ci == grows - 1 else ya + cellH
            nb = S[max(y - 1, ya):min(y + 2, yb), max(x - 1, xa):min(x + 2, xb)]
            if (nb >= S[y, x]).sum() == 1:
                cand.setdefault((ci, cj), []).append((y, x, int(S[y, x])))
        allc = {(y, x) for c in cand.values() for (y, x, _) in c}
        assert kept <= allc, lv                                            # corners, maxima of their cell
        for cell, c in cand.items():
            high = [p for p in c if p[2] > 20]
            pool = high if high else c                                     # iniThFAST, else minThFAST for this cell
            k_in = [p for p in pool if (p[0], p[1]) in kept]
            assert all((p[0], p[1]) not in kept for p in c if p not in pool), (lv, cell)
            if k_in:
                worst_kept = min(p[2] for p in k_in)
                assert all(p[2] <= worst_kept for p in pool if (p[0], p[1]) not in kept), (lv, cell)


def test_window_matches_satisfy_the_constraints_of_the_search(oracle, synth):
    """ORBmatcher::MatchByWindow (/root/reference/src/ORBmatcher.cpp:278-381) as order-independent properties of its
    result: every accepted pair lies inside the search window around the previous position (Frame::GetFeaturesInArea,
    Frame.cpp:222-286), within the level band, at a Hamming distance (independent popcount) of at most TH_LOW = 75; the
    assignment is one to one; the rotation differences of the survivors fall into at most three bins of the 30-bin
    histogram (ComputeThreeMaxima keeps three); and the previous positions were moved onto the partners."""
    pop = np.array([bin(i).count("1") for i in range(256)], np.int32)
    for f0, f1, win, off in ((0, 1, 20, 1), (3, 5, 25, 2), (10, 10, 10, 0)):
        k0, d0 = oracle.orb_extract(synth.frame(f0))
        k1, d1 = oracle.orb_extract(synth.frame(f1))
        m12, nm, prev = oracle.match_window(k0, d0, k1, d1, win=win, level_offset=off)
        i1 = np.nonzero(m12 >= 0)[0]
        i2 = m12[i1]
        assert nm == len(i1) > 100
        assert len(set(i2.tolist())) == len(i2)
        dx = k1["x"][i2] - k0["x"][i1]; dy = k1["y"][i2] - k0["y"][i1]
        assert (np.abs(dx) < win + 1).all() and (np.abs(dy) < win + 1).all()
        l1, l2 = k0["octave"][i1], k1["octave"][i2]
        assert (l2 >= np.maximum(l1 - off, 0)).all() and (l2 <= l1 + off).all()
        dist = pop[d0[i1] ^ d1[i2]].sum(1)
        assert (dist <= 75).all()
        rot = k0["angle"][i1] - k1["angle"][i2]
        rot = np.where(rot < 0, rot + np.float32(360), rot)
        bins = np.floor(rot * np.float32(30.0 / 360.0) + 0.5).astype(int) % 30          # round(); bin 30 wraps to 0
        assert len(np.unique(bins)) <= 3
        assert np.array_equal(prev[i1], np.stack([k1["x"][i2], k1["y"][i2]], 1))


def test_harris_responses_equal_a_vectorised_sobel_structure_tensor(oracle, synth):
    """scoreType == HARRIS_SCORE: the responses of the key points are the Harris measure of the 7 x 7 block of integer Sobel
    gradients around the key point on the UN-blurred level (HarrisResponses, ORBextractor.cpp:85-126) - recomputed here
    with whole-image numpy convolutions instead of the per-point pointer walk."""
    t = oracle.orb_tables()
    img = synth.frame(4)
    params = oracle.orb_params(score_type=oracle.HARRIS_SCORE)
    kps, _ = oracle.orb_extract(img, params)
    assert len(kps) > 500
    scale = np.float32(1.0) / (np.float32(4 * 7) * np.float32(255.0))
    s4 = scale * scale * scale * scale
    worst = 0.0
    for lv in range(8):
        I = oracle.orb_level(img, lv, params=params, bordered=True).astype(np.int64)
        Ix = np.zeros_like(I); Iy = np.zeros_like(I)
        Ix[1:-1, 1:-1] = 2 * (I[1:-1, 2:] - I[1:-1, :-2]) + (I[:-2, 2:] - I[:-2, :-2]) + (I[2:, 2:] - I[2:, :-2])
        Iy[1:-1, 1:-1] = 2 * (I[2:, 1:-1] - I[:-2, 1:-1]) + (I[2:, :-2] - I[:-2, :-2]) + (I[2:, 2:] - I[:-2, 2:])
        sel = kps[kps["octave"] == lv]
        xs, ys = _level_xy(sel, float(t["scale"][lv]))
        for k in range(len(sel)):
            cy, cx = ys[k] + 16, xs[k] + 16
            wx = Ix[cy - 3:cy + 4, cx - 3:cx + 4]; wy = Iy[cy - 3:cy + 4, cx - 3:cx + 4]
            a, b, c = int((wx * wx).sum()), int((wy * wy).sum()), int((wx * wy).sum())
            ref = (np.float32(a) * np.float32(b) - np.float32(c) * np.float32(c)
                   - np.float32(0.04) * (np.float32(a) + np.float32(b)) * (np.float32(a) + np.float32(b))) * s4
            got = float(sel["response"][k])
            worst = max(worst, abs(got - float(ref)) / max(abs(float(ref)), 1e-12))
    assert worst < 1e-5, worst


def test_projection_matches_satisfy_the_constraints_of_the_search(oracle, synth):
    """ORBmatcher::MatchByProjection (ORBmatcher.cpp:383-454) as properties of its result, with the projection done here in
    float64 numpy: a matched map point is good (not skipped), projects into the image, lies within
    mMainOctave * winSize pixels of its key point (both axes) and within the level band, at a popcount distance of at most
    TH_HIGH = 100; the key point was not observed before; no map point is used twice."""
    pop = np.array([bin(i).count("1") for i in range(256)], np.int32)
    k0, d0 = oracle.orb_extract(synth.frame(0))
    k1, d1 = oracle.orb_extract(synth.frame(1))
    for seed in (0, 1, 2):
        rng = np.random.default_rng(100 + seed)
        fx = fy = 400.0; cx, cy = 320.0, 240.0
        m = 1500
        src = rng.integers(0, len(k0), m)
        depth = rng.uniform(800, 6000, m)
        Xc = np.stack([(k0["x"][src] - cx) / fx * depth, (k0["y"][src] - cy) / fy * depth, depth], 1)
        th = 0.01
        R = np.array([[np.cos(th), 0, np.sin(th)], [0, 1, 0], [-np.sin(th), 0, np.cos(th)]])
        tt = np.array([15.0, -4.0, 8.0])
        Tcw = np.concatenate([R, tt[:, None]], 1).astype(np.float32)
        mp_pos = ((Xc - tt) @ R).astype(np.float32)
        mp_desc = d0[src].copy()
        mp_desc ^= (rng.integers(0, 256, (m, 32)).astype(np.uint8) & ((rng.random((m, 32)) < 0.03) * 255).astype(np.uint8))
        mp_oct = k0["octave"][src].astype(np.int32)
        skip = (rng.random(m) < 0.1).astype(np.uint8)
        seen = (rng.random(len(k1)) < 0.2).astype(np.uint8)
        win, off = 15, 2
        idx, nm = oracle.match_projection(mp_pos, mp_desc, mp_oct, skip, Tcw, (fx, fy, cx, cy), k1, d1, seen, win, off, 0.6)
        kp = np.nonzero(idx >= 0)[0]
        mp = idx[kp]
        assert nm == len(kp) > 50
        assert len(set(mp.tolist())) == len(mp)
        assert not skip[mp].any() and not seen[kp].any()
        Xw = mp_pos[mp].astype(np.float64)
        Xcam = Xw @ Tcw[:, :3].astype(np.float64).T + Tcw[:, 3].astype(np.float64)
        u = fx * Xcam[:, 0] / Xcam[:, 2] + cx; v = fy * Xcam[:, 1] / Xcam[:, 2] + cy
        assert (Xcam[:, 2] > 0).all() and (u >= 0).all() and (u < 640).all() and (v >= 0).all() and (v < 480).all()
        r = mp_oct[mp] * win
        assert (np.abs(k1["x"][kp] - u) <= r + 1e-2).all() and (np.abs(k1["y"][kp] - v) <= r + 1e-2).all()
        lv = k1["octave"][kp]
        assert (lv >= np.maximum(mp_oct[mp] - off, 0)).all() and (lv <= mp_oct[mp] + off).all()
        assert (pop[mp_desc[mp] ^ d1[kp]].sum(1) <= 100).all()
