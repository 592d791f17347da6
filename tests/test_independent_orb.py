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
