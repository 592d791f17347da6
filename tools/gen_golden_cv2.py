#!/usr/bin/env python3
"""Regenerate tests/golden/opencv_pin.npz from REAL OpenCV (run wherever `import cv2` works; it does not in the build
container - SURVEY.md §8c - so the fixture is absent there and tests/test_opencv_pin.py skips).

The fixture holds what the reference's third-party call sites produce on the seeded synthetic frames:
  cv::resize / copyMakeBorder chain   src/ORBextractor.cpp:809-826   -> pyramid levels 1..7 of frames 0 and 5
  cv::GaussianBlur(7x7, 2, 2, REFLECT_101)            :769           -> blurred levels (blur applied to the level view
                                                                        inside its bordered image, as the reference does)
  cv::FAST(cell, thr, true) per cell                   :616,622       -> per-level key points before retainBest (frame 0)
  cv::fastAtan2                                        :156           -> 4096 samples
  cv::findFundamentalMat(FM_RANSAC, 3, 0.99)           src/Track.cpp:326 -> inlier masks of three seeded match sets
tests/test_opencv_pin.py compares the oracle (CPU) and the HIP path (GPU) with it.  The OpenCV version is stored: the
reference targets 2.4.x / 3.1+ (CI: 3.2.0); 3.4+/4.x changed GaussianBlur's fixed-point path and RANSAC's subset check.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def pyramid(cv2, img, nlevels=8, scale=1.2, edge=16):
    """ORBextractor::ComputePyramid with cv2 calls; returns (levels, bordered images)."""
    inv = [1.0]
    sf = [np.float32(1.0)]
    for i in range(1, nlevels):
        sf.append(np.float32(sf[-1] * np.float32(scale)))
    inv = [np.float32(1.0) / s for s in sf]
    levels, framed = [], []
    for lv in range(nlevels):
        w = int(round(float(np.float32(img.shape[1]) * inv[lv])))   # cvRound((float)cols * scale)
        h = int(round(float(np.float32(img.shape[0]) * inv[lv])))
        if lv == 0:
            cur = img
        else:
            cur = cv2.resize(levels[-1], (w, h), interpolation=cv2.INTER_LINEAR)
        levels.append(cur)
        framed.append(cv2.copyMakeBorder(cur, edge, edge, edge, edge, cv2.BORDER_REFLECT_101))
    return levels, framed


def fast_cells(cv2, framed, level_w, level_h, nfeatures_level, edge=16, W=30):
    """The FAST part of ORBextractor::ComputeKeyPoints (src/ORBextractor.cpp:531-630) on one level: list of (x, y, response)."""
    minX = minY = edge
    maxX, maxY = level_w - edge, level_h - edge      # in level coordinates; the reference works in the bordered image
    Wf, Hf = float(maxX - minX), float(maxY - minY)
    cols, rows = int(Wf / W), int(Hf / W)
    cw, ch = int(np.ceil(Wf / cols)), int(np.ceil(Hf / rows))
    out = []
    for i in range(rows):
        iniY = minY + i * ch - 3
        hY = ch + 6
        if i == rows - 1:
            hY = maxY + 3 - iniY
            if hY <= 0:
                continue
        for j in range(cols):
            iniX = minX + j * cw - 3
            hX = cw + 6
            if j == cols - 1:
                hX = maxX + 3 - iniX
                if hX <= 0:
                    continue
            cell = framed[iniY + edge:iniY + hY + edge, iniX + edge:iniX + hX + edge]
            kps = cv2.FastFeatureDetector_create(threshold=20, nonmaxSuppression=True).detect(np.ascontiguousarray(cell))
            if len(kps) <= 3:
                kps = cv2.FastFeatureDetector_create(threshold=7, nonmaxSuppression=True).detect(np.ascontiguousarray(cell))
            for k in kps:
                out.append((k.pt[0] + iniX, k.pt[1] + iniY, k.response, i, j))
    return np.asarray(out, np.float32).reshape(-1, 5)


def main():
    import cv2
    from se2lam_amd import synth
    out = {"cv_version": np.array(cv2.__version__)}
    for t in (0, 5):
        img = synth.frame(t)
        levels, framed = pyramid(cv2, img)
        for lv in range(1, 8):
            out[f"f{t}_level{lv}"] = levels[lv]
        for lv in range(8):
            h, w = levels[lv].shape
            view = framed[lv][16:16 + h, 16:16 + w]            # a view: GaussianBlur reads the frame at the edges
            out[f"f{t}_blur{lv}"] = cv2.GaussianBlur(view, (7, 7), 2, 2, borderType=cv2.BORDER_REFLECT_101)
        if t == 0:
            for lv in range(8):
                h, w = levels[lv].shape
                out[f"f0_fast{lv}"] = fast_cells(cv2, framed[lv], w, h, 0)
    rng = np.random.default_rng(7)
    xy = (rng.normal(size=(4096, 2)) * 500).astype(np.float32)
    out["atan2_xy"] = xy
    out["atan2_deg"] = np.array([cv2.fastAtan2(float(y), float(x)) for x, y in xy], np.float32)
    # epipolar masks: planar-ish scene + outliers, three sizes
    for n, seed in ((300, 1), (700, 2), (1000, 3)):
        r = np.random.default_rng(seed)
        X = np.stack([r.uniform(-2000, 2000, n), r.uniform(-1500, 1500, n), r.uniform(2500, 6000, n)], 1)
        p1 = np.stack([400 * X[:, 0] / X[:, 2] + 320, 400 * X[:, 1] / X[:, 2] + 240], 1)
        X2 = X - np.array([150.0, 10.0, 30.0])
        p2 = np.stack([400 * X2[:, 0] / X2[:, 2] + 320, 400 * X2[:, 1] / X2[:, 2] + 240], 1)
        p2 += r.normal(0, 0.7, p2.shape)
        bad = r.choice(n, n // 8, replace=False)
        p2[bad] += r.normal(0, 40, (bad.size, 2))
        p1 = p1.astype(np.float32); p2 = p2.astype(np.float32)
        cv2.setRNGSeed(0)   # cv::findFundamentalMat seeds its own RNG(-1); kept for determinism across builds
        _, mask = cv2.findFundamentalMat(p1, p2, cv2.FM_RANSAC, 3.0, 0.99)
        out[f"fm{n}_p1"] = p1; out[f"fm{n}_p2"] = p2
        out[f"fm{n}_mask"] = (np.zeros(n, np.uint8) if mask is None else mask.reshape(-1).astype(np.uint8))
    dst = os.path.join(ROOT, "tests", "golden", "opencv_pin.npz")
    np.savez_compressed(dst, **out)
    print("wrote", dst, "from OpenCV", cv2.__version__)


if __name__ == "__main__":
    main()
