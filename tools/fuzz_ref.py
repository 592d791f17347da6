"""Randomised sweep, CPU only: the compiled reference (oracle/_ref) against the restatement (oracle/orb_ref.cpp, match_ref.cpp)
on random image sizes / textures / extractor parameters and random matcher parameters.  usage: python tools/fuzz_ref.py [seconds]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle, ref  # noqa: E402
from se2lam_amd import synth  # noqa: E402


def main(budget):
    rng = np.random.default_rng(int(os.environ.get("SEED", "20260926")))
    t0 = time.time()
    n_ex = n_mw = n_raise = n_trig = n_e2e = 0
    feats = []
    while time.time() - t0 < budget:
        kind = rng.integers(0, 4)
        h, w = int(rng.integers(120, 700)), int(rng.integers(160, 900))
        if kind == 0:
            base = synth.frame(int(rng.integers(0, 10)))
            y0, x0 = int(rng.integers(0, max(1, 480 - min(h, 480) + 1))), int(rng.integers(0, max(1, 640 - min(w, 640) + 1)))
            img = base[y0:y0 + h, x0:x0 + w]
        elif kind == 1:
            img = rng.integers(0, 256, (h, w)).astype(np.uint8)
        elif kind == 2:   # blocky texture with flat areas
            img = np.kron(rng.integers(0, 256, (h // 16 + 1, w // 16 + 1)), np.ones((16, 16)))[:h, :w].astype(np.uint8)
            img = np.clip(img.astype(np.int32) + rng.integers(-4, 5, img.shape), 0, 255).astype(np.uint8)
        else:             # smooth gradient + sparse dots
            yy, xx = np.mgrid[0:h, 0:w]
            img = ((xx * 255 // max(w, 1) + yy * 128 // max(h, 1)) % 256).astype(np.uint8)
            pts = rng.integers(0, h * w, 300)
            img.reshape(-1)[pts] = rng.integers(0, 256, 300).astype(np.uint8)
        img = np.ascontiguousarray(img)
        levels = int(rng.integers(1, 9))
        scale = float(rng.choice([1.1, 1.2, 1.3, 1.5]))
        # (level sizes below the 2 x 16 px border + a cell make the reference divide by zero: keep the top level >= 64 px)
        while min(img.shape) / scale ** (levels - 1) < 64:
            levels -= 1
        p = oracle.orb_params(int(rng.integers(50, 3000)), scale, levels, int(rng.integers(5, 40)), int(rng.integers(0, 2)))
        try:
            a = oracle.orb_extract(img, p, cap=16384)
        except AssertionError:
            continue
        try:
            b = ref.orb_extract(img, p, cap=16384)
        except ValueError:
            n_raise += 1
            print("reference raises:", img.shape, p.nfeatures, round(p.scale_factor, 2), p.nlevels, "oracle key points:", len(a[0]))
            continue
        # round 5: the ARRAYS are compared as they come - same key points in the same order (the reference's push_back order under
        # libstdc++'s nth_element, oracle/stl_nth.h), same descriptor rows; no sorting on either side
        (ka, da), (kb, db) = a, b
        assert len(ka) == len(kb) and np.array_equal(ka, kb), ("extract", img.shape, p.nfeatures, p.scale_factor, p.nlevels, p.fast_th, p.score_type)
        if not np.array_equal(da, db):
            # would be a difference in the steering trigonometry: the restatement's default is glibc's sinf / cosf restated
            # (mode 2); report whether this machine's libm (mode 1) agrees with the compiled reference instead
            oracle.orb_trig_libm(1)
            try:
                kc, dc = oracle.orb_extract(img, p, cap=16384)
            finally:
                oracle.orb_trig_libm(2)
            rows = int((da != db).any(1).sum()); bits = int(np.unpackbits(da ^ db).sum())
            n_trig += 1
            print("libm-dependent descriptor:", img.shape, p.nfeatures, round(p.scale_factor, 2), p.nlevels, "-", rows, "descriptor(s),", bits,
                  "bit(s) of", len(da), "; libm mode equal:", bool(np.array_equal(dc, db)))
            assert False, "descriptors differ from the compiled reference"
        n_ex += 1
        if len(b[0]) > 20:
            feats.append(b)
            feats[:] = feats[-6:]
        if kind == 0 and len(b[0]) > 50 and n_ex % 3 == 0:
            # end to end FROM IMAGES: a shifted crop of the same texture through both extractors, then MatchByWindow the way
            # Track does it (prevMatched = the first frame's key-point positions): key points, descriptors and vnMatches12 equal
            dy, dx = int(rng.integers(0, 4)), int(rng.integers(0, 6))
            img2 = np.ascontiguousarray(np.roll(np.roll(img, -dy, 0), -dx, 1))
            a2, b2 = oracle.orb_extract(img2, p, cap=16384), ref.orb_extract(img2, p, cap=16384)
            assert np.array_equal(a2[0], b2[0]) and np.array_equal(a2[1], b2[1]), ("extract 2", img.shape)
            prev_o = np.ascontiguousarray(np.stack([a[0]["x"], a[0]["y"]], 1), np.float32)
            prev_r = np.ascontiguousarray(np.stack([b[0]["x"], b[0]["y"]], 1), np.float32)
            r = ref.match_window(b[0], b[1], b2[0], b2[1], prev_r, 20, 0, 0, 8, 0.9)
            o = oracle.match_window(a[0], a[1], a2[0], a2[1], prev_o, 20, 0, 0, 8, 0.9)
            assert r[1] == o[1] and np.array_equal(r[0], o[0]), ("images -> matches", img.shape, r[1], o[1])
            n_e2e += 1
        if len(feats) >= 2:
            (k1, d1), (k2, d2) = feats[int(rng.integers(0, len(feats)))], feats[int(rng.integers(0, len(feats)))]
            win = int(rng.integers(3, 60)); lo = int(rng.integers(0, 3)); mn = int(rng.integers(0, 3)); mx = mn + int(rng.integers(0, 8))
            ratio = float(rng.choice([0.6, 0.75, 0.9, 1.0]))
            prev = np.stack([k1["x"], k1["y"]], 1) + rng.normal(0, 6, (len(k1), 2)).astype(np.float32)
            r = ref.match_window(k1, d1, k2, d2, prev, win, lo, mn, mx, ratio)
            o = oracle.match_window(k1, d1, k2, d2, prev, win, lo, mn, mx, ratio)
            assert r[1] == o[1] and np.array_equal(r[0], o[0]) and np.array_equal(r[2], o[2]), ("window", win, lo, mn, mx, ratio)
            n_mw += 1
    print(f"fuzz_ref: {n_ex} extractor cases (arrays equal in order, no sorting), {n_e2e} image pairs through extract -> MatchByWindow,"
          f" {n_mw} MatchByWindow cases in {time.time() - t0:.0f} s - compiled reference == restatement"
          f" ({n_raise} inputs on which the reference itself raises; {n_trig} frames with a libm-dependent descriptor)")


if __name__ == "__main__":
    main(float(sys.argv[1]) if len(sys.argv) > 1 else 60.0)
