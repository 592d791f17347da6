#!/usr/bin/env python3
"""Randomised parity sweep on the GPU (not part of the test suite: it runs for as long as it is given).
   python tools/fuzz_gpu.py [seconds] [seed]
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
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
t_end = time.time() + budget
tex = synth.texture()
ncase = 0
while time.time() < t_end:
    ncase += 1
    if ncase % 2:
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
        except Exception as e:   # geometry the library refuses (cell grid limits): not a parity case
            print(f"orb  {W}x{H} nf {nf} levels {nl}: refused ({str(e)[:60]})")
            continue
        try:
            out = ex.extract_batch(np.stack(imgs)) if B > 1 else [ex(imgs[0])]
        except Exception as e:
            # HARRIS_SCORE retains by a response that is only known after all FAST corners of a cell have been scored: a cell
            # with more than 4096 of them is a documented capacity error there (FAST_SCORE cuts by score first)
            if st == 0 and "capacity overflow" in str(e):
                print(f"orb  {W}x{H} nf {nf} levels {nl} score {st} batch {B}: refused (Harris cell capacity)")
                continue
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
        P = int(rng.integers(2, 70))
        L = int(rng.integers(max(8, P), 40 * P))
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
