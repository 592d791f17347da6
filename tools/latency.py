#!/usr/bin/env python3
"""Single-call latencies through the HOST-buffer entry points (PCIe-inclusive), the way the reference's Track /
LocalMapper threads would call them one frame / one key frame at a time.  Printed as JSON for DESIGN.md."""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from se2lam_amd import synth  # noqa: E402
from se2lam_amd.matcher import ORBmatcher  # noqa: E402
from se2lam_amd.optimizer import SlamOptimizer  # noqa: E402
from se2lam_amd.orb import ORBextractor  # noqa: E402


def timeit(fn, n=30, warm=5):
    for _ in range(warm):
        fn()
    ts = []
    for _ in range(n):
        t0 = time.perf_counter()
        fn()
        ts.append(time.perf_counter() - t0)
    return 1e3 * float(np.median(ts))


def main():
    out = {}
    ex = ORBextractor()
    imgs = [synth.frame(t) for t in range(4)]
    out["orb_extract_640x480_ms"] = timeit(lambda: ex(imgs[0]))
    (k1, d1), (k2, d2) = ex(imgs[0]), ex(imgs[1])
    mt = ORBmatcher(0.9)
    prev = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], 1), np.float32)
    out["match_by_window_ms"] = timeit(lambda: mt.MatchByWindow(k1, d1, k2, d2, prev.copy(), 20))
    for P, L in ((50, 5000), (200, 20000)):
        g = synth.ba_graph(P, L)
        o = SlamOptimizer()
        o.load(g)
        t0 = time.perf_counter()
        o.initializeOptimization(0)
        out[f"ba_{P}kf_initialize_ms"] = 1e3 * (time.perf_counter() - t0)

        def run():
            o.reset_estimates()
            o.optimize(10)
        out[f"ba_{P}kf_optimize10_ms"] = timeit(run, n=10, warm=2)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
