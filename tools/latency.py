#!/usr/bin/env python3
"""Single-call latencies through the HOST-buffer entry points (PCIe-inclusive), the way the reference's Track /
LocalMapper threads would call them one frame / one key frame at a time.  Printed as JSON for DESIGN.md."""
import json
import os
import sys
import subprocess
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from se2lam_amd import synth  # noqa: E402
from se2lam_amd.matcher import ORBmatcher  # noqa: E402
from se2lam_amd.optimizer import SlamOptimizer, estimateVertexSE2, estimateVertexSBAXYZ  # noqa: E402
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

    def fresh():   # Track.cpp:131: "ORBmatcher matcher(0.9);" on the stack of every frame
        m = ORBmatcher(0.9)
        m.MatchByWindow(k1, d1, k2, d2, prev.copy(), 20)
        del m
    out["match_by_window_fresh_matcher_ms"] = timeit(fresh)
    # Track thread: removeOutliers + doTriangulate on the MatchByWindow result of frames 0 -> 5
    from se2lam_amd.track import Track
    tr = Track()
    k5, d5 = ex(synth.frame(5))
    prev5 = np.ascontiguousarray(np.stack([k1["x"], k1["y"]], 1), np.float32)
    nm, m12 = mt.MatchByWindow(k1, d1, k5, d5, prev5, 20)
    out["remove_outliers_ms"] = timeit(lambda: tr.removeOutliers(k1, k5, np.ascontiguousarray(m12, np.int32).copy()))
    K = np.array([[400, 0, 320], [0, 400, 240], [0, 0, 1.0]], np.float32)
    Tcr = np.eye(4, dtype=np.float32); Tcr[0, 3] = -112.5; Tcr[1, 3] = -37.5
    P0 = K @ np.eye(4, dtype=np.float32)[:3]; P1 = (K @ Tcr[:3]).astype(np.float32)
    oc = np.array([112.5, 37.5, 0], np.float32)
    out["do_triangulate_ms"] = timeit(lambda: tr.doTriangulate(k1, k5, m12, None, P0, P1, oc, 300, 12000, 2))
    # LocalMapper thread: MatchByProjection of 1500 local map points into a new key frame
    rng = np.random.default_rng(0)
    m = 1500
    src = rng.integers(0, len(k1), m)
    depth = rng.uniform(800, 6000, m).astype(np.float32)
    Xc = np.stack([(k1["x"][src] - 320) / 400 * depth, (k1["y"][src] - 240) / 400 * depth, depth], 1).astype(np.float32)
    Tcw = np.concatenate([np.eye(3, dtype=np.float32), np.array([[15.0], [-4.0], [8.0]], np.float32)], 1)
    mp_pos = (Xc - Tcw[:, 3]).astype(np.float32)
    args = (mp_pos, d1[src].copy(), k1["octave"][src].astype(np.int32), np.zeros(m, np.uint8), Tcw, (400.0, 400.0, 320.0, 240.0),
            k2, d2, np.zeros(len(k2), np.uint8))
    mp = ORBmatcher()
    out["match_by_projection_1500mp_ms"] = timeit(lambda: mp.MatchByProjection(*args, 15, 2))
    for P, L in ((50, 5000), (200, 20000)):
        g = synth.ba_graph(P, L)
        o = SlamOptimizer()
        o.load(g)
        t0 = time.perf_counter()
        o.initializeOptimization(0)
        out[f"ba_{P}kf_initialize_first_ms"] = 1e3 * (time.perf_counter() - t0)   # one un-warmed shot: first graph of this size in the process
        inits, opts10 = [], []

        def run():
            o.reset_estimates()
            o.optimize(10)
        out[f"ba_{P}kf_optimize10_ms"] = timeit(run, n=10, warm=2)

        def cycle():   # LocalMapper::localBA as the reference runs it: a fresh SlamOptimizer per call
            q = SlamOptimizer()
            q.load(g)
            ta = time.perf_counter()
            q.initializeOptimization(0)
            tb = time.perf_counter()
            q.optimize(10)
            tc = time.perf_counter()
            estimateVertexSE2(q, 1); estimateVertexSBAXYZ(q, g.P)   # served from one download of all estimates
            q.estimates()
            del q
            inits.append(1e3 * (tb - ta)); opts10.append(1e3 * (tc - tb))
        out[f"ba_{P}kf_construct_load_initialize_optimize10_ms"] = timeit(cycle, n=10, warm=2)
        # inside that cycle (warmed medians): initializeOptimization of a fresh optimizer, and its first - never replayed - optimize(10)
        out[f"ba_{P}kf_initialize_ms"] = float(np.median(inits[2:]))
        out[f"ba_{P}kf_first_optimize10_ms"] = float(np.median(opts10[2:]))
    # the FIRST localBA of a process (cold: code objects, stream, mailbox, ~50 allocations) against the first one after
    # se2gpu_ba_reserve(P, L, E) at start-up - each measured in a fresh process
    for tag, reserve in (("cold", False), ("after_reserve", True)):
        r = subprocess.run([sys.executable, os.path.abspath(__file__), "--first-cycle", "1" if reserve else "0"],
                           capture_output=True, text=True, timeout=300)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode == 0 and lines:
            d = json.loads(lines[-1])
            out[f"ba_50kf_first_cycle_{tag}_ms"] = d["first_cycle_ms"]
            out[f"ba_50kf_first_initialize_{tag}_ms"] = d["first_initialize_ms"]
            if reserve:
                out["ba_reserve_50kf_ms"] = d["reserve_ms"]
        else:
            out[f"ba_50kf_first_cycle_{tag}_ms"] = None
            print(r.stderr[-500:], file=sys.stderr)
    print(json.dumps(out))


def first_cycle(reserve):
    from se2lam_amd import capi, synth
    from se2lam_amd.optimizer import SlamOptimizer, estimateVertexSE2, estimateVertexSBAXYZ
    g = synth.ba_graph(50, 5000)
    capi.device_count()
    t0 = time.perf_counter()
    if reserve:
        capi.check(capi.lib().se2gpu_ba_reserve(g.P, g.L, g.E))
    t1 = time.perf_counter()
    q = SlamOptimizer()
    q.load(g)
    ti = time.perf_counter()
    q.initializeOptimization(0)
    tj = time.perf_counter()
    q.optimize(10)
    estimateVertexSE2(q, 1); estimateVertexSBAXYZ(q, g.P)
    q.estimates()
    del q
    t2 = time.perf_counter()
    print(json.dumps({"reserve_ms": 1e3 * (t1 - t0), "first_cycle_ms": 1e3 * (t2 - t1), "first_initialize_ms": 1e3 * (tj - ti)}))


if __name__ == "__main__":
    if len(sys.argv) >= 3 and sys.argv[1] == "--first-cycle":
        first_cycle(sys.argv[2] == "1")
    else:
        main()
