#!/usr/bin/env python3
"""Where the time of one LocalMapper::localBA call goes (construct, load, initializeOptimization, optimize(10), read back,
destroy), host clock, medians; with SE2GPU_BA_INIT_TRACE=1 the library adds the phases of initializeOptimization on stderr."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

from se2lam_amd import synth  # noqa: E402
from se2lam_amd.optimizer import SlamOptimizer, estimateVertexSE2, estimateVertexSBAXYZ  # noqa: E402

for P, L in ((50, 5000), (200, 20000)):
    g = synth.ba_graph(P, L)
    rows = []
    for it in range(12):
        t = [time.perf_counter()]
        q = SlamOptimizer(); t.append(time.perf_counter())
        q.load(g); t.append(time.perf_counter())
        q.initializeOptimization(0); t.append(time.perf_counter())
        q.optimize(10); t.append(time.perf_counter())
        estimateVertexSE2(q, 1); estimateVertexSBAXYZ(q, g.P); q.estimates(); t.append(time.perf_counter())
        del q; t.append(time.perf_counter())
        rows.append(np.diff(t))
    m = 1e3 * np.median(np.array(rows[2:]), axis=0)
    print(f"{P} KF: construct {m[0]:.3f}  load {m[1]:.3f}  initialize {m[2]:.3f}  optimize10 {m[3]:.3f}  readback {m[4]:.3f}  destroy {m[5]:.3f}  total {m.sum():.3f} ms")
