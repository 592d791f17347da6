"""GPU probe of the one-workgroup-per-window BA path (csrc/ba_window.hip): parity against the multi-launch path on a few windows, then
LM iterations/s of uniform batches on both paths.  usage: python tools/resident_probe.py [counts...]"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from se2lam_amd import synth  # noqa: E402
from se2lam_amd.optimizer import SlamOptimizer, optimize_batch, reset_estimates_batch  # noqa: E402


def opt_of(g):
    o = SlamOptimizer()
    o.load(g)
    o.initializeOptimization(0)
    return o


def parity():
    gs = [synth.ba_graph(8, 60), synth.ba_graph(21, 800), synth.ba_graph(50, 5000), synth.ba_graph(30, 2000),
          synth.kidnapped(synth.ba_graph(12, 300), 400.0, 0.2, 3, 5), synth.ba_graph(20, 200, obs_per_lm=14.0)]
    os.environ["SE2GPU_BA_RESIDENT"] = "0"
    ref = []
    for g in gs:
        o = opt_of(g)
        o.optimize(10)
        ref.append((o.stats, o.estimates()))
    os.environ["SE2GPU_BA_RESIDENT"] = "1"
    os.environ["SE2GPU_BA_RESIDENT_TRACE"] = "1"
    opts = [opt_of(g) for g in gs]
    optimize_batch(opts, 10)
    os.environ["SE2GPU_BA_RESIDENT_TRACE"] = "0"
    for g, o, (st, (p, l)) in zip(gs, opts, ref):
        pp, ll = o.estimates()
        print("P %3d L %5d E %6d  trials %s | %s  chi2 %.12g | %.12g  init %.12g | %.12g  dpose %.2e dlm %.2e  lambda %.6g | %.6g" % (
            g.P, g.L, g.E, o.stats["trials_hist"][:o.stats["iterations"]], st["trials_hist"][:st["iterations"]],
            o.stats["chi2_final"], st["chi2_final"], o.stats["chi2_init"], st["chi2_init"],
            np.abs(pp - p).max(), np.abs(ll - l).max(), o.stats["lambda_final"], st["lambda_final"]))


def throughput(counts):
    g = synth.ba_graph(50, 5000)
    for path in ("0", "1"):
        os.environ["SE2GPU_BA_RESIDENT"] = path
        for n in counts:
            opts = [opt_of(g) for _ in range(n)]
            for _ in range(2):
                reset_estimates_batch(opts)
                optimize_batch(opts, 10)
            reps = 5
            t0 = time.perf_counter()
            for _ in range(reps):
                reset_estimates_batch(opts)
                optimize_batch(opts, 10)
            dt = (time.perf_counter() - t0) / reps
            its = sum(o.stats["iterations"] for o in opts)
            print("resident=%s windows %3d: %.3f ms per optimize(10) batch, %.0f LM it/s" % (path, n, dt * 1e3, its / dt), flush=True)
            del opts


if __name__ == "__main__":
    parity()
    throughput([int(a) for a in sys.argv[1:]] or [16, 64, 128, 256])
