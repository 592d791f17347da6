"""GPU probe: the bench's batch of DISTINCT windows (synth.mixed_windows) on the resident and on the lock-step path, as a whole and
by size class.  usage: python tools/resident_mixed_probe.py [count]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from se2lam_amd import synth  # noqa: E402
from se2lam_amd.optimizer import SlamOptimizer, optimize_batch, reset_estimates_batch  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
gs = synth.mixed_windows(n)
opts = []
for g in gs:
    o = SlamOptimizer(); o.load(g); o.initializeOptimization(0); opts.append(o)


def rate(sel, path, reps=4):
    if path:
        os.environ["SE2GPU_BA_RESIDENT"] = path
    else:
        os.environ.pop("SE2GPU_BA_RESIDENT", None)   # the library's own choice
    cur = [opts[i] for i in sel]
    for _ in range(2):
        reset_estimates_batch(cur); optimize_batch(cur, 10)
    t0 = time.perf_counter()
    its = 0
    for _ in range(reps):
        reset_estimates_batch(cur); its += sum(optimize_batch(cur, 10))
    dt = time.perf_counter() - t0
    return its / dt, 1e3 * dt / reps


allw = list(range(n))
for path in ("1", "0", ""):
    r, ms = rate(allw, path)
    from se2lam_amd import capi
    print(f"resident={path or 'default'} all {n} windows: {r:.0f} it/s, {ms:.2f} ms per batch (path {capi.lib().se2gpu_ba_last_batch_path()})", flush=True)
for lo, hi in ((0, 40), (40, 50), (50, 56), (56, 58), (58, 61)):
    sel = [i for i, g in enumerate(gs) if lo <= g.P < hi]
    if sel:
        r, ms = rate(sel, "1")
        print(f"resident=1 P in [{lo}, {hi}): {len(sel)} windows, {r:.0f} it/s, {ms:.2f} ms per batch, trials {max(max(opts[i].stats['trials_hist']) for i in sel)}", flush=True)
