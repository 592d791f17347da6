import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
os.environ["SE2GPU_BA_RESIDENT"] = "1"
from se2lam_amd import synth
from se2lam_amd.optimizer import SlamOptimizer, optimize_batch, reset_estimates_batch
g = synth.ba_graph(50, 5000)
opts = []
for _ in range(256):
    o = SlamOptimizer(); o.load(g); o.initializeOptimization(0); opts.append(o)
for it in (0, 1, 2, 5, 10):
    for _ in range(2):
        reset_estimates_batch(opts); optimize_batch(opts, it)
    t0 = time.perf_counter()
    for _ in range(5):
        reset_estimates_batch(opts); optimize_batch(opts, it)
    print("optimize(%d): %.3f ms per batch" % (it, (time.perf_counter() - t0) / 5 * 1e3), flush=True)
