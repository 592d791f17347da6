import os, sys, time
sys.path.insert(0, os.getcwd())
from se2lam_amd import synth
from se2lam_amd.optimizer import SlamOptimizer, optimize_batch, reset_estimates_batch
g = synth.ba_graph(50, 5000)
os.environ["SE2GPU_BA_RESIDENT"] = "1"
opts = []
for _ in range(16):
    o = SlamOptimizer(); o.load(g); o.initializeOptimization(0); opts.append(o)
for dbg in ("0", "1", "2", "3"):
    os.environ["SE2GPU_BA_RESIDENT_DEBUG"] = dbg
    os.environ["SE2GPU_BA_RESIDENT_TRACE"] = "1"
    print("debug", dbg, flush=True)
    for _ in range(2):
        reset_estimates_batch(opts)
        optimize_batch(opts, 3)
