"""repeat the lock-step batch scenario and print the first difference in full (debug tool)"""
import os, sys, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import numpy as np
from se2lam_amd import synth
from se2lam_amd.optimizer import SlamOptimizer, optimize_batch, reset_estimates_batch
import test_ba_gpu as T

def opt(g):
    o = SlamOptimizer(); o.load(g); o.initializeOptimization(0); return o

graphs = [T._kidnapped(synth, *c[0]) for c in T.LM_REJECT_CASES] + [synth.ba_graph(8, 60), synth.ba_graph(30, 2000), synth.ba_graph(50, 5000)]
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
bad = 0
for rep in range(reps):
    for iters in (4, 10):
        ref = []
        for g in graphs:
            o = opt(g); o.optimize(iters); ref.append(o.stats)
        opts = [opt(g) for g in graphs]
        for variant in ("single-reset", "batch-reset"):
            if variant == "single-reset":
                for o in opts: o.reset_estimates()
            else:
                reset_estimates_batch(opts)
            optimize_batch(opts, iters)
            for i, (o, r) in enumerate(zip(opts, ref)):
                if o.stats != r:
                    bad += 1
                    print("MISMATCH rep", rep, "iters", iters, variant, "window", i)
                    for k in r:
                        if o.stats[k] != r[k]:
                            print("   ", k, "\n      got ", o.stats[k], "\n      want", r[k])
        # single runs again on fresh handles: is the single path itself reproducible?
        for i, g in enumerate(graphs):
            o = opt(g); o.optimize(iters)
            if o.stats != ref[i]:
                bad += 1
                print("SINGLE-RUN MISMATCH rep", rep, "iters", iters, "window", i)
                for k in ref[i]:
                    if o.stats[k] != ref[i][k]:
                        print("   ", k, "\n      got ", o.stats[k], "\n      want", ref[i][k])
print("reps", reps, "mismatches", bad)
