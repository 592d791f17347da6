"""Compare the two device factorisation paths (k_chol_tiles vs k_chol_step) with numpy on the same system."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se2lam_amd import synth
from se2lam_amd.optimizer import SlamOptimizer

def make(g, mode):
    if mode == "steps":
        os.environ["SE2GPU_BA_CHOL"] = "steps"
    else:
        os.environ.pop("SE2GPU_BA_CHOL", None)
    o = SlamOptimizer(); o.load(g); o.initializeOptimization(0)
    return o

for P, L in ((8, 60), (50, 5000), (200, 20000)):
    g = synth.ba_graph(P, L)
    for mode in ("steps", "tiles"):
        o = make(g, mode)
        for lam in (50.0, 5.0, 50.0):
            S, bs = o.reduced_system(lam)
            xr = np.linalg.solve(S, bs)
            for rep in range(3):
                x, ok = o.solve(lam)
                e = np.abs(x - xr)
                print(f"P={P} {mode} lam={lam} rep={rep}: ok={ok} max|x-numpy| = {e.max():.3e} (|x|max {np.abs(xr).max():.3e}) worst idx {int(np.argmax(e))}")
