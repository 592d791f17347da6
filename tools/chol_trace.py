"""Critical-path trace of k_chol_tiles: SE2GPU_BA_CHOL_TRACE=1 python tools/chol_trace.py [P] 2> trace.txt"""
import os, sys
os.environ["SE2GPU_BA_CHOL_TRACE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se2lam_amd import synth
from se2lam_amd.optimizer import SlamOptimizer
P = int(sys.argv[1]) if len(sys.argv) > 1 else 200
g = synth.ba_graph(P, 100 * P)
o = SlamOptimizer(); o.load(g); o.initializeOptimization(0)
for _ in range(3):
    x, ok = o.solve(50.0)
