import time, sys, numpy as np
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from se2lam_amd import synth
from se2lam_amd.matcher import ORBmatcher
from se2lam_amd.orb import ORBextractor
from oracle import oracle as O
import test_match_gpu as T
ex = ORBextractor()
k0, d0 = ex(synth.frame(0)); k1, d1 = ex(synth.frame(1))
mt = ORBmatcher(0.9)
def prev(): return np.ascontiguousarray(np.stack([k0["x"], k0["y"]], 1), np.float32)
for _ in range(5): mt.MatchByWindow(k0, d0, k1, d1, prev(), 20)
t0 = time.perf_counter()
for _ in range(100): nm, m = mt.MatchByWindow(k0, d0, k1, d1, prev(), 20)
print("MatchByWindow single pair: %.1f us per call, %d matches" % ((time.perf_counter() - t0) / 100 * 1e6, nm))
feats = [O.orb_extract(synth.frame(t)) for t in range(2)]
args = T._projection_case(O, feats, 0)
mp = ORBmatcher()
for _ in range(5): mp.MatchByProjection(*args, 15, 2)
t0 = time.perf_counter()
for _ in range(50): nm, idx = mp.MatchByProjection(*args, 15, 2)
print("MatchByProjection m=1500: %.1f us per call, %d matches" % ((time.perf_counter() - t0) / 50 * 1e6, nm))
