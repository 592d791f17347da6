import sys
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se2lam_amd import synth
from se2lam_amd.orb import ORBextractor
ex = ORBextractor()
img = synth.frame(0)
for _ in range(20): ex(img)
