"""GPU probe: where a local BA of the drop-in pipeline spends its time (SE2_DROPIN_TIMES=1 prints the four parts of every forwarded
optimize()), beside the mapper step's total.  usage: python tools/dropin_times_probe.py [frames]"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["SE2_DROPIN_TIMES"] = "1"
from oracle import pipeline  # noqa: E402
from se2lam_amd import synth  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
frames, odo = synth.frames(n), pipeline.odometry(n)
cfg = pipeline.default_config()
cfg.fps = 10
pipeline.run("dropin", frames[:12], odo[:12], cfg, raw_matches=False)
print("---- timed run", file=sys.stderr, flush=True)
res = pipeline.run("dropin", frames, odo, cfg, raw_matches=False)
for i, r in enumerate(res["frames"]):
    if r["local_ba"] or r["new_kf"]:
        print(f"frame {i}: new_kf {r['new_kf']} local_ba {r['local_ba']} ms_track {r['ms_track']:.3f} ms_mapper {r['ms_mapper']:.3f} n_kfs {r['n_kfs']} n_mps {r['n_mps']}")
print("track ms/frame", res["ms_track"] / n, "mapper total ms", res["ms_mapper"])
