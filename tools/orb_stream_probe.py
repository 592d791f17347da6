"""PCIe-inclusive ORB leg in a fresh process against the number of placeholder streams created first (which hardware queue
every later stream lands on: four queues, dealt in creation order).  usage: python tools/orb_stream_probe.py <placeholders>"""
import json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from se2lam_amd import capi, orb_bench
k = int(sys.argv[1]) if len(sys.argv) > 1 else 0
keep = [capi.Stream() for _ in range(k)]
out = orb_bench.streaming_child(256, 40)
print(json.dumps({"placeholders": k, "queues": os.environ.get("GPU_MAX_HW_QUEUES"), "value": round(out["value"]), "h2d_gbs": round(out["h2d_gbs"], 1)}))
