#!/bin/bash
# round 5: the retainBest cuts as libstdc++'s nth_element on the device - parity suites, fuzz, ORB bench
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out/r05b; mkdir -p $O
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_ref_compiled.py tests/test_golden_ref.py tests/test_golden.py tests/test_match_gpu.py tests/test_independent_orb.py -m gpu -x -q > $O/orb_tests.log 2>&1; echo "orb tests rc=$?"; tail -25 $O/orb_tests.log
timeout 200 python tools/fuzz_gpu.py 90 502 1,2 > $O/fuzz_gpu.log 2>&1; echo "fuzz rc=$?"; tail -3 $O/fuzz_gpu.log
timeout 300 python bench.py --no-cpu-baseline --ba-windows 0 --steps 50 > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<'P'
import json
d=json.load(open("gpurun_out/r05b/bench.json"))
o=d["orb"]; print("BA", round(d["value"],1), "ORB", round(o["value"]), "frames/s", o["ms_per_batch"], "ms/batch", o["roofline"]["kernels_us"], "streaming", round(o["streaming"]["value"]))
P
