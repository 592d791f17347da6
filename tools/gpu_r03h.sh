#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03h; mkdir -p $O
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_concurrency.py -q -x > $O/pytest_ba.log 2>&1; echo "pytest rc=$?"; tail -30 $O/pytest_ba.log
for nd in 1 0; do
  SE2GPU_BA_ND=$nd timeout 300 python bench.py --steps 100 --warmup 20 --no-orb --no-cpu-baseline > $O/bench_nd$nd.json 2> $O/bench_nd$nd.err
  python - <<PY
import json
d=json.load(open("$O/bench_nd$nd.json"))
print("nd=$nd", round(d["value"],1), "it/s", d["roofline"]["kernels_us"], "|", [(r["windows_per_gpu"], round(r["iters_per_s"])) for r in d["ba_windows"]["sweep"]])
PY
done
SE2GPU_BA_INIT_TRACE=1 python tools/latency.py 2> $O/latency.err | tail -c 900
