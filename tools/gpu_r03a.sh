#!/bin/bash
# round 3, GPU call A: validate the CPU-side batch (tests), A/B of the pivot-seed shortcut, first-cycle latency
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp
O=gpurun_out/r03a; mkdir -p $O
timeout 900 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -5 $O/pytest_gpu.log
for s in 1 0; do
  SE2GPU_BA_CHOL_SEED=$s timeout 300 python bench.py --steps 200 --warmup 20 --no-orb --no-cpu-baseline --ba-windows 0 > $O/bench_seed$s.json 2> $O/bench_seed$s.err
  python - <<PY
import json
d=json.load(open("$O/bench_seed$s.json"))
print("seed=$s", round(d["value"],1), "it/s", d["roofline"]["kernels_us"], "timed_s", round(d["timed_s"],3))
PY
done
timeout 300 python tools/latency.py > $O/latency.json 2> $O/latency.err; tail -c 1500 $O/latency.json
