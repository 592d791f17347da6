#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03k; mkdir -p $O
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_concurrency.py -q -x > $O/pytest_ba.log 2>&1; echo "pytest rc=$?"; tail -5 $O/pytest_ba.log
for gr in 1 2 3 4; do
  SE2GPU_BA_BATCH_GROUPS=$gr timeout 300 python bench.py --steps 100 --warmup 20 --no-orb --no-cpu-baseline > $O/bench_g$gr.json 2> $O/bench_g$gr.err
  python - <<PY
import json
d=json.load(open("$O/bench_g$gr.json"))
print("groups=$gr", [(r["windows_per_gpu"], round(r["iters_per_s"])) for r in d["ba_windows"]["sweep"]])
PY
done
SE2GPU_BA_BATCH_GROUPS=2 timeout 300 python bench.py --steps 100 --warmup 20 --no-orb --no-cpu-baseline --ba-windows 128 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('128 windows g2', [(r['windows_per_gpu'], round(r['iters_per_s'])) for r in d['ba_windows']['sweep']])"
