#!/bin/bash
# round 3, GPU call C: lock-step window batches - parity tests + windows sweep with and without
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03c; mkdir -p $O
timeout 900 python -m pytest tests/test_ba_gpu.py tests/test_concurrency.py -q -x > $O/pytest_ba.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_ba.log
tail -25 $O/pytest_ba.log
for ls in 1 0; do
  SE2GPU_BA_LOCKSTEP=$ls timeout 300 python bench.py --steps 100 --warmup 20 --no-orb --no-cpu-baseline > $O/bench_ls$ls.json 2> $O/bench_ls$ls.err
  python - <<PY
import json
d=json.load(open("$O/bench_ls$ls.json"))
print("lockstep=$ls", round(d["value"],1), "it/s |", [(r["windows_per_gpu"], round(r["iters_per_s"])) for r in d["ba_windows"]["sweep"]])
PY
done
SE2GPU_BA_LOCKSTEP=1 timeout 300 python bench.py --steps 100 --warmup 20 --no-orb --no-cpu-baseline --ba-windows 256 > $O/bench_w256.json 2> $O/bench_w256.err
python -c "
import json; d=json.load(open('$O/bench_w256.json')); print('256 windows', [(r['windows_per_gpu'], round(r['iters_per_s'])) for r in d['ba_windows']['sweep']])"
