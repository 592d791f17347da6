#!/bin/bash
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/r03f; mkdir -p $O
timeout 600 python -m pytest tests/test_ba_gpu.py -q -x -k "solve or batch or lm_10 or global_window or sharded_config4" > $O/pytest_ba.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest_ba.log
for lz in 1 0; do export SE2GPU_BA_CHOL_WT=$lz;
  timeout 300 python bench.py --steps 100 --warmup 20 --no-orb --no-cpu-baseline --ba-windows 64 > $O/bench_lz$lz.json 2> $O/bench_lz$lz.err
  python - <<PY
import json
d=json.load(open("$O/bench_lz$lz.json"))
print("write-through-all=$lz", round(d["value"],1), "it/s chol", d["roofline"]["kernels_us"]["k_chol_tiles"], "|", [(r["windows_per_gpu"], round(r["iters_per_s"])) for r in d["ba_windows"]["sweep"]])
PY
done
unset SE2GPU_BA_CHOL_WT; bash tools/prof_windows.sh r03f 64 | grep k_batched | head -4
