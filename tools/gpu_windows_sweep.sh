#!/bin/bash
# window-batch throughput against the number of windows in flight (where does it peak?)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
for n in 16 24 32 40 48 64 96; do
  echo -n "windows $n: "; timeout 120 python bench.py --steps 50 --warmup 10 --no-orb --no-cpu-baseline --ba-windows $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); b=d['ba_windows']['best']; print(round(b['iters_per_s']), 'it/s', round(b['ms_per_optimize10'],3), 'ms', 'mixed', round((d['ba_windows'].get('mixed') or {}).get('iters_per_s',0)))"
done 2>&1 | tee gpurun_out/windows_sweep.txt
