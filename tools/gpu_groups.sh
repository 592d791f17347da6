#!/bin/bash
# window batch throughput by number of stream groups
cd "$GRAFT_REPO_ROOT" || exit 1
for g in 1 2 3 4; do for n in 32 64 128; do
echo -n "groups $g windows $n: "; SE2GPU_BA_BATCH_GROUPS=$g timeout 120 python bench.py --steps 20 --warmup 10 --no-orb --no-cpu-baseline --ba-windows $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['ba_windows']['best']['iters_per_s']))"
done; done
