#!/bin/bash
# usage: tools/gpu_ab_env.sh VAR "v1 v2 ..." [windows]  -> BA bench (single window + window batch) for each value of an environment switch
cd "$GRAFT_REPO_ROOT" || exit 1
VAR=$1; N=${3:-64}
for v in $2; do for rep in 1 2; do
echo -n "$VAR=$v: "; env $VAR=$v timeout 120 python bench.py --steps 200 --warmup 20 --no-orb --no-cpu-baseline --ba-windows $N 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'it/s |', round(d['ba_windows']['best']['iters_per_s']), 'windows it/s')"
done; done
