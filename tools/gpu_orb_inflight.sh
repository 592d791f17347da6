#!/bin/bash
# ORB leg by number of batches in flight
cd "$GRAFT_REPO_ROOT" || exit 1
for n in 1 2 3 4; do echo -n "inflight $n: "; timeout 200 python bench.py --steps 20 --warmup 10 --no-cpu-baseline --ba-windows 0 --orb-steps 30 --orb-inflight $n 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['orb']['value']), round(d['orb']['ms_per_batch'],3), round(d['orb']['streaming']['value']))"; done
