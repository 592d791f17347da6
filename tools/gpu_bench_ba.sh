#!/bin/bash
# the BA leg of the bench only (no tests): quick A/B of a kernel change
cd "$GRAFT_REPO_ROOT" || exit 1
for i in 1 2; do
timeout 120 python bench.py --steps 200 --warmup 20 --no-orb --no-cpu-baseline ${1:+--ba-windows $1} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'it/s', d['roofline']['kernels_us'], (d.get('ba_windows') or {}).get('best'))"
done
