#!/bin/bash
# BA parity tests + the BA leg of the bench (no ORB, no CPU baseline): the quick check after a change to csrc/ba.hip
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 600 python -m pytest tests/test_ba_gpu.py tests/test_golden.py tests/test_independent_pin.py -m gpu -x -q > gpurun_out/ba_quick.log 2>&1 || grep -v '^$' gpurun_out/ba_quick.log | tail -70
tail -2 gpurun_out/ba_quick.log
for i in 1 2; do
timeout 120 python bench.py --steps 200 --warmup 20 --no-orb --no-cpu-baseline ${1:+--ba-windows $1} 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value'],1), 'it/s', d['roofline']['kernels_us'], (d.get('ba_windows') or {}).get('best'))"
done
