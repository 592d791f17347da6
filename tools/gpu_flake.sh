#!/bin/bash
# repeat the bit-identity tests and keep the first failure's message
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
timeout 400 python tools/flake_hunt.py ${1:-20} 2>&1 | tail -25
for i in $(seq 1 ${2:-12}); do
  timeout 200 python -m pytest tests/test_ba_gpu.py -m gpu -x -q > gpurun_out/flake_run.log 2>&1 || { echo "run $i FAILED"; grep -v "^$" gpurun_out/flake_run.log | tail -60; break; }
done
echo "loop done"
