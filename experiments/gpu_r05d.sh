#!/bin/bash
# round 5: the whole GPU suite + smoke, then the captures of tools/capture.sh (kernel-trace stats, PMC traffic, default bench line)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out/r05d; mkdir -p $O
timeout 1500 python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?"; tail -4 $O/pytest_gpu.log
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" > $O/smoke.log 2>&1; echo "smoke rc=$?"; grep -v "INFO MS" $O/smoke.log | tail -12
GRAFT_COMMIT=$1 bash tools/capture.sh r05d > $O/capture_stdout.log 2>&1; echo "capture rc=$?"; tail -3 $O/capture_stdout.log
timeout 200 python tools/latency.py > $O/latency.json 2> $O/latency.err; echo "latency rc=$?"
