#!/bin/bash
# round 5, end: a longer randomised sweep over all eleven kinds and a short fresh-process soak of the (unchanged) dataflow solve
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out/r05f; mkdir -p $O
timeout 400 python tools/fuzz_gpu.py 300 905 > $O/fuzz_gpu_all.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz_gpu_all.log; grep -c refused $O/fuzz_gpu_all.log
timeout 300 bash tools/soak_fresh.sh 60 > $O/soak.log 2>&1; echo "soak rc=$?"; tail -3 $O/soak.log
