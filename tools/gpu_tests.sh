#!/bin/bash
# full GPU test suite -> gpurun_out/<tag>/pytest_gpu.log
cd "$GRAFT_REPO_ROOT" || exit 1
O=gpurun_out/${1:-tests}; mkdir -p $O
timeout 1200 python -m pytest tests -m gpu -q ${@:2} > $O/pytest_gpu.log 2>&1; echo "pytest rc=$?" | tee -a $O/pytest_gpu.log
tail -15 $O/pytest_gpu.log
