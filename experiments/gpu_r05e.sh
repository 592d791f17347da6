#!/bin/bash
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out/r05e; mkdir -p $O
timeout 900 python -m pytest tests/test_match_gpu.py tests/test_ref_compiled.py tests/test_pipeline.py tests/test_golden.py tests/test_golden_ref.py -m gpu -x -q > $O/match_tests.log 2>&1; echo "match tests rc=$?"; tail -4 $O/match_tests.log
timeout 200 python tools/fuzz_gpu.py 100 77 2,6,10 > $O/fuzz_match.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz_match.log
