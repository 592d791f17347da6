#!/bin/bash
# round 5, first lease: (1) k_angle_trig with glibc's sinf / cosf restated - ORB parity suite + a short fuzz; (2) the 64-wide block
# column experiments; (3) A/B of the scalar pivot test (se2lam_amd/lib/alt/libse2gpu_pivot.so = the tree + tools/patches/r05_k_chol_tiles_scalar_pivot_test.patch)
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
O=gpurun_out/r05a; mkdir -p $O
timeout 900 python -m pytest tests/test_orb_gpu.py tests/test_ref_compiled.py tests/test_golden_ref.py tests/test_golden.py tests/test_match_gpu.py -m gpu -x -q > $O/orb_tests.log 2>&1; echo "orb tests rc=$?"; tail -3 $O/orb_tests.log
timeout 200 python tools/fuzz_gpu.py 90 501 1,2 > $O/fuzz_gpu.log 2>&1; echo "fuzz rc=$?"; tail -2 $O/fuzz_gpu.log
timeout 280 bash tools/gpu_chol64.sh > $O/chol64_stdout.log 2>&1; echo "chol64 rc=$?"; cp gpurun_out/chol64.txt $O/ 2>/dev/null
run() { timeout 120 python bench.py --steps 200 --warmup 20 --no-orb --no-cpu-baseline --ba-windows 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   ', round(d['value'],1), 'it/s', d['roofline']['kernels_us'])"; }
cp se2lam_amd/lib/libse2gpu.so /tmp/base.so
for v in base pivot base pivot; do
  if [ $v = pivot ]; then cp se2lam_amd/lib/alt/libse2gpu_pivot.so se2lam_amd/lib/libse2gpu.so; else cp /tmp/base.so se2lam_amd/lib/libse2gpu.so; fi
  echo "== $v"; run
done 2>&1 | tee $O/pivot_ab.txt
cp se2lam_amd/lib/alt/libse2gpu_pivot.so se2lam_amd/lib/libse2gpu.so
timeout 400 python -m pytest tests/test_ba_gpu.py -m gpu -x -q > $O/ba_tests_pivot.log 2>&1; echo "ba tests (pivot) rc=$?"; tail -2 $O/ba_tests_pivot.log
cp /tmp/base.so se2lam_amd/lib/libse2gpu.so
