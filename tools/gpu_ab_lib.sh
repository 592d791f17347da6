#!/bin/bash
# A/B of alternative builds of the library (se2lam_amd/lib/alt/libse2gpu_<tag>.so) on the ORB leg of the bench; usage: gpu_ab_lib.sh tag ...
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
cp se2lam_amd/lib/libse2gpu.so /tmp/base.so
run() { timeout 200 python bench.py --steps 20 --warmup 5 --no-cpu-baseline --ba-windows 0 --orb-steps 60 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); o=d['orb']; k=o['roofline']['kernels_us']; print('   ORB', round(o['value']), 'frames/s', round(o['ms_per_batch'],4), 'ms  fast_score', k.get('k_fast_score'), 'collect', k.get('k_cell_collect'), 'blur', k.get('k_blur'))"; }
for rep in 1 2; do for v in base "$@"; do
  if [ $v = base ]; then cp /tmp/base.so se2lam_amd/lib/libse2gpu.so; else cp se2lam_amd/lib/alt/libse2gpu_$v.so se2lam_amd/lib/libse2gpu.so; fi
  echo "== $v"; run
done; done 2>&1 | tee gpurun_out/ab_lib.txt
cp /tmp/base.so se2lam_amd/lib/libse2gpu.so
for v in "$@"; do cp se2lam_amd/lib/alt/libse2gpu_$v.so se2lam_amd/lib/libse2gpu.so; timeout 300 python -m pytest tests/test_orb_gpu.py -m gpu -x -q 2>&1 | tail -1; done
cp /tmp/base.so se2lam_amd/lib/libse2gpu.so
