#!/bin/bash
# A/B of the solver's pose order on the 200-KF graph (natural order against nested dissection)
cd "$GRAFT_REPO_ROOT" || exit 1
run() { timeout 120 env "$@" python bench.py --steps 200 --warmup 20 --no-orb --no-cpu-baseline --ba-windows 0 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('   ', round(d['value'],1), 'it/s  chol', d['roofline']['kernels_us']['k_chol_tiles'])"; }
for nd in 0 1 0 1; do
  echo "ND=$nd"; run SE2GPU_BA_ND=$nd
done
