#!/bin/bash
# rocprofv3 kernel trace of the ORB resident leg with 2 and with 3 batches in flight -> how much of the wall time has 0 / 1 / 2 / 3+ kernels
# in flight (tools/orb_timeline.py); kernel trace only, no counters
cd "${GRAFT_REPO_ROOT:-.}" || exit 1
export TMPDIR=/tmp
R=$(pwd)
for n in 2 3; do
  O=$R/gpurun_out/timeline_$n; mkdir -p $O
  (cd /tmp && timeout 300 rocprofv3 --kernel-trace -f csv -d $O -o tl -- python $R/bench.py --steps 5 --warmup 5 --no-cpu-baseline --ba-windows 0 --orb-steps 60 --orb-inflight $n > $O/stdout.log 2>&1) || true
  echo "== $n batches in flight"; python tools/orb_timeline.py $(dirname $(find $O -name "*kernel_trace.csv" | head -1)) 2>&1 | head -24
  rm -f $(find $O -name "*kernel_trace.csv")
done | tee gpurun_out/orb_timeline.txt
