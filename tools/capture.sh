#!/bin/bash
# usage (GPU box, repo root): tools/capture.sh <tag>   -> everything the round's profiles/ entry needs, under gpurun_out/
#   kernel-trace summary (rocprofv3 --kernel-trace --stats), two PMC passes (FETCH_SIZE, WRITE_SIZE: separate runs),
#   per-kernel HBM traffic json, and the plain default bench line.
set -e
export TMPDIR=/tmp
# one launch per kernel and batch, as bench.py's per-kernel (profiled) pass and its algorithmic bytes per launch assume: the
# level pipeline of large batches (orb_run) would split k_fast_score / k_blur into four launches each, and two batches in
# flight (--orb-inflight 2, the default) would stretch the kernels that overlap
export SE2GPU_ORB_PIPELINE_MIN=1000000
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
cd $R
timeout 400 env SE2_BENCH_MIN_TIMED_S=0.1 tools/prof.sh $TAG --steps 50 --warmup 10 --orb-steps 5 --orb-inflight 1 --no-cpu-baseline --ba-windows 128 --ba-mixed 0 > gpurun_out/${TAG}_prof_stdout.log 2>&1 || true
export SE2_BENCH_MIN_TIMED_S=0.1   # (the floors of the bench line are for the bench line: a profiled pass needs launches, not seconds)
for C in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $R/gpurun_out/pmc_$C
  (cd /tmp && timeout 400 rocprofv3 --pmc $C --kernel-trace -f csv -d $R/gpurun_out/pmc_$C -o pmc -- python $R/bench.py --steps 10 --warmup 10 --orb-batch 256 --orb-steps 2 --orb-inflight 1 --no-cpu-baseline --ba-windows 128 --ba-mixed 0 > $R/gpurun_out/pmc_$C/stdout.log 2>&1) || true
done
F=$(find gpurun_out/pmc_FETCH_SIZE -name "*counter_collection.csv" | head -1)
W=$(find gpurun_out/pmc_WRITE_SIZE -name "*counter_collection.csv" | head -1)
SE2_PMC_WINDOWS=128 python tools/pmc_summarize.py $F $W gpurun_out/${TAG}_pmc_traffic.json || true
cp gpurun_out/${TAG}_pmc_traffic.json profiles/pmc_traffic.json || true
rm -f $F $W
unset SE2_BENCH_MIN_TIMED_S
SE2GPU_ORB_PIPELINE_MIN=16 timeout 500 python bench.py > gpurun_out/${TAG}_bench.json 2> gpurun_out/${TAG}_bench.err || true
tail -c 400 gpurun_out/${TAG}_bench.json
