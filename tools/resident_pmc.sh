#!/bin/bash
# usage (GPU box, repo root): tools/resident_pmc.sh [windows]  -> HBM traffic of the resident BA kernel per window and LM iteration
# (two rocprofv3 PMC passes, FETCH_SIZE and WRITE_SIZE; units and the factor 2 on FETCH_SIZE as in tools/pmc_summarize.py)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
N=${1:-128}
cat > /tmp/resident_pmc_run.py <<PY
import os, sys
sys.path.insert(0, "$R")
os.environ["SE2GPU_BA_RESIDENT"] = "1"
from se2lam_amd import synth
from se2lam_amd.optimizer import SlamOptimizer, optimize_batch, reset_estimates_batch
g = synth.ba_graph(50, 5000)
opts = []
for _ in range($N):
    o = SlamOptimizer(); o.load(g); o.initializeOptimization(0); opts.append(o)
its = 0
for _ in range(3):
    reset_estimates_batch(opts); its = sum(optimize_batch(opts, 10))
print("ITERATIONS_PER_LAUNCH", its, "E", g.E, "L", g.L)
PY
for C in FETCH_SIZE WRITE_SIZE; do
  rm -rf /tmp/rpmc_$C; mkdir -p /tmp/rpmc_$C
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace -f csv -d /tmp/rpmc_$C -o pmc -- python /tmp/resident_pmc_run.py > /tmp/rpmc_$C/stdout.log 2>&1)
done
python - <<PY
import csv, glob, re
def avg(c):
    f = glob.glob("/tmp/rpmc_%s/**/*counter_collection.csv" % c, recursive=True)[0]
    v = [float(r["Counter_Value"]) for r in csv.DictReader(open(f)) if "k_window_lm" in r["Kernel_Name"]]
    return sum(v) / len(v), len(v)
its = int(re.search(r"ITERATIONS_PER_LAUNCH (\d+)", open("/tmp/rpmc_FETCH_SIZE/stdout.log").read()).group(1))
line = re.search(r"ITERATIONS_PER_LAUNCH.*", open("/tmp/rpmc_FETCH_SIZE/stdout.log").read()).group(0)
(f, n), (w, _) = avg("FETCH_SIZE"), avg("WRITE_SIZE")
print(line)
print("k_window_lm: %d launches, FETCH_SIZE %.0f KiB, WRITE_SIZE %.0f KiB per launch -> %.3f MB per window and LM iteration (fetch %.3f, write %.3f)" % (
    n, f, w, (2 * f + w) * 1024 / its / 1e6, 2 * f * 1024 / its / 1e6, w * 1024 / its / 1e6))
PY
