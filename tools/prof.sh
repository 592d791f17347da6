#!/bin/bash
# usage (on the GPU box, from the repo root): tools/prof.sh <tag> <bench args...>
# rocprofv3 kernel-trace summary of bench.py -> gpurun_out/prof_<tag>/
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/prof_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace --stats -f csv -d $OUT -o $TAG -- python $R/bench.py "$@" > $OUT/stdout.log 2>&1 || true
grep "^{" $OUT/stdout.log > $OUT/${TAG}_bench.json || true
rm -f $OUT/*kernel_trace.csv
python - "$OUT/${TAG}_kernel_stats.csv" <<'PY'
import csv, sys, re
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows[:40]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]); name = name.split("(")[0][:28]
    print(f"{name:28s} calls {int(r['Calls']):6d} avg {float(r['AverageNs'])/1e3:10.2f} us  total {float(r['TotalDurationNs'])/1e6:9.3f} ms  {float(r['Percentage']):6.2f}%")
PY
