#!/bin/bash
# usage (GPU box, repo root): tools/trace_gaps.sh <tag> <bench args...>  -> per-kernel start/end timeline of the BA loop
set -e
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; shift
OUT=$R/gpurun_out/trace_$TAG
mkdir -p $OUT
cd /tmp
rocprofv3 --kernel-trace -f csv -d $OUT -o $TAG -- python $R/bench.py "$@" > $OUT/stdout.log 2>&1 || true
python - "$OUT" "$TAG" <<'PY'
import csv, sys, glob, re, collections
out, tag = sys.argv[1], sys.argv[2]
f = glob.glob(f"{out}/**/*kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
names = [re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].split("<")[0] for r in rows]
# find the steady-state sequence: take the last 400 kernels
seq = list(zip(names, [int(r["Start_Timestamp"]) for r in rows], [int(r["End_Timestamp"]) for r in rows]))[-420:-20]
dur = collections.defaultdict(list); gap = collections.defaultdict(list)
for k in range(1, len(seq)):
    dur[seq[k][0]].append(seq[k][2] - seq[k][1])
    gap[seq[k - 1][0] + " -> " + seq[k][0]].append(seq[k][1] - seq[k - 1][2])
with open(f"{out}/{tag}_gaps.txt", "w") as fo:
    for n, v in dur.items():
        fo.write(f"dur {n:28s} n={len(v):4d} avg {sum(v)/len(v)/1e3:8.2f} us\n")
    for n, v in gap.items():
        fo.write(f"gap {n:50s} n={len(v):4d} avg {sum(v)/len(v)/1e3:8.2f} us\n")
print(open(f"{out}/{tag}_gaps.txt").read())
PY
rm -f $OUT/*/*kernel_trace.csv $OUT/*kernel_trace.csv 2>/dev/null || true
