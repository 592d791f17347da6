#!/bin/bash
# kernel-trace of the BA leg: gaps between consecutive kernels (where an iteration's 162 us go besides the kernels)
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
D=$R/gpurun_out/gaps; rm -rf $D; mkdir -p $D
(cd /tmp && timeout 200 rocprofv3 --kernel-trace -f csv -d $D -o tr -- python $R/bench.py --steps 100 --warmup 20 --no-orb --no-cpu-baseline --ba-windows 0 > $D/stdout.log 2>&1)
python - "$D" <<'PY'
import csv, glob, sys, re, collections
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::|void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0]) for r in csv.DictReader(open(f))]
rows.sort()
# the last 60 % of the run: steady state (graph replays)
rows = rows[int(len(rows) * 0.5):]
gap = collections.defaultdict(list)
for (s0, e0, n0), (s1, e1, n1) in zip(rows, rows[1:]):
    gap[(n0, n1)].append((s1 - e0) / 1e3)
for k, v in sorted(gap.items(), key=lambda kv: -sum(kv[1]))[:14]:
    print("%-18s -> %-18s  n %5d  mean gap %7.2f us  total %8.1f us" % (k[0][:18], k[1][:18], len(v), sum(v) / len(v), sum(v)))
dur = collections.defaultdict(list)
for s, e, n in rows: dur[n].append((e - s) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:8]:
    print("%-22s n %5d  mean %7.2f us" % (k[:22], len(v), sum(v) / len(v)))
span = (rows[-1][1] - rows[0][0]) / 1e3
nit = len(dur.get("k_chol_tiles", []))
print("span %.1f us, %d solves -> %.2f us per iteration; kernels %.1f us" % (span, nit, span / max(nit, 1), sum(sum(v) for v in dur.values()) / max(nit, 1)))
PY
rm -rf $D
