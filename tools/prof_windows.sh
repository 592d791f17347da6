#!/bin/bash
# rocprofv3 kernel-trace summary of the lock-step window batch (bench.py --ba-windows N) -> gpurun_out/<tag>_windows_kernel_stats.csv
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-win}; N=${2:-64}
D=$R/gpurun_out/prof_$TAG; mkdir -p $D
(cd /tmp && timeout 300 rocprofv3 --kernel-trace --stats -f csv -d $D -o win -- python $R/bench.py --steps 20 --warmup 10 --no-orb --no-cpu-baseline --ba-windows $N > $D/stdout.log 2> $D/stderr.log)
S=$(find $D -name "*kernel_stats.csv" | head -1)
cp $S $R/gpurun_out/${TAG}_windows_kernel_stats.csv
python - <<PY
import csv
rows=list(csv.DictReader(open("$S")))
for r in rows[:16]:
    n=r["Name"]
    if "k_batched" in n:
        import re
        m=re.search(r"d_[a-z_0-9]+(<[a-z]+>)?", n); n="k_batched:"+(m.group(0) if m else n[:60])
    print(f'{n[:48]:48s} calls {r["Calls"]:>6s} avg_us {float(r["AverageNs"])/1e3:9.1f} total_ms {float(r["TotalDurationNs"])/1e6:9.2f} pct {r["Percentage"]}')
PY
find $D -name "*.csv" ! -name "*kernel_stats.csv" -delete
