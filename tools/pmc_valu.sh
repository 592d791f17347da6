#!/bin/bash
# usage (GPU box, repo root): tools/pmc_valu.sh <tag>   -> gpurun_out/<tag>_valu.json
# VALUBusy (% of cycles a SIMD's VALU is busy) and LDSBankConflict (% of LDS cycles lost to bank conflicts) per kernel of
# the bench workloads: separate rocprofv3 --pmc passes (derived metrics, never combined with the trace domains gpurun refuses)
set -e
export TMPDIR=/tmp
# one launch per kernel and batch, as bench.py's per-kernel (profiled) pass and its algorithmic bytes per launch assume: the
# level pipeline of large batches (orb_run) would split k_fast_score / k_blur into four launches each, and two batches in
# flight (--orb-inflight 2, the default) would stretch the kernels that overlap
export SE2GPU_ORB_PIPELINE_MIN=1000000
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1
for C in VALUBusy LDSBankConflict; do
  mkdir -p $R/gpurun_out/pmc_$C
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace -f csv -d $R/gpurun_out/pmc_$C -o pmc -- python $R/bench.py --steps 10 --warmup 10 --orb-batch 256 --orb-steps 2 --orb-inflight 1 --no-cpu-baseline --ba-windows 0 > $R/gpurun_out/pmc_$C/stdout.log 2>&1) || true
done
python - "$R" "$TAG" <<'PY'
import csv, glob, json, re, sys, collections
R, tag = sys.argv[1], sys.argv[2]
out = {"_meta": {"source": "rocprofv3 --pmc VALUBusy / --pmc LDSBankConflict (separate passes), bench.py --steps 10 --warmup 10 --orb-batch 256 --orb-steps 2", "unit": "percent, mean over launches"}}
for C in ("VALUBusy", "LDSBankConflict"):
    fs = glob.glob(f"{R}/gpurun_out/pmc_{C}/*counter_collection.csv")
    if not fs:
        continue
    acc = collections.defaultdict(list)
    for r in csv.DictReader(open(fs[0])):
        if r.get("Counter_Name") != C:
            continue
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        acc[name].append(float(r["Counter_Value"]))
    for k, v in acc.items():
        out.setdefault(k, {})[C] = round(sum(v) / len(v), 2)
        out[k]["launches"] = len(v)
json.dump(out, open(f"{R}/gpurun_out/{tag}_valu.json", "w"), indent=1)
for k in sorted(out, key=lambda k: -out[k].get("VALUBusy", 0) if k != "_meta" else 1e9):
    if k != "_meta":
        print("%-28s %s" % (k[:28], out[k]))
PY
rm -f $R/gpurun_out/pmc_VALUBusy/*counter_collection.csv $R/gpurun_out/pmc_LDSBankConflict/*counter_collection.csv
