#!/bin/bash
# usage (GPU box, repo root): tools/pmc_sq.sh <tag> [kernel regex]  -> gpurun_out/<tag>_sq.json
# SQ counters per kernel (two passes of 8): where a wave's cycles go (parked on memory / issue-stalled / issuing) and how many
# instructions of which kind a wave issues.  Counter passes only (--kernel-trace), never combined with the other trace domains.
export TMPDIR=/tmp
export SE2GPU_ORB_PIPELINE_MIN=1000000
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=$1; PAT=${2:-k_cell|k_quota|k_level_select|k_resize|k_level0|k_blur|k_describe|k_orientation|k_fast_score}
A="SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_SALU"
B="SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_SCA SQ_WAIT_INST_LDS SQ_INSTS_SMEM"
i=0
for C in "$A" "$B"; do
  i=$((i+1)); mkdir -p $R/gpurun_out/pmc_sq$i
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace -f csv -d $R/gpurun_out/pmc_sq$i -o pmc -- python $R/bench.py --steps 5 --warmup 5 --orb-batch 256 --orb-steps 2 --orb-inflight 1 --no-cpu-baseline --ba-windows 0 > $R/gpurun_out/pmc_sq$i/stdout.log 2>&1) || tail -5 $R/gpurun_out/pmc_sq$i/stdout.log
done
python - "$R" "$TAG" "$PAT" <<'PY'
import csv, glob, json, re, sys, collections
R, tag, pat = sys.argv[1:4]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for i in (1, 2):
    for f in glob.glob(f"{R}/gpurun_out/pmc_sq{i}/*counter_collection.csv"):
        for r in csv.DictReader(open(f)):
            name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
            name = re.sub(r"<.*", "", name)
            if re.search(pat, name):
                acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    m = {c: sum(v) / len(v) for c, v in d.items()}
    w = max(m.get("SQ_WAVES", 1), 1)
    wc = max(m.get("SQ_WAVE_CYCLES", 1), 1)
    out[k] = {"waves": w, "per_wave": {c[3:].lower(): round(m[c] / w, 1) for c in m if c.startswith("SQ_INSTS")},
              "wave_quad_cycles_per_wave": round(wc / w, 1),
              "frac_of_wave_cycles": {c[3:].lower(): round(m[c] / wc, 3) for c in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU", "SQ_ACTIVE_INST_LDS", "SQ_ACTIVE_INST_SCA", "SQ_WAIT_INST_LDS") if c in m},
              "busy_cycles": m.get("SQ_BUSY_CYCLES")}
json.dump(out, open(f"{R}/gpurun_out/{tag}_sq.json", "w"), indent=1)
for k, v in sorted(out.items()):
    print(k, json.dumps(v))
PY
rm -f $R/gpurun_out/pmc_sq*/*counter_collection.csv $R/gpurun_out/pmc_sq*/*kernel_trace.csv
