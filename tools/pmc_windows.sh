#!/bin/bash
# usage (GPU box, repo root): tools/pmc_windows.sh <tag> [windows]  -> gpurun_out/<tag>_windows_pmc.json
# What bounds the lock-step window batch: wave occupancy / wait / issue counters (SQ), vector-memory instruction counts and
# L1 -> L2 request counts per kernel, separate rocprofv3 --pmc passes (never combined with trace domains).
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
TAG=${1:-win}; N=${2:-64}
PASSES=(
"SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_INSTS_VALU"
"SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_LDS SQ_INSTS_SALU SQ_INSTS_LDS SQ_INST_LEVEL_VMEM SQ_CYCLES"
"TCP_TCC_READ_REQ TCP_TCC_WRITE_REQ TCP_TOTAL_CACHE_ACCESSES TCP_PENDING_STALL_CYCLES"
"TCC_REQ TCC_HIT TCC_MISS TCC_TAG_STALL"
)
i=0
for C in "${PASSES[@]}"; do
  D=$R/gpurun_out/pmcw_$i; mkdir -p $D
  (cd /tmp && timeout 300 rocprofv3 --pmc $C --kernel-trace -f csv -d $D -o pmc -- python $R/bench.py --steps 10 --warmup 10 --no-orb --no-cpu-baseline --ba-windows $N > $D/stdout.log 2>&1) || echo "pass $i failed"
  i=$((i+1))
done
python - "$R" "$TAG" <<'PY'
import csv, glob, json, re, sys, collections
R, tag = sys.argv[1], sys.argv[2]
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(f"{R}/gpurun_out/pmcw_*/*counter_collection.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"k_batched.*?(d_[a-z_0-9]+)", n)
        name = "k_batched:" + m.group(1) if m else re.sub(r"\(anonymous namespace\)::", "", n).split("(")[0].replace("void ", "")
        acc[name][r["Counter_Name"]].append(float(r["Counter_Value"]))
out = {}
for k, d in acc.items():
    out[k] = {c: round(sum(v) / len(v), 1) for c, v in d.items()}
    out[k]["launches"] = max(len(v) for v in d.values())
json.dump(out, open(f"{R}/gpurun_out/{tag}_windows_pmc.json", "w"), indent=1)
for k in sorted(out):
    if k.startswith("k_batched") or k in ("k_linearize<true>", "k_reduce2", "k_update", "k_chol_tiles<true>"):
        d = out[k]
        wc = d.get("SQ_WAVE_CYCLES", 0) or 1
        print(k, "launches", d["launches"])
        print("   waves %.0f  wave-cycles %.3g  wait_any %.0f%%  wait_inst %.0f%%  active_inst %.0f%%  (valu %.0f%%)  busy_cycles %.3g" % (
            d.get("SQ_WAVES", 0), wc, 100 * d.get("SQ_WAIT_ANY", 0) / wc, 100 * d.get("SQ_WAIT_INST_ANY", 0) / wc,
            100 * d.get("SQ_ACTIVE_INST_ANY", 0) / wc, 100 * d.get("SQ_ACTIVE_INST_VALU", 0) / wc, d.get("SQ_BUSY_CYCLES", 0)))
        print("   insts: valu %.3g  vmem_rd %.3g  vmem_wr %.3g  lds %.3g  salu %.3g | L1->L2 read req %.3g  write req %.3g  | L2 req %.3g hit %.3g miss %.3g tag stall %.3g" % (
            d.get("SQ_INSTS_VALU", 0), d.get("SQ_INSTS_VMEM_RD", 0), d.get("SQ_INSTS_VMEM_WR", 0), d.get("SQ_INSTS_LDS", 0), d.get("SQ_INSTS_SALU", 0),
            d.get("TCP_TCC_READ_REQ", 0), d.get("TCP_TCC_WRITE_REQ", 0), d.get("TCC_REQ", 0), d.get("TCC_HIT", 0), d.get("TCC_MISS", 0), d.get("TCC_TAG_STALL", 0)))
PY
rm -rf $R/gpurun_out/pmcw_*
