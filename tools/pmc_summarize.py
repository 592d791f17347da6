#!/usr/bin/env python3
"""Summarise two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE - separate runs, TCC slots do not fit both) of bench.py
into per-kernel HBM traffic per launch.

    python tools/pmc_summarize.py gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv \
                                  gpurun_out/pmc_WRITE_SIZE/pmc_counter_collection.csv profiles/pmc_traffic.json

Units / corrections (MI355X_MICROARCH.md §HBM, re-calibrated here on known byte counts of this code base):
  * both counters are in KiB;
  * WRITE_SIZE is exact    (D2D copy of 307,232,768 B reads back as 300,032 KiB);
  * FETCH_SIZE reports 1/2 of the bytes actually fetched (same copy: 150,028 KiB; k_level0, byte loads of 78.6 MB
    of images: 41.3 MB) -> traffic = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 bytes.
"""
import collections
import csv
import json
import re
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def load(path):
    d = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        name = re.sub(r"\(anonymous namespace\)::", "", r["Kernel_Name"]).split("(")[0].replace("void ", "")
        name = re.sub(r"<.*>", "", name)
        d[name].append(float(r["Counter_Value"]))
    return d


def main():
    F, W = load(sys.argv[1]), load(sys.argv[2])
    out = {"_meta": {"unit": "bytes per launch", "formula": "(2*FETCH_SIZE + WRITE_SIZE) * 1024",
                     "command": "bench.py --steps 10 --warmup 10 --orb-batch 256 --orb-steps 2 (BA 200 KF / 20k landmarks)",
                     "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes"}}
    # what the capture is valid for: the kernel sources it ran (bench.py nulls `traffic` when they have changed since) and
    # the commit the caller says it is at (the GPU box has no .git: pass GRAFT_COMMIT=$(git rev-parse --short HEAD))
    # what the capture is valid for: every kernel's entry carries the hash of the gfx950 code object it was measured on
    # (se2lam_amd/devcode.py reads it out of libse2gpu.so; bench.py drops an entry whose kernel has changed since - and only that)
    import bench
    from se2lam_amd import devcode
    code = devcode.kernel_code_hashes()
    out["_meta"]["source_sha"] = bench.source_sha()
    out["_meta"]["commit"] = os.environ.get("GRAFT_COMMIT") or None
    for k in sorted(F, key=lambda k: -sum(F[k])):
        f = sum(F[k]) / len(F[k])
        w = sum(W.get(k, [0.0])) / max(len(W.get(k, [0.0])), 1)
        out[k] = {"launches": len(F[k]), "fetch_size_kib_avg": f, "write_size_kib_avg": w,
                  "traffic_bytes": (2 * f + w) * 1024, "code_sha": code.get(k)}
        if k == "k_window_lm" and os.environ.get("SE2_PMC_WINDOWS"):   # the resident BA kernel: one launch = this many windows' optimize(10)
            out[k]["windows_per_launch"] = int(os.environ["SE2_PMC_WINDOWS"])
    json.dump(out, open(sys.argv[3], "w"), indent=1)
    for k, v in out.items():
        if k != "_meta":
            print(f"{k:28s} {v['traffic_bytes'] / 1e6:10.3f} MB/launch")


if __name__ == "__main__":
    main()
