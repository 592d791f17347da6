"""Timeline of the ORB leg from a rocprofv3 kernel trace: how much of the wall time has 0 / 1 / 2+ kernels in flight, per kernel
the time it runs alone vs overlapped.   python tools/orb_timeline.py <dir with *kernel_trace.csv>"""
import csv, glob, re, sys, collections
f = glob.glob(sys.argv[1] + "/*kernel_trace.csv")[0]
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), re.sub(r"\(anonymous namespace\)::|se2gpu::|void ", "", r["Kernel_Name"]).split("(")[0].split("<")[0], r.get("Queue_Id")) for r in csv.DictReader(open(f))]
rows.sort()
rows = [r for r in rows if not r[2].startswith("k_plan") ]
# steady state: the middle 60 % of the kernels named k_fast_score define the window
fs = [r for r in rows if r[2] == "k_fast_score"]
# the resident leg comes first (orb_bench.run: warm-up, then `steps` batches, 4 score launches each when pipelined by level);
# the profiled pass and the streaming leg follow - keep to launches 40 .. 200 of a run with --orb-steps 60
lo, hi = fs[40][0], fs[200][0]
rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
ev = []
for s, e, n, q in rows: ev += [(s, 1, n), (e, -1, n)]
ev.sort()
depth, last, hist = 0, ev[0][0], collections.Counter()
for t, d, n in ev:
    hist[min(depth, 3)] += t - last
    last = t
    depth += d
tot = sum(hist.values())
print("window %.2f ms, %d kernels; time with 0 / 1 / 2 / 3+ kernels in flight: %s" % (tot / 1e6, len(rows), "  ".join("%d: %.1f %%" % (k, 100 * hist[k] / tot) for k in sorted(hist))))
nb = len([r for r in rows if r[2] == "k_describe"])
print("batches in the window (k_describe launches): %d -> %.3f ms per batch; kernel time per batch %.3f ms" % (nb, tot / 1e6 / max(nb, 1), sum(e - s for s, e, n, q in rows) / 1e6 / max(nb, 1)))
dur = collections.defaultdict(list)
for s, e, n, q in rows: dur[n].append((e - s) / 1e3)
for k, v in sorted(dur.items(), key=lambda kv: -sum(kv[1]))[:16]:
    print("  %-22s n %4d  mean %8.1f us  per batch %7.1f us  queues %s" % (k[:22], len(v), sum(v) / len(v), sum(v) / max(nb, 1), sorted({q for s, e, n, q in rows if n == k})))
