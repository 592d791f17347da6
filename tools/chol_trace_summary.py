"""Per-block-column critical path of k_chol_tiles from a trace written by `SE2GPU_BA_CHOL_TRACE=1 python tools/chol_trace.py
200 2> trace.txt`:   python tools/chol_trace_summary.py trace.txt"""
import sys

rows = [l.split() for l in open(sys.argv[1]) if l.startswith("choltrace")]
n = len(rows) // 3          # tools/chol_trace.py solves three times; the last solve is analysed
T = {}
for r in rows[-n:]:
    t, i, isr, j = map(int, r[1:5])
    T[(i, isr, j)] = list(map(int, r[5:]))
nb = max(j for (_, _, j) in T) + 1
print("total %.1f us" % (max(st[5] for st in T.values()) / 100))
prev = 0
for j in range(nb):
    col = {k: v for k, v in T.items() if k[2] == j}
    k = max(col, key=lambda k: col[k][5])   # the task that publishes last
    st = col[k]
    first = min(v[5] for v in col.values())
    print(f"col {j:2d} last publisher (tile row {k[0]:2d}, {'R' if k[1] else 'A'}): flag {(st[1] - prev) / 100 if j else 0:5.2f}  "
          f"operand loads {(st[7] - st[1]) / 100 if j else 0:5.2f}  MFMA products {(st[2] - st[7]) / 100 if j else 0:5.2f}  "
          f"staging {(st[3] - st[2]) / 100:5.2f}  elimination {(st[4] - st[3]) / 100:5.2f}  publish {(st[5] - st[4]) / 100:5.2f}  "
          f"| spread of the column's publish times {(st[5] - first) / 100:5.2f}   [us]")
    prev = st[5]
