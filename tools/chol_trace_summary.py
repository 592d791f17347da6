"""Critical path of k_chol_tiles from a trace written by `SE2GPU_BA_CHOL_TRACE=1 python tools/chol_trace.py 200 2> trace.txt`:
    python tools/chol_trace_summary.py trace.txt
Walks back from the tile task that publishes last through the producer of its last dependency that published later (the
tasks' dependency lists are in ascending block-column order, so the last entry is the one the elimination waits for)."""
import sys

rows = [l.split() for l in open(sys.argv[1]) if l.startswith("choltrace")]
n = len(rows) // 3          # tools/chol_trace.py solves three times; the last solve is analysed
T = {}
for r in rows[-n:]:
    t, i, kind, j, dep = map(int, r[1:6])
    T[(i, kind, j)] = (dep, list(map(int, r[6:])))
tiles = {k: v for k, v in T.items() if k[1] != 2}
print("tile tasks %d, x tasks %d, last tile published at %.1f us, last x task started at %.1f us" % (
    len(tiles), len(T) - len(tiles), max(v[1][5] for v in tiles.values()) / 100, max(v[1][0] for k, v in T.items() if k[1] == 2) / 100))
k = max(tiles, key=lambda k: tiles[k][1][5])
path = []
while True:
    dep, st = tiles[k]
    path.append((k, dep, st))
    if dep < 0: break
    m, hasT = dep & 0x7fff, dep >> 15
    i, kind, j = k
    cands = [(j, 0, m)]
    if hasT: cands.append((i, kind, m))
    cands = [c if c in tiles else (c[0], 1, c[2]) for c in cands]   # the diagonal task publishes R(m, m)
    cands = [c for c in cands if c in tiles]
    if not cands: break
    k = max(cands, key=lambda c: tiles[c][1][5])
path.reverse()
prev = None
tot = dict(flag=0, loads=0, mfma=0, staging=0, elim=0, publish=0)
for (k, dep, st) in path:
    f = (st[1] - prev) / 100 if prev is not None else 0.0
    ld = (st[7] - st[1]) / 100 if dep >= 0 else 0.0
    mf = (st[2] - st[7]) / 100 if dep >= 0 else 0.0
    sg, el, pb = (st[3] - st[2]) / 100, (st[4] - st[3]) / 100, (st[5] - st[4]) / 100
    for key, v in zip(tot, (f, ld, mf, sg, el, pb)): tot[key] += v
    print(f"tile ({k[0]:2d},{k[2]:2d}){'R' if k[1] else 'A'} after column {dep & 0x7fff if dep >= 0 else -1:2d}: last slab flag {f:5.2f}  its loads + staging {ld:5.2f}  "
          f"rest of the products {mf:5.2f}  to LDS {sg:5.2f}  elimination {el:5.2f}  publish {pb:5.2f}   [us]  published at {st[5] / 100:6.1f}")
    prev = st[5]
print("chain of %d tasks: " % len(path) + "  ".join(f"{k} {v:.1f}" for k, v in tot.items()))
