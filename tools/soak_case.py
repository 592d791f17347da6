"""One fresh-process soak case (tools/soak_fresh.sh runs hundreds of them): everything that must be bit-reproducible, run the
way the flaky test ran it, with the first difference printed in full.

  * direct / captured / replayed optimize() on one handle (the SE(2) model on a plain and on a trial-rejecting start, the
    SE3-expmap model, the pose graph) - tests/test_ba_gpu.py::test_repeated_optimize_replays_a_graph_with_identical_results
  * lock-step and per-stream window batches against one-by-one runs, through plan eviction
  * with SE2GPU_BA_CHOL_VERIFY=1 in the environment: every tile hand-off of the dataflow solve checked against the checksum
    its producer published (se2gpu_ba_debug_chol_verify)
Prints one line: `SOAK ok ...` or `SOAK FAIL ...` followed by the details."""
import gc
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

from se2lam_amd import optimizer as op, synth  # noqa: E402
import test_ba_gpu as T  # noqa: E402

verify = os.environ.get("SE2GPU_BA_CHOL_VERIFY") == "1"
seed = int(sys.argv[1]) if len(sys.argv) > 1 else 0
fails = []
checked = mism = 0


def vfy(o, what):
    global checked, mism
    if not verify or o.solver_path() != 0:
        return
    try:
        m, c, rec = o.chol_verify()
    except Exception:
        return
    checked += c
    if m:
        mism += m
        fails.append(f"{what}: {m} hand-off mismatches of {c}; first records (epoch, task, tile, slab|part, xcc, got, want, t): "
                     + "; ".join(str([hex(int(v)) for v in r]) for r in rec[:4]))


def opt(g, loader=None):
    o = op.SlamOptimizer()
    loader(o, g) if loader else o.load(g)
    o.initializeOptimization(0)
    return o


case, trials = T.LM_REJECT_CASES[seed % len(T.LM_REJECT_CASES)]
P = 30 + 7 * (seed % 5)
for name, g, loader in (("se2", synth.ba_graph(P, 30 * P), None), ("se2-reject", T._kidnapped(synth, *case), None),
                        ("se2-200kf", synth.ba_graph(200, 20000), None) if seed % 4 == 0 else ("se2-50kf", synth.ba_graph(50, 5000), None),
                        ("se3", synth.ba3_graph(12, 300, 2), op.load_se3_graph), ("posegraph", synth.pose_graph(40), op.load_pose_graph)):
    o = opt(g, loader)
    runs = []
    for rep in range(5):
        o.reset_estimates()
        o.optimize(7 if rep < 4 else 3)
        runs.append((o.stats["trials_hist"], o.stats["chi2_hist"], o.stats["lambda_hist"], [a.copy() for a in o.estimates()]))
    for k in (1, 2, 3):
        for f, nm in ((0, "trials"), (1, "chi2"), (2, "lambda")):
            if runs[k][f] != runs[0][f]:
                fails.append(f"{name}: run {k} {nm} differs from run 0\n   got  {runs[k][f]}\n   want {runs[0][f]}")
        for a, b, nm in zip(runs[k][3], runs[0][3], ("poses", "landmarks")):
            if not np.array_equal(a, b):
                d = np.argwhere(a != b)
                fails.append(f"{name}: run {k} {nm} differ in {len(d)} entries, first {d[0].tolist()}: {a[tuple(d[0])]!r} vs {b[tuple(d[0])]!r}")
    if runs[4][0] != runs[0][0][:3] or runs[4][1] != runs[0][1][:3]:
        fails.append(f"{name}: the 3-iteration run differs from the first 3 of the 7-iteration run")
    vfy(o, name)
    del o

for lockstep in ("1", "0"):
    # (read per process by the library: the per-stream path is exercised in its own soak runs, see soak_fresh.sh)
    if lockstep != os.environ.get("SE2GPU_BA_LOCKSTEP", "1"):
        continue
    for round_, iters in enumerate((3, 5, 2, 4, 6)):
        graphs = [synth.ba_graph(9 + round_ + k, 70 + 10 * k, seed=50 + 7 * round_ + k + 13 * seed) for k in range(6)]
        if round_ == 1:
            graphs += [T._kidnapped(synth, *c[0]) for c in T.LM_REJECT_CASES[:3]]
        ref = []
        for g in graphs:
            o = opt(g)
            o.optimize(iters)
            ref.append((o.stats, o.estimates()))
        opts = [opt(g) for g in graphs]
        for rep in range(2):
            op.reset_estimates_batch(opts) if rep else None
            op.optimize_batch(opts, iters)
            for i, (o, (st, (p, l))) in enumerate(zip(opts, ref)):
                if o.stats != st or not np.array_equal(o.estimates()[0], p) or not np.array_equal(o.estimates()[1], l):
                    fails.append(f"batch round {round_} rep {rep} window {i}: differs from its one-by-one run\n   got  {o.stats}\n   want {st}")
        for o in opts:
            vfy(o, f"batch round {round_}")
        del opts, o
        gc.collect()

if fails:
    print(f"SOAK FAIL seed {seed} verify {int(verify)}: {len(fails)} findings")
    for f in fails:
        print("  ", f)
    sys.exit(1)
print(f"SOAK ok seed {seed} verify {int(verify)} handoffs_checked {checked} mismatches {mism}")
