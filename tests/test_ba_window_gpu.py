"""The one-workgroup-per-window path of se2gpu_ba_optimize_batch (csrc/ba_window.hip): a window lives in ONE compute unit's LDS for
its whole optimize(iters) - reduced system, dense solve and the Levenberg controller included.  Sums into the reduced system are
LDS atomics (no fixed order), so this path is held to the oracle's g2o decisions trial for trial and to its costs / poses within
tolerances far inside north_star's 1e-5 - not bit for bit as the multi-launch batch paths are (tests/test_ba_gpu.py)."""
import os

import numpy as np
import pytest

from test_ba_gpu import LM_REJECT_CASES, _kidnapped, _opt


def _merged_landmarks(synth, P=40, L=240, groups=12, per_group=3):
    """landmarks with 17 to 64 observations (the wave-per-landmark class): the generator's landmarks are seen from a dozen
    neighbouring key frames at most, so `per_group` landmarks whose observers do not overlap are fused into one - its observations
    then disagree about where it is, which the robust least squares takes like any other outlier-ridden point"""
    import copy
    g = copy.copy(synth.ba_graph(P, L))
    e_kf, e_lm = np.asarray(g.e_kf).copy(), np.asarray(g.e_lm).copy()
    seen = [set(e_kf[e_lm == l].tolist()) for l in range(g.L)]
    used, fused = set(), 0
    for a in range(g.L):
        if fused == groups:
            break
        if a in used:
            continue
        members, kfs = [a], set(seen[a])
        for b in range(a + 1, g.L):
            if b not in used and not (kfs & seen[b]):
                members.append(b)
                kfs |= seen[b]
                if len(members) == per_group:
                    break
        if len(members) == per_group:
            used.update(members)
            for b in members[1:]:
                e_lm[e_lm == b] = a
            fused += 1
    assert fused == groups
    keep = np.unique(e_lm)                       # the fused-away landmarks leave the graph
    renum = np.full(g.L, -1, np.int64)
    renum[keep] = np.arange(len(keep))
    e_lm = renum[e_lm]
    g.lms = np.asarray(g.lms)[keep].copy()
    if g.lms_true is not None:
        g.lms_true = np.asarray(g.lms_true)[keep].copy()
    order = np.argsort(e_lm, kind="stable")
    g.e_kf, g.e_lm = e_kf[order].astype(np.int32), e_lm[order].astype(np.int32)
    g.e_uv, g.e_info = np.asarray(g.e_uv)[order].copy(), np.asarray(g.e_info)[order].copy()
    return g

pytestmark = pytest.mark.gpu
RTOL = 1e-9      # cost / lambda histories against the multi-launch path and the oracle (observed: 1e-12)


@pytest.fixture
def resident():
    """every batch through the resident kernel, whatever its size (SE2GPU_BA_RESIDENT is read per call)"""
    old = os.environ.get("SE2GPU_BA_RESIDENT")
    os.environ["SE2GPU_BA_RESIDENT"] = "1"
    yield
    if old is None:
        os.environ.pop("SE2GPU_BA_RESIDENT", None)
    else:
        os.environ["SE2GPU_BA_RESIDENT"] = old


def _multi_launch(g, iters, mode=0):
    old = os.environ.get("SE2GPU_BA_RESIDENT")
    os.environ["SE2GPU_BA_RESIDENT"] = "0"
    try:
        o = _opt(g)
        o.optimize(iters, mode) if mode else o.optimize(iters)
        return o.stats, o.estimates()
    finally:
        if old is None:
            os.environ.pop("SE2GPU_BA_RESIDENT", None)
        else:
            os.environ["SE2GPU_BA_RESIDENT"] = old


def _same(o, st, est, what):
    n = st["iterations"]
    assert o.stats["iterations"] == n and o.stats["trials"] == st["trials"], what
    assert o.stats["trials_hist"] == st["trials_hist"], (what, o.stats["trials_hist"], st["trials_hist"])
    assert o.stats["terminated"] == st["terminated"] and o.stats["stopped"] == st["stopped"], what
    assert np.isclose(o.stats["chi2_init"], st["chi2_init"], rtol=RTOL), what
    assert np.allclose(o.stats["chi2_hist"][:n], st["chi2_hist"][:n], rtol=RTOL), what
    assert np.allclose(o.stats["lambda_hist"][:n], st["lambda_hist"][:n], rtol=1e-7), what
    p, l = o.estimates()
    assert np.allclose(p, est[0], rtol=1e-9, atol=1e-7) and np.allclose(l, est[1], rtol=1e-9, atol=1e-6), what


def test_resident_batch_equals_multi_launch_and_the_oracle(oracle, synth, resident):
    """windows of several sizes - landmarks with 2 to 20 observations (4-, 8-, 16-lane groups and whole waves), starts that reject
    trials - in one batch: g2o's decisions trial for trial, costs and poses to 1e-9"""
    from se2lam_amd.optimizer import optimize_batch
    graphs = [synth.ba_graph(8, 60), synth.ba_graph(21, 800), synth.ba_graph(50, 5000), synth.ba_graph(30, 2000),
              synth.ba_graph(20, 200, obs_per_lm=14.0), synth.ba_graph(40, 150, obs_per_lm=30.0), _merged_landmarks(synth)] + \
             [_kidnapped(synth, *c[0]) for c in LM_REJECT_CASES]
    kmax = [int(np.bincount(np.asarray(g.e_lm), minlength=g.L).max()) for g in graphs[:7]]
    assert max(kmax) > 16 and any(8 < k <= 16 for k in kmax), kmax                  # the wide-landmark classes are exercised
    ref = [_multi_launch(g, 10) for g in graphs]
    opts = [_opt(g) for g in graphs]
    its = optimize_batch(opts, 10)
    for g, o, (st, est), n in zip(graphs, opts, ref, its):
        assert n == st["iterations"]
        _same(o, st, est, (g.P, g.L))
    assert [r[0]["trials_hist"] for r in ref[7:]] == [c[1] for c in LM_REJECT_CASES]
    for g, o in list(zip(graphs, opts))[:2] + list(zip(graphs, opts))[7:10]:          # the oracle itself on the small ones
        poses, lms, st = oracle.ba_optimize(g, 10, 0)
        assert o.stats["trials_hist"] == st["trials_hist"]
        assert np.allclose(o.stats["chi2_hist"][:10], st["chi2_hist"][:10], rtol=1e-7)
        assert np.allclose(o.estimates()[0], poses, rtol=1e-6, atol=1e-6)


def test_resident_batch_repeats_gauss_newton_short_runs_and_the_stop_flag(synth, resident):
    from se2lam_amd.optimizer import optimize_batch, reset_estimates_batch
    graphs = [synth.ba_graph(8, 60), synth.ba_graph(30, 2000), _kidnapped(synth, *LM_REJECT_CASES[2][0]), synth.ba_graph(50, 5000)]
    opts = [_opt(g) for g in graphs]
    for mode, iters in ((0, 10), (1, 10), (0, 4), (0, 0)):
        ref = [_multi_launch(g, iters, mode) for g in graphs]
        for rep in range(2):                       # the second run starts from the reset estimates on the same handles
            reset_estimates_batch(opts)
            optimize_batch(opts, iters, mode)
            for i, (g, o, (st, est)) in enumerate(zip(graphs, opts, ref)):
                if mode == 1:
                    # Undamped Gauss-Newton amplifies the last bits of every sum: from the start that is metres off the two paths
                    # drift apart to 1e-5 within six iterations, and on the 50-key-frame window the un-damped system loses definiteness
                    # to rounding at the fifth step, which one factorisation flags and the other does not (g2o would take either).
                    # Gauss-Newton is compared over its first steps only.
                    assert np.allclose(o.stats["chi2_hist"][:3], st["chi2_hist"][:3], rtol=1e-8), (g.P, g.L)
                    continue
                _same(o, st, est, (g.P, g.L, mode, iters, rep))
    # a window optimised alone (multi-launch path, its own stream) right after a resident batch, and the other way round
    reset_estimates_batch(opts)
    optimize_batch(opts, 10)
    a = opts[1].estimates()
    opts[1].reset_estimates()
    os.environ["SE2GPU_BA_RESIDENT"] = "0"
    opts[1].optimize(10)
    os.environ["SE2GPU_BA_RESIDENT"] = "1"
    assert np.allclose(opts[1].estimates()[0], a[0], rtol=1e-9, atol=1e-7)
    # SparseOptimizer::setForceStopFlag raised from the start: no iteration, the starting cost reported
    stop = np.ones(1, np.uint8)
    reset_estimates_batch(opts)
    its = optimize_batch(opts, 10, 0, stop)
    assert its == [0] * len(opts) and all(o.stats["stopped"] for o in opts)
    assert all(np.isclose(o.stats["chi2_final"], o.stats["chi2_init"]) for o in opts)


def test_windows_too_large_for_the_lds_fall_back(synth, resident):
    """a batch with a window of 80 key frames (its reduced system does not fit 160 KiB) runs on the multi-launch paths as a whole"""
    from se2lam_amd.optimizer import optimize_batch
    graphs = [synth.ba_graph(8, 60), synth.ba_graph(80, 1500)]
    ref = [_multi_launch(g, 6) for g in graphs]
    opts = [_opt(g) for g in graphs]
    optimize_batch(opts, 6)
    for o, (st, est) in zip(opts, ref):
        assert o.stats == st and np.array_equal(o.estimates()[0], est[0])       # (the multi-launch paths are bit-identical)


def test_default_threshold_keeps_small_batches_on_the_multi_launch_paths(synth):
    """without the switch a batch below SE2GPU_BA_RESIDENT_MIN windows stays bit-identical to one-by-one runs, a larger one takes
    the resident kernel (same decisions, costs to 1e-9)"""
    from se2lam_amd.optimizer import optimize_batch
    os.environ.pop("SE2GPU_BA_RESIDENT", None)
    g = synth.ba_graph(21, 800)
    st, est = _multi_launch(g, 8)
    small = [_opt(g) for _ in range(3)]
    optimize_batch(small, 8)
    assert all(o.stats == st and np.array_equal(o.estimates()[0], est[0]) for o in small)
    large = [_opt(g) for _ in range(100)]
    optimize_batch(large, 8)
    for o in large:
        _same(o, st, est, "100 windows")


def test_a_window_of_the_128_thread_class_keeps_the_batch_off_the_resident_path_unless_forced(synth):
    """a window with 61+ free key frames only fits the 128-thread workgroup and would take twice as long there as the whole batch
    takes in lock step: by default such a batch is left to the multi-launch paths (bit-identical to one-by-one runs); forced, the
    kernel takes it (same decisions, costs to 1e-9)"""
    from se2lam_amd import capi
    from se2lam_amd.optimizer import optimize_batch
    small, big = synth.ba_graph(8, 60), synth.ba_graph(62, 400)
    refs = [_multi_launch(small, 5), _multi_launch(big, 5)]
    os.environ.pop("SE2GPU_BA_RESIDENT", None)
    opts = [_opt(small) for _ in range(99)] + [_opt(big)]
    optimize_batch(opts, 5)
    assert capi.lib().se2gpu_ba_last_batch_path() != 2
    for o in opts[:3]:
        assert o.stats == refs[0][0] and np.array_equal(o.estimates()[0], refs[0][1][0])
    assert opts[-1].stats == refs[1][0] and np.array_equal(opts[-1].estimates()[0], refs[1][1][0])
    os.environ["SE2GPU_BA_RESIDENT"] = "1"
    try:
        forced = [_opt(small), _opt(big)]
        optimize_batch(forced, 5)
        assert capi.lib().se2gpu_ba_last_batch_path() == 2
        _same(forced[0], *refs[0], "small window, forced")
        _same(forced[1], *refs[1], "62 key frames, forced")
    finally:
        os.environ.pop("SE2GPU_BA_RESIDENT", None)


def test_resident_edge_cases(synth, resident):
    """windows at the edges of what the kernel meets: two key frames (configs[0]'s first local BA), no odometry edge, a single free
    pose, every pose fixed (only the landmarks move), landmarks with a single observation"""
    import copy
    from se2lam_amd.optimizer import optimize_batch
    base = synth.ba_graph(8, 60)
    two = synth.ba_graph(2, 40)
    no_odo = copy.copy(base)
    no_odo.o_i, no_odo.o_j = base.o_i[:0], base.o_j[:0]
    no_odo.o_meas, no_odo.o_info = base.o_meas[:0], base.o_info[:0]
    one_free = copy.copy(base)
    one_free.fixed = np.ones_like(base.fixed)
    one_free.fixed[3] = 0
    all_fixed = copy.copy(base)
    all_fixed.fixed = np.ones_like(base.fixed)
    single = copy.copy(synth.ba_graph(12, 120))
    keep = np.ones(single.E, bool)
    e_lm = np.asarray(single.e_lm)
    for l in range(0, single.L, 7):                       # every seventh landmark keeps only its first observation
        idx = np.nonzero(e_lm == l)[0]
        keep[idx[1:]] = False
    single.e_kf, single.e_lm = np.asarray(single.e_kf)[keep], e_lm[keep]
    single.e_uv, single.e_info = np.asarray(single.e_uv)[keep], np.asarray(single.e_info)[keep]
    graphs = [two, no_odo, one_free, all_fixed, single]
    ref = [_multi_launch(g, 6) for g in graphs]
    opts = [_opt(g) for g in graphs]
    optimize_batch(opts, 6)
    from se2lam_amd import capi
    assert int(capi.lib().se2gpu_ba_last_batch_path()) == 2
    for g, o, (st, est) in zip(graphs, opts, ref):
        _same(o, st, est, (g.P, g.L, g.E, int(np.asarray(g.fixed).sum())))


def test_a_batch_of_distinct_windows_over_both_workgroup_widths(synth, resident):
    """24 windows of 30-60 key frames (distinct sizes and seeds, some starts that reject trials): the batch is dealt to the 512- and
    the 256-thread launches, every window decides like its multi-launch run, costs and poses to 1e-9"""
    from se2lam_amd import capi
    from se2lam_amd.optimizer import optimize_batch
    graphs = synth.mixed_windows(24, p_range=(30, 60), l_range=(300, 900), kidnapped_every=5)
    assert min(g.P for g in graphs) < 40 and max(g.P for g in graphs) >= 58          # both widths are in the batch
    ref = [_multi_launch(g, 8) for g in graphs]
    opts = [_opt(g) for g in graphs]
    optimize_batch(opts, 8)
    assert int(capi.lib().se2gpu_ba_last_batch_path()) == 2
    for g, o, (st, est) in zip(graphs, opts, ref):
        _same(o, st, est, (g.P, g.L, g.E))


def test_a_window_without_room_for_the_record_copy_is_left_to_the_other_paths(synth, resident):
    """the kernel keeps its list of landmarks and its copy of the observations in the idle W buffer of the multi-launch path (72 B per
    observation): a window with far more landmarks than observations has no room there - the call runs it elsewhere, bit for bit
    (unless the handle comes out of the library's pool with a buffer that grew on an earlier, larger window)"""
    import copy
    from se2lam_amd import capi
    from se2lam_amd.optimizer import optimize_batch
    g = copy.copy(synth.ba_graph(10, 400))
    e_lm = np.asarray(g.e_lm)
    keep = e_lm % 8 == 0                                   # seven landmarks in eight lose all their observations
    g.e_kf, g.e_lm = np.asarray(g.e_kf)[keep], e_lm[keep]
    g.e_uv, g.e_info = np.asarray(g.e_uv)[keep], np.asarray(g.e_info)[keep]
    assert 16 * g.L + 44 * int(keep.sum()) > 72 * int(keep.sum())
    st, est = _multi_launch(g, 6)
    o = _opt(g)
    optimize_batch([o], 6)
    if int(capi.lib().se2gpu_ba_last_batch_path()) == 2:
        _same(o, st, est, "a handle from the pool whose buffer had grown on a larger window before: room after all")
    else:
        assert o.stats == st and np.array_equal(o.estimates()[0], est[0])
