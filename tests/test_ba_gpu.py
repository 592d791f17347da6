"""GPU parity tests of the HIP bundle adjustment (through the C ABI) against the CPU oracle.

Tolerance (BASELINE.json north_star): BA cost and pose updates within 1e-5 relative.  The kernels
sum J^T W J in a different order than the oracle, so FP64 results agree to ~1e-12 on one
linearisation and to ~1e-9 after 10 LM iterations; the asserted bounds are the 1e-5 of the
north star or tighter.
"""
import os
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))

REL = 1e-5


def _opt(g):
    from se2lam_amd.optimizer import SlamOptimizer
    o = SlamOptimizer()
    o.load(g)
    o.initializeOptimization(0)
    return o


def _pose_update_close(p_gpu, p_ref, p0, rel=REL):
    """pose UPDATES (estimate - initial) agree within `rel` relative (norm-wise per component)."""
    d_gpu, d_ref = p_gpu - p0, p_ref - p0
    for c in range(3):
        scale = np.abs(d_ref[:, c]).max() + 1e-12
        assert np.abs(d_gpu[:, c] - d_ref[:, c]).max() <= rel * scale, (c, np.abs(d_gpu[:, c] - d_ref[:, c]).max(), scale)


def test_chi2_matches_oracle(oracle, synth):
    for P, L in ((8, 60), (50, 5000)):
        g = synth.ba_graph(P, L)
        o = _opt(g)
        assert o.activeRobustChi2() == pytest.approx(oracle.ba_chi2(g), rel=1e-12)


def test_reduced_system_matches_oracle(oracle, synth):
    for P, L, lam in ((8, 60, 0.0), (8, 60, 3.0), (50, 5000, 17.5)):
        g = synth.ba_graph(P, L)
        o = _opt(g)
        S, bs = o.reduced_system(lam)
        ref = oracle.ba_reduced_system(g, lam)
        scale = np.abs(ref["S"]).max()
        assert np.abs(S - ref["S"]).max() <= 1e-11 * scale
        assert np.abs(bs - ref["bs"]).max() <= 1e-11 * np.abs(ref["bs"]).max()
        assert np.abs(S - S.T).max() <= 1e-12 * scale


@pytest.mark.parametrize("P,L", [(8, 60), (50, 5000)])
def test_lm_10_iterations_match_oracle(oracle, synth, P, L):
    """config 3: localBA 50 KF / 5k landmarks / ~30k EdgeSE2XYZ, 10 iterations, cost within 1e-5."""
    g = synth.ba_graph(P, L)
    o = _opt(g)
    assert o.optimize(10) == 10
    p_ref, l_ref, st = oracle.ba_optimize(g, 10, 0)
    s = o.stats
    assert s["trials_hist"] == st["trials_hist"]
    assert np.allclose(s["chi2_hist"], st["chi2_hist"], rtol=REL, atol=0)
    assert np.allclose(s["lambda_hist"], st["lambda_hist"], rtol=REL, atol=0)
    assert s["chi2_init"] == pytest.approx(st["chi2_init"], rel=1e-12)
    assert s["chi2_final"] == pytest.approx(st["chi2_final"], rel=REL)
    poses, lms = o.estimates()
    _pose_update_close(poses, p_ref, g.poses)
    dl = np.abs((lms - g.lms) - (l_ref - g.lms)).max()
    assert dl <= REL * np.abs(l_ref - g.lms).max()
    # the estimate the library holds reproduces its reported cost
    assert o.activeRobustChi2() == pytest.approx(s["chi2_final"], rel=1e-12)


def test_global_window_200kf_20k_landmarks(oracle, synth):
    """config 4 formulation on one GPU: 200 KF / 20k landmarks / ~120k edges."""
    g = synth.ba_graph(200, 20000)
    o = _opt(g)
    o.optimize(10)
    p_ref, l_ref, st = oracle.ba_optimize(g, 10, 0)
    assert o.stats["trials_hist"] == st["trials_hist"]
    assert np.allclose(o.stats["chi2_hist"], st["chi2_hist"], rtol=REL, atol=0)
    poses, _ = o.estimates()
    _pose_update_close(poses, p_ref, g.poses)


def test_gauss_newton_mode(oracle, synth):
    g = synth.ba_graph(8, 60)
    o = _opt(g)
    o.optimize(3, mode=1)
    p_ref, l_ref, st = oracle.ba_optimize(g, 3, 1)
    assert np.allclose(o.stats["chi2_hist"], st["chi2_hist"], rtol=REL)


def test_reference_call_surface_equals_bulk_load(oracle, synth):
    """Graph built call by call as Map::loadLocalGraph does (Map.cpp:891-1053) == bulk load."""
    from se2lam_amd import optimizer as op
    g = synth.ba_graph(8, 60)
    opt = op.SlamOptimizer()
    K = np.array([[g.fx, 0, g.cx], [0, g.fx, g.cy], [0, 0, 1]], np.float32)
    op.addCamPara(opt, K, 0)
    op.setExtParameter(opt, g.Rbc, g.tbc)
    nLocal = g.P
    maxKFid = nLocal + 0 + 1                                   # Map.cpp:972
    for i in range(g.P):
        op.addVertexSE2(opt, g.poses[i], i, bool(g.fixed[i]))  # Map.cpp:925-930
    for k in range(g.O):
        op.addEdgeSE2(opt, g.o_meas[k], int(g.o_i[k]), int(g.o_j[k]), g.o_info[k])
    for l in range(g.L):
        op.addVertexSBAXYZ(opt, g.lms[l], maxKFid + l)         # Map.cpp:985-988
    for k in range(g.E):
        w = g.e_info[k]
        op.addEdgeSE2XYZ(opt, g.e_uv[k], int(g.e_kf[k]), maxKFid + int(g.e_lm[k]),
                         [[w[0], w[1]], [w[1], w[2]]], g.huber)
    opt.initializeOptimization(0)
    opt.optimize(5)
    ref = _opt(g)
    ref.optimize(5)
    assert opt.stats["chi2_hist"] == ref.stats["chi2_hist"]
    for i in range(g.P):
        assert np.array_equal(op.estimateVertexSE2(opt, i), ref.estimates()[0][i])
    assert np.array_equal(op.estimateVertexSBAXYZ(opt, maxKFid + 7), ref.estimates()[1][7])


def test_force_stop_flag(synth):
    """setForceStopFlag(&mbAbortBA) (LocalMapper.cpp:246): a raised flag stops before iteration 0."""
    g = synth.ba_graph(8, 60)
    o = _opt(g)
    flag = np.ones(1, np.uint8)
    o.setForceStopFlag(flag)
    assert o.optimize(10) == 0
    assert o.stats["stopped"]
    poses, _ = o.estimates()
    assert np.array_equal(poses, g.poses)
    flag[0] = 0
    assert o.optimize(2) == 2


def test_edge_cases(oracle, synth):
    import dataclasses
    g = synth.ba_graph(8, 60)
    # (a) no odometry edges
    g0 = dataclasses.replace(g, o_i=g.o_i[:0], o_j=g.o_j[:0], o_meas=g.o_meas[:0], o_info=g.o_info[:0])
    o = _opt(g0)
    o.optimize(4)
    _, _, st = oracle.ba_optimize(g0, 4, 0)
    assert np.allclose(o.stats["chi2_hist"], st["chi2_hist"], rtol=REL)
    # (b) several fixed poses (reference KFs are all fixed, Map.cpp:959-970)
    fx = g.fixed.copy()
    fx[[0, 3, 7]] = 1
    g1 = dataclasses.replace(g, fixed=fx)
    o = _opt(g1)
    o.optimize(4)
    p_ref, _, st = oracle.ba_optimize(g1, 4, 0)
    assert np.allclose(o.stats["chi2_hist"], st["chi2_hist"], rtol=REL)
    poses, _ = o.estimates()
    assert np.array_equal(poses[[0, 3, 7]], g.poses[[0, 3, 7]])
    # (c) a landmark nobody observes and a single-observation landmark
    keep = (g.e_lm != 5) & ~((g.e_lm == 6) & (np.cumsum(g.e_lm == 6) > 1))
    g2 = dataclasses.replace(g, e_kf=g.e_kf[keep], e_lm=g.e_lm[keep], e_uv=g.e_uv[keep], e_info=g.e_info[keep])
    o = _opt(g2)
    o.optimize(3)
    # an unobserved landmark has Hll = 0: g2o would invert a singular block; LM damping keeps it finite
    _, l_ref, st = oracle.ba_optimize(g2, 3, 0)
    assert np.allclose(o.stats["chi2_hist"], st["chi2_hist"], rtol=REL)


def test_error_codes(synth):
    from se2lam_amd import capi, optimizer as op
    opt = op.SlamOptimizer()
    with pytest.raises(capi.Se2GpuError) as e:
        opt.optimize(1)
    assert e.value.code == capi.ERR_STATE
    op.addVertexSE2(opt, [0, 0, 0], 1, True)
    with pytest.raises(capi.Se2GpuError) as e:
        op.addVertexSE2(opt, [0, 0, 0], 1, True)
    assert e.value.code == capi.ERR_INVALID
    with pytest.raises(capi.Se2GpuError):
        op.addEdgeSE2XYZ(opt, [0, 0], 1, 99, np.eye(2), 1.0)
    with pytest.raises(capi.Se2GpuError) as e:
        opt.initializeOptimization(0)  # no camera parameters
    assert e.value.code == capi.ERR_STATE


def test_landmark_sharded_two_ranks_on_one_gpu(oracle, synth):
    """The multi-GPU path with world=2 on ONE device: two handles, each holding a
    landmark shard, run their LM loops in two threads; the all-reduce callback sums the fused buffer
    [S | bs | scalars] over the two handles through host memory.  Result == single-handle run."""
    _sharded_equals_single(synth.ba_graph(8, 60), 6)


def test_landmark_sharded_lm_schedule_on_a_start_that_rejects(synth):
    """computeScale() of the sharded run must include every rank's x.b_p term - on an easy graph
    every step hits the 1/3 clamp of the lambda schedule and a wrong gain denominator goes unnoticed.  A kidnapped start
    (rejected trials, gain ratios anywhere in (0, 1)) pins rho, the lambda history and the trial counts to the
    single-handle run."""
    case, trials = LM_REJECT_CASES[2]
    _sharded_equals_single(_kidnapped(synth, *case), 10, trials)


def _sharded_equals_single(g, iters, trials=None, world=2, debug_lam=None):
    """`world` landmark shards as `world` handles on `world` threads of ONE GPU; the all-reduce callback (no caller-owned
    buffer, so the library hands it the PACKED exchange: the lower-triangular tiles of [S; b^T], se2gpu_ba_exchange_doubles)
    sums the ranks' buffers on the host.  Same collective pattern as the RCCL path, which takes this buffer too."""
    from se2lam_amd import capi
    from se2lam_amd.optimizer import SlamOptimizer
    single = _opt(g)
    single.optimize(iters)
    if trials is not None:
        assert single.stats["trials_hist"] == trials
    # the packed exchange of the system the solver factorises: the ranks of a sharded run re-order the poses like the
    # single handle does (nested dissection from the merged block pattern), so its size comes from the handle
    packed = single.exchange_doubles()
    assert packed >= int(capi.lib().se2gpu_ba_exchange_doubles(g.P))      # = in the natural order, padded partitions otherwise
    rows = 3 * g.P + 1
    rect = rows * (-(-rows // 32) * 32)              # the rows of [S; b^T] the rectangular exchange would ship (ld = 32-padded)
    if packed == int(capi.lib().se2gpu_ba_exchange_doubles(g.P)):
        assert packed <= rect and (packed < rect or rows <= 32)   # the packed triangle is smaller from two tile rows on
    barrier = threading.Barrier(world)
    stage = [None] * world
    results = [None] * world
    dbg = [None] * world
    counts = [set() for _ in range(world)]
    errors = []

    def make_cb(rank):
        def cb(ptr, count, stream):
            counts[rank].add(int(count))
            capi.check(capi.lib().se2gpu_device_synchronize())
            buf = np.empty(count)
            capi.check(capi.lib().se2gpu_memcpy_d2h(capi.vp(buf), ptr, buf.nbytes))
            stage[rank] = buf
            barrier.wait()
            tot = stage[0].copy()                    # fixed summation order on every rank: bit-identical sums
            for r in range(1, world):
                tot += stage[r]
            barrier.wait()
            capi.check(capi.lib().se2gpu_memcpy_h2d(ptr, capi.vp(tot), tot.nbytes))
        return cb

    def run(rank):
        try:
            o = SlamOptimizer()
            o.set_shard(rank, world)
            o.set_allreduce(make_cb(rank))
            o.load(g.shard(rank, world))
            o.initializeOptimization(0)
            assert o.exchange_doubles() == packed     # every rank chose the single handle's order
            if debug_lam is not None:                 # the debug entry points are collectives too
                dbg[rank] = (o.reduced_system(debug_lam), o.solve(debug_lam))
            o.optimize(iters)
            assert o.solver_path() == 0
            results[rank] = (o.stats, o.estimates())
        except Exception as exc:  # pragma: no cover
            errors.append(exc)
            barrier.abort()

    ts = [threading.Thread(target=run, args=(r,)) for r in range(world)]
    [t.start() for t in ts]
    [t.join() for t in ts]
    assert not errors, errors
    for r in range(world):
        st, (poses, _) = results[r]
        assert st["trials_hist"] == single.stats["trials_hist"]
        assert np.allclose(st["chi2_hist"], single.stats["chi2_hist"], rtol=1e-9)
        assert np.allclose(st["lambda_hist"], single.stats["lambda_hist"], rtol=1e-9)
        assert np.allclose(poses, single.estimates()[0], rtol=1e-8, atol=1e-8)
        assert np.array_equal(poses, results[0][1][0])  # replicated poses stay bit-identical across the ranks
        # the system exchange went through the packed triangle (the other exchanges are 4 scalars / world slots / 3P diagonals)
        assert packed in counts[r] and (packed >= rect or debug_lam is not None or not any(c >= rect for c in counts[r])), sorted(counts[r])
    if debug_lam is not None:   # se2gpu_ba_debug_reduced_system / _solve on a sharded handle whose poses were re-ordered
        fresh = _opt(g)
        S0, b0 = fresh.reduced_system(debug_lam)
        x0, ok0 = fresh.solve(debug_lam)
        assert ok0
        for r in range(world):
            (S, b), (x, ok) = dbg[r]
            assert ok and np.abs(S - S0).max() <= 1e-10 * np.abs(S0).max() and np.abs(b - b0).max() <= 1e-10 * np.abs(b0).max()
            assert np.abs(x - x0).max() <= 1e-8 * np.abs(x0).max()
    # every landmark lives on exactly one rank: the shards' landmark estimates together are the single run's
    lms = np.full_like(single.estimates()[1], np.nan)
    for r in range(world):
        own = g.shard_landmarks(r, world)
        lms[own] = results[r][1][1]
    assert np.allclose(lms, single.estimates()[1], rtol=1e-8, atol=1e-6)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 4, 8])
def test_landmark_sharded_config4_packed_exchange(synth, world):
    """VERDICT r02 missing #2: BASELINE config 4 (200 KF / 20,000 landmarks) landmark-sharded over 2 / 4 / 8 ranks on a
    device, through the PACKED exchange (k_tri_pack -> all-reduce -> unpack): trial counts equal, chi^2 / lambda histories
    within 1e-9 of the single-GPU run, replicated poses bit-identical across ranks (Map.cpp:891-1053)."""
    _sharded_equals_single(synth.ba_graph(200, 20000), 4, world=world)


@pytest.mark.gpu
def test_sharded_debug_entry_points_reduce_the_reordered_system(synth):
    """with the nested-dissection order the system has nsys > 3 P rows and its right-hand side sits in
    row nsys; se2gpu_ba_debug_reduced_system / _solve summed only (3 P + 1) rows over the ranks.  A loop of 120 key frames
    (re-ordered: the packed exchange is larger than the natural one) over 2 shards against the single handle."""
    from se2lam_amd import capi
    g = synth.ba_graph(120, 6000)
    o = _opt(g)
    assert o.exchange_doubles() > int(capi.lib().se2gpu_ba_exchange_doubles(g.P)), "expected a re-ordered (padded) system"
    _sharded_equals_single(g, 3, world=2, debug_lam=7.0)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [4, 8])
def test_landmark_sharded_rejecting_start_many_ranks(synth, world):
    """the 50-KF kidnapped start (rejected trials, gain ratios anywhere in (0, 1)) over 4 and 8 shards"""
    case, trials = LM_REJECT_CASES[2]
    _sharded_equals_single(_kidnapped(synth, *case), 10, trials, world=world)


def test_reserve_parks_a_warm_handle(oracle, synth):
    """se2gpu_ba_reserve(P, L, E) (start-up pre-warm): a throw-away window of that size is run and its handle parked; the
    optimizer constructed next gets it - and behaves like a new one (results equal the oracle's, nothing of the synthetic
    window left)."""
    from se2lam_amd import capi
    g = synth.ba_graph(8, 60)
    capi.check(capi.lib().se2gpu_ba_reserve(g.P, g.L, g.E))
    o = _opt(g)
    o.optimize(6)
    _, _, st = oracle.ba_optimize(g, 6, 0)
    assert np.allclose(o.stats["chi2_hist"], st["chi2_hist"], rtol=1e-9)
    with pytest.raises(capi.Se2GpuError):
        capi.check(capi.lib().se2gpu_ba_reserve(1, 0, 0))


def test_edge_information_on_device(oracle, synth):
    """Map::loadLocalGraph's per-observation information (Map.cpp:1024-1049) computed on the GPU."""
    from se2lam_amd import optimizer as op
    from test_ba_oracle import _info_inputs
    for P, L in ((12, 200), (50, 5000)):
        inp, g, _ = _info_inputs(synth, P, L)
        got = op.edge_information(**inp)
        ref = oracle.ba_edge_information(**inp)
        assert got.shape == ref.shape == (g.E, 2, 2)
        assert np.abs(got - ref).max() <= 1e-12 * np.abs(ref).max()
    assert op.edge_information(inp["lc"][:0], inp["lw"][:0], inp["e_kf"][:0], inp["sigma2"][:0], inp["Rcw"],
                               inp["twb_xy"], inp["fx"]).shape == (0, 2, 2)


@pytest.mark.parametrize("path", ["tiles", "steps"])
@pytest.mark.parametrize("P,L", [(8, 60), (11, 300), (21, 600), (32, 900), (43, 1200), (50, 5000), (200, 20000)])
def test_device_pose_solve_matches_host_cholesky(synth, monkeypatch, path, P, L):
    """The device factorisation that stands in for CHOLMOD (k_chol_tiles: one dataflow launch; k_chol_step: one launch
    per block column) against a host Cholesky refined in extended precision; solved twice (flag epochs).  The sizes
    cover the layouts of the augmented right-hand-side row: inside the last diagonal tile (3P % 32 != 0), as its last
    row (3P = 63) and in a tile row of its own (3P = 96)."""
    if path == "steps":
        monkeypatch.setenv("SE2GPU_BA_CHOL", "steps")
    else:
        monkeypatch.delenv("SE2GPU_BA_CHOL", raising=False)
    g = synth.ba_graph(P, L)
    o = _opt(g)
    for lam in (30.0, 3.0):
        S, bs = o.reduced_system(lam)
        c = np.linalg.cholesky(S)
        solve = lambda r: np.linalg.solve(c.T, np.linalg.solve(c, r))
        x_ref = solve(bs).astype(np.longdouble)
        for _ in range(3):
            x_ref = x_ref + solve((bs.astype(np.longdouble) - S.astype(np.longdouble) @ x_ref).astype(np.float64))
        for _ in range(2):
            x, ok = o.solve(lam)
            assert ok
            assert np.abs(x - x_ref.astype(np.float64)).max() <= 1e-9 * np.abs(x_ref).max()


@pytest.mark.parametrize("path", ["tiles", "steps"])
def test_device_pose_solve_reports_indefinite_system(synth, monkeypatch, path):
    """A non-positive pivot must come back as factor_ok = False (the LM controller then retries with a larger lambda,
    as g2o does when CHOLMOD fails) - never as a hang or a silent wrong step."""
    if path == "steps":
        monkeypatch.setenv("SE2GPU_BA_CHOL", "steps")
    else:
        monkeypatch.delenv("SE2GPU_BA_CHOL", raising=False)
    g = synth.ba_graph(50, 5000)
    o = _opt(g)
    S, _ = o.reduced_system(0.0)
    x, ok = o.solve(-2.0 * float(np.abs(np.diag(S)).max()))   # S - 2 max(diag) I is negative definite
    assert not ok
    x, ok = o.solve(10.0)                                      # and the next solve is clean again
    assert ok and np.isfinite(x).all()


def test_dataflow_solve_with_more_tiles_than_resident_workgroups(synth, monkeypatch):
    """500 key frames: 47 x 48 = 2,256 tile tasks, several times what the GPU keeps resident.  The dataflow launch
    must still complete (tasks only wait for lower workgroup ids, dispatch is in order) and be exact."""
    monkeypatch.delenv("SE2GPU_BA_CHOL", raising=False)
    g = synth.ba_graph(500, 10000)
    o = _opt(g)
    S, bs = o.reduced_system(20.0)
    c = np.linalg.cholesky(S)
    x_ref = np.linalg.solve(c.T, np.linalg.solve(c, bs))
    for _ in range(2):
        x, ok = o.solve(20.0)
        assert ok
        assert np.abs(x - x_ref).max() <= 1e-9 * np.abs(x_ref).max()


def _kidnapped(synth, P, L, dxy, dth, nbad, seed):
    """A few key frames displaced by metres / tens of degrees (landmarks untouched, so no point comes near a camera
    plane): the Gauss-Newton-like first steps overshoot and Levenberg-Marquardt has to REJECT trials."""
    import copy
    g = copy.copy(synth.ba_graph(P, L))   # the generator caches its graphs: never modify the shared instance
    rng = np.random.default_rng(seed)
    g.poses = g.poses.copy()
    idx = rng.choice(np.arange(1, g.P), nbad, replace=False)
    g.poses[idx, :2] += rng.normal(0, dxy, (nbad, 2))
    g.poses[idx, 2] += rng.normal(0, dth, nbad)
    # headings stay in [-pi, pi): the reference's Se2 normalises on construction (src/Config.cpp), and an out-of-range
    # theta would meet PreEdgeSE2's missing angle wrap (EdgeSE2XYZ.h:80) as a 2 pi cost jump at the first oplus
    g.poses[:, 2] = (g.poses[:, 2] + np.pi) % (2 * np.pi) - np.pi
    return g


# (P, L, dxy [mm], dtheta [rad], displaced key frames, seed) -> the trials per iteration of g2o's policy on that start.
# Chosen (tools-free search over seeds) so that EVERY accept / reject decision has a wide margin: |rho| >= 0.4 in all
# trials, where rho = 0 is the decision boundary - the last bits of the host libm cannot flip one - and no observed
# point is closer than 0.9 m to its camera plane at the start.
LM_REJECT_CASES = [
    ((8, 60, 2000.0, 0.8, 1, 4), [1, 1, 6, 5, 1, 3, 2, 3, 4, 3]),
    ((8, 60, 3000.0, 0.3, 4, 6), [1, 7, 1, 1, 1, 1, 1, 1, 1, 1]),
    ((21, 800, 3000.0, 0.3, 4, 13), [1, 1, 1, 1, 1, 8, 1, 1, 4, 1]),
    ((50, 5000, 3000.0, 0.3, 4, 12), [1, 7, 1, 1, 3, 2, 4, 3, 3, 3]),
    ((50, 5000, 3000.0, 0.3, 2, 23), [1, 1, 7, 1, 1, 3, 3, 2, 3, 4]),
]


@pytest.mark.parametrize("case,trials", LM_REJECT_CASES)
def test_lm_rejected_trials_follow_the_oracle(oracle, synth, case, trials):
    """The retry path (new lambda on the same linearisation: k_schur_lm -> k_reduce2 -> solve -> evaluate, lambda *= ni,
    ni *= 2, <= 10 trials) takes the oracle's g2o decisions trial for trial, on fixed starts that
    are known to reject - no search, no skip."""
    g = _kidnapped(synth, *case)
    p_ref, l_ref, st = oracle.ba_optimize(g, 10, 0)
    assert st["trials_hist"] == trials                       # the fixture itself: fails loudly if a host ever disagrees
    assert np.abs(st["rho_log"]).min() > 0.2
    o = _opt(g)
    assert o.optimize(10) == st["iterations"]
    assert o.stats["trials_hist"] == trials
    assert o.stats["trials"] == st["trials"] and bool(o.stats["terminated"]) == st["terminated"]
    assert np.allclose(o.stats["chi2_hist"], st["chi2_hist"], rtol=REL, atol=0)
    assert np.allclose(o.stats["lambda_hist"], st["lambda_hist"], rtol=REL, atol=0)
    poses, _ = o.estimates()
    _pose_update_close(poses, p_ref, g.poses, rel=1e-4)


def test_pooled_handles_behave_like_new_ones(synth, monkeypatch):
    """se2gpu_ba_destroy parks handles and se2gpu_ba_create hands them out again (LocalMapper::localBA constructs its
    SlamOptimizer per call): windows of different sizes through recycled handles give exactly the results of handles
    that were never used before, cached estimates are refreshed after every optimize, sparse / negative vertex ids
    still work."""
    from se2lam_amd import optimizer as op
    graphs = [synth.ba_graph(50, 5000), synth.ba_graph(8, 60), synth.ba_graph(21, 800), synth.ba_graph(8, 60)]
    monkeypatch.setenv("SE2GPU_BA_POOL", "1")
    pooled = []
    for g in graphs:                         # each optimizer is destroyed before the next one is created
        o = _opt(g)
        o.optimize(4)
        p4, h4 = o.estimates()[0].copy(), list(o.stats["chi2_hist"])
        o.optimize(3)                        # the host copy of the estimates must follow
        pooled.append((o.stats["chi2_hist"], p4, o.estimates()[0].copy(), o.estimates()[1].copy(), h4))
        assert not np.array_equal(p4, pooled[-1][2])
        del o
    keep = []                                # handles that are all alive at once: none of them comes from the pool twice
    for g, ref in zip(graphs, pooled):
        o = _opt(g)
        keep.append(o)
        o.optimize(4)
        p4 = o.estimates()[0].copy()
        o.optimize(3)
        assert o.stats["chi2_hist"] == ref[0]
        assert np.array_equal(p4, ref[1]) and np.array_equal(o.estimates()[0], ref[2])
        assert np.array_equal(o.estimates()[1], ref[3])
    # ids outside the dense table (negative, huge) go through the map
    g = graphs[1]
    o = op.SlamOptimizer()
    K = np.array([[g.fx, 0, g.cx], [0, g.fx, g.cy], [0, 0, 1]], np.float32)
    op.addCamPara(opt := o, K, 0)
    op.setExtParameter(opt, g.Rbc, g.tbc)
    pid = lambda i: -5 - i if i % 2 else (1 << 27) + i
    lid = lambda l: (1 << 26) + 3 * l
    for i in range(g.P):
        op.addVertexSE2(opt, g.poses[i], pid(i), bool(g.fixed[i]))
    for k in range(g.O):
        op.addEdgeSE2(opt, g.o_meas[k], pid(int(g.o_i[k])), pid(int(g.o_j[k])), g.o_info[k])
    for l in range(g.L):
        op.addVertexSBAXYZ(opt, g.lms[l], lid(l))
    for k in range(g.E):
        w = g.e_info[k]
        op.addEdgeSE2XYZ(opt, g.e_uv[k], pid(int(g.e_kf[k])), lid(int(g.e_lm[k])), [[w[0], w[1]], [w[1], w[2]]], g.huber)
    opt.initializeOptimization(0)
    opt.optimize(4)
    assert opt.stats["chi2_hist"] == pooled[1][4]
    assert np.array_equal(op.estimateVertexSE2(opt, pid(3)), pooled[1][1][3])


def test_optimize_batch_equals_one_by_one(synth):
    """se2gpu_ba_optimize_batch: several independent windows (different sizes, one of them a start that rejects trials)
    driven at once, every controller on the device - bit-identical to optimising them one after the other."""
    from se2lam_amd.optimizer import optimize_batch
    graphs = [synth.ba_graph(50, 5000), synth.ba_graph(8, 60), _kidnapped(synth, *LM_REJECT_CASES[3][0]),
              synth.ba_graph(21, 800), synth.ba_graph(50, 5000)]
    ref = []
    for g in graphs:
        o = _opt(g)
        o.optimize(10)
        ref.append((o.stats, o.estimates()))
    opts = [_opt(g) for g in graphs]
    its = optimize_batch(opts, 10)
    for o, (st, (p, l)), n in zip(opts, ref, its):
        assert n == st["iterations"]
        assert o.stats == st
        pp, ll = o.estimates()
        assert np.array_equal(pp, p) and np.array_equal(ll, l)
    assert ref[2][0]["trials_hist"] == LM_REJECT_CASES[3][1]


def _opt_initial(o):
    """the estimates the optimizer was initialised with (reset + download on its own stream)"""
    o.reset_estimates()
    return o.estimates()


def test_lockstep_batch_mixed_rejections_repeats_and_gauss_newton(synth):
    """the windows of a batch share ONE launch per stage (k_batched: blockIdx.y = window) and still decide
    for themselves - five starts that reject trials at different iterations next to windows that never do, run twice (the
    second call re-uses the argument packs on the device) and once in Gauss-Newton mode: every window bit-identical to its
    own one-by-one run."""
    from se2lam_amd.optimizer import optimize_batch
    graphs = [_kidnapped(synth, *c[0]) for c in LM_REJECT_CASES] + [synth.ba_graph(8, 60), synth.ba_graph(30, 2000), synth.ba_graph(50, 5000)]
    for mode in (0, 1):
        ref = []
        for g in graphs:
            o = _opt(g)
            o.optimize(10, mode) if mode else o.optimize(10)
            ref.append((o.stats, o.estimates()))
        if mode == 0:
            assert [r[0]["trials_hist"] for r in ref[:len(LM_REJECT_CASES)]] == [c[1] for c in LM_REJECT_CASES]
        opts = [_opt(g) for g in graphs]
        for rep in range(2):
            for o in opts:
                o.reset_estimates()
            its = optimize_batch(opts, 10, mode)
            for o, (st, (p, l)), n in zip(opts, ref, its):
                assert n == st["iterations"] and o.stats == st
                pp, ll = o.estimates()
                assert np.array_equal(pp, p) and np.array_equal(ll, l)
    # se2gpu_ba_reset_estimates_batch: one launch for all windows, ordered before whatever follows on any of them - the next
    # batch (same stream), or a single window optimised on its own stream
    from se2lam_amd.optimizer import reset_estimates_batch
    reset_estimates_batch(opts)
    optimize_batch(opts, 10, 1)
    assert [o.stats for o in opts] == [r[0] for r in ref]       # (ref holds the Gauss-Newton runs at this point)
    reset_estimates_batch(opts)
    o3 = opts[3]
    o3.optimize(10, 1)
    assert o3.stats == ref[3][0] and np.array_equal(o3.estimates()[0], ref[3][1][0])
    reset_estimates_batch(opts)
    for o, (st, (p, l)) in zip(opts, ref):
        e0 = o.estimates()                      # (joins the batch stream before the download)
        assert np.array_equal(e0[0], _opt_initial(o)[0])
    # a shorter run on the same handles (another plan), and a stop flag raised from the start
    ref4 = []
    for g in graphs:
        o = _opt(g)
        o.optimize(4)
        ref4.append(o.stats)
    for o in opts:
        o.reset_estimates()
    optimize_batch(opts, 4)
    assert [o.stats for o in opts] == ref4
    stop = np.ones(1, np.uint8)
    for o in opts:
        o.reset_estimates()
    its = optimize_batch(opts, 10, 0, stop)
    assert its == [0] * len(opts) and all(o.stats["stopped"] for o in opts)


def test_mixed_window_batch_equals_the_oracle_window_by_window(oracle, synth):
    """a batch of DISTINCT windows (sizes, seeds, solve plans all different; some starts that reject
    trials) against the ORACLE window by window - not only against one-by-one GPU runs: trial counts and lambda / chi^2
    histories of every window, and its poses."""
    from se2lam_amd.optimizer import optimize_batch
    gs = synth.mixed_windows(14, p_range=(10, 26), l_range=(200, 1100), kidnapped_every=4, seed=77)
    assert len({(g.P, g.L) for g in gs}) == len(gs)
    gs = gs[:5] + [_kidnapped(synth, *c[0]) for c in LM_REJECT_CASES[:3]] + gs[5:]   # starts whose rejections have wide margins
    opts = [_opt(g) for g in gs]
    optimize_batch(opts, 8)
    rejecting = 0
    for g, o in zip(gs, opts):
        poses, lms, st = oracle.ba_optimize(g, 8, 0)
        assert o.stats["trials_hist"] == st["trials_hist"], (g.P, g.L, o.stats["trials_hist"], st["trials_hist"])
        assert np.allclose(o.stats["chi2_hist"], st["chi2_hist"], rtol=1e-6) and np.allclose(o.stats["lambda_hist"], st["lambda_hist"], rtol=1e-6)
        assert np.allclose(o.estimates()[0], poses, rtol=1e-6, atol=1e-6)
        rejecting += max(st["trials_hist"]) > 1
    assert rejecting >= 2


def test_batch_plans_outlive_their_windows(synth):
    """the lock-step driver keeps the argument packs of its last four batches per thread; a plan is evicted long after the
    windows it was built for - and their streams - have been destroyed (more windows than the handle pool parks).  Seven
    batches of six fresh windows each, with different shapes, run through that eviction; every batch still equals the
    one-by-one runs.  (The eviction used to synchronise the evicted plan's stream - the stream of a destroyed window.)"""
    import gc
    from se2lam_amd.optimizer import optimize_batch
    for round_, iters in enumerate((3, 5, 2, 4, 6, 3, 2)):
        graphs = [synth.ba_graph(9 + round_ + k, 70 + 10 * k, seed=50 + 7 * round_ + k) for k in range(6)]
        ref = []
        for g in graphs:
            o = _opt(g)
            o.optimize(iters)
            ref.append(o.stats)
        opts = [_opt(g) for g in graphs]
        optimize_batch(opts, iters)
        assert [o.stats for o in opts] == ref
        del opts, o
        gc.collect()           # six destroys: four handles are parked, two (the batch's first windows) really freed


def test_force_stop_flag_and_synchronous_controller(synth):
    """setForceStopFlag (LocalMapper.cpp:246): a flag that is already set leaves the estimate untouched and reports
    `stopped`; the synchronous controller (SE2GPU_BA_SYNC=1 semantics are the same code path as verbose) agrees with the
    asynchronous one."""
    g = synth.ba_graph(21, 800)
    o = _opt(g)
    flag = np.ones(1, np.uint8)
    o.setForceStopFlag(flag)
    assert o.optimize(10) == 0 and o.stats["stopped"]
    assert o.stats["chi2_init"] == pytest.approx(o.activeRobustChi2(), rel=1e-14)
    assert np.array_equal(o.estimates()[0], g.poses)
    flag[0] = 0
    assert o.optimize(4) == 4 and not o.stats["stopped"]
    a = _opt(g)
    a.optimize(4)
    assert a.stats["chi2_hist"] == o.stats["chi2_hist"]
    v = _opt(g)
    v.setVerbose(True)              # verbose = synchronous controller: one host round trip per trial
    v.optimize(4)
    assert v.stats == a.stats and np.array_equal(v.estimates()[0], a.estimates()[0])
    z = _opt(g)
    assert z.optimize(0) == 0 and z.stats["chi2_init"] == pytest.approx(a.stats["chi2_init"], rel=1e-14)


def _run_plan(g, plan, monkeypatch, iters=6):
    if plan:
        monkeypatch.setenv("SE2GPU_BA_PLAN", plan)
    else:
        monkeypatch.delenv("SE2GPU_BA_PLAN", raising=False)
    o = _opt(g)
    S, bs = o.reduced_system(3.5)
    o.optimize(iters)
    return S, bs, o.stats, o.estimates()


def test_device_plan_equals_host_plan(synth, monkeypatch):
    """initializeOptimization builds the landmark / pose CSR lists and k_reduce2's contributor plan ON THE DEVICE
    (k_plan_*, stable radix sorts, the packing scan); SE2GPU_BA_PLAN=host keeps the host builder.  Same lists, same
    summation orders: the reduced system and the whole LM run are bit-identical - on plain windows, with several fixed
    key frames, with an unobserved and a once-observed landmark, for a single key frame, and at the bench size."""
    import dataclasses
    g8 = synth.ba_graph(8, 60)
    fx = g8.fixed.copy(); fx[[0, 3, 7]] = 1
    keep = (g8.e_lm != 5) & ~((g8.e_lm == 6) & (np.cumsum(g8.e_lm == 6) > 1))
    one = dataclasses.replace(g8, poses=g8.poses[:1], fixed=np.zeros(1, np.uint8), e_kf=g8.e_kf[g8.e_kf == 0] * 0,
                              e_lm=g8.e_lm[g8.e_kf == 0], e_uv=g8.e_uv[g8.e_kf == 0], e_info=g8.e_info[g8.e_kf == 0],
                              o_i=g8.o_i[:0], o_j=g8.o_j[:0], o_meas=g8.o_meas[:0], o_info=g8.o_info[:0])
    graphs = [g8, dataclasses.replace(g8, fixed=fx),
              dataclasses.replace(g8, e_kf=g8.e_kf[keep], e_lm=g8.e_lm[keep], e_uv=g8.e_uv[keep], e_info=g8.e_info[keep]),
              one, synth.ba_graph(21, 800), synth.ba_graph(50, 5000), synth.ba_graph(200, 20000)]
    for g in graphs:
        Sd, bd, sd, (pd_, ld_) = _run_plan(g, None, monkeypatch)
        Sh, bh, sh, (ph, lh) = _run_plan(g, "host", monkeypatch)
        assert np.array_equal(Sd, Sh) and np.array_equal(bd, bh), g.P
        assert sd == sh
        assert np.array_equal(pd_, ph) and np.array_equal(ld_, lh)


def test_edges_in_any_order(synth, oracle):
    """Edges that do not arrive grouped by landmark (the reference adds them map point by map point, but the call surface
    does not require it) are stably sorted first; only the summation order inside a landmark changes."""
    import dataclasses
    g = synth.ba_graph(21, 800)
    perm = np.random.default_rng(3).permutation(g.E)
    gs = dataclasses.replace(g, e_kf=g.e_kf[perm], e_lm=g.e_lm[perm], e_uv=g.e_uv[perm], e_info=g.e_info[perm])
    a, b = _opt(g), _opt(gs)
    a.optimize(6); b.optimize(6)
    assert a.stats["trials_hist"] == b.stats["trials_hist"]
    assert np.allclose(a.stats["chi2_hist"], b.stats["chi2_hist"], rtol=1e-10)
    assert np.allclose(a.estimates()[0], b.estimates()[0], rtol=1e-9, atol=1e-9)


def test_load_local_graph_pod_call(synth, oracle):
    """Map::loadLocalGraph through ONE POD call (se2gpu_ba_load_local_graph): vertex ids, the fixed rule, cov^-1 of the
    PreSE2 edges and the per-observation information are the library's business.  Checked against the reference's call
    sequence spelled out with the free functions (Map.cpp:891-1053) and the oracle's information matrices - with and
    without reference key frames, with an observation by a key frame outside both lists."""
    from se2lam_amd import optimizer as op
    from test_ba_oracle import _info_inputs
    inp, g, level = _info_inputs(synth, 12, 200)
    K = np.array([[g.fx, 0, g.cx], [0, g.fx, g.cy], [0, 0, 1]], np.float32)
    info = oracle.ba_edge_information(**inp)
    for n_ref in (0, 3):
        nL = g.P - n_ref
        kf_id = np.arange(100, 100 + g.P, dtype=np.int32)      # KeyFrame::id; the smallest one is local key frame 4
        kf_id[4] = 50
        if n_ref == 0:
            fixed = (kf_id == kf_id.min())
        else:
            fixed = np.r_[np.zeros(nL, bool), np.ones(n_ref, bool)]
        odo_to = np.full(nL, -1, np.int32); odo_meas = np.zeros((nL, 3)); odo_cov = np.tile(np.eye(3).reshape(-1), (nL, 1))
        for k in range(g.O):
            i, j = int(g.o_i[k]), int(g.o_j[k])
            if i < nL and j < nL:
                odo_to[i] = j; odo_meas[i] = g.o_meas[k]; odo_cov[i] = np.linalg.inv(g.o_info[k].reshape(3, 3)).reshape(-1)
        obs_kf = g.e_kf.astype(np.int32).copy()
        dropped = np.zeros(g.E, bool); dropped[::37] = True     # observed by a key frame that is in neither list
        obs_kf[dropped] = -1
        # the reference's call sequence
        ref = op.SlamOptimizer()
        op.addCamPara(ref, K, 0)
        op.setExtParameter(ref, g.Rbc, g.tbc)
        twb = np.c_[inp["twb_xy"], g.poses[:, 2].astype(np.float32)].astype(np.float32)
        for i in range(nL):
            op.addVertexSE2(ref, twb[i].astype(np.float64), i, bool(fixed[i]))
        for i in range(nL):
            if odo_to[i] >= 0:
                op.addEdgeSE2(ref, odo_meas[i], i, int(odo_to[i]), np.linalg.inv(odo_cov[i].reshape(3, 3)))
        for i in range(n_ref):
            op.addVertexSE2(ref, twb[nL + i].astype(np.float64), nL + i, True)
        maxKFid = g.P + 1
        for l in range(g.L):
            op.addVertexSBAXYZ(ref, g.lms[l].astype(np.float32).astype(np.float64), maxKFid + l)
        uv32 = g.e_uv.astype(np.float32)
        for k in range(g.E):
            if not dropped[k]:
                op.addEdgeSE2XYZ(ref, uv32[k].astype(np.float64), int(g.e_kf[k]), maxKFid + int(g.e_lm[k]), info[k],
                                 float(np.float32(g.huber)))   # const float delta = Config::TH_HUBER (Map.cpp:977)
        ref.initializeOptimization(0)
        ref.optimize(5)
        # one POD call
        pod = op.SlamOptimizer()
        op.loadLocalGraph(pod, kf_id=kf_id, kf_Twb=twb, kf_Rcw=inp["Rcw"], n_local=nL, odo_to=odo_to, odo_meas=odo_meas,
                          odo_cov=odo_cov, mp_pos=g.lms.astype(np.float32), obs_mp=g.e_lm, obs_kf=obs_kf, obs_uv=uv32,
                          obs_lc=inp["lc"], obs_sigma2=inp["sigma2"], K=K, Rbc=g.Rbc, tbc=g.tbc, huber=np.float32(g.huber))
        pod.initializeOptimization(0)
        pod.optimize(5)
        assert pod.stats["trials_hist"] == ref.stats["trials_hist"]
        assert np.allclose(pod.stats["chi2_hist"], ref.stats["chi2_hist"], rtol=1e-9)
        for i in range(g.P):
            assert np.allclose(op.estimateVertexSE2(pod, i), op.estimateVertexSE2(ref, i), rtol=1e-9, atol=1e-9)
        assert np.allclose(op.estimateVertexSBAXYZ(pod, maxKFid + 7), op.estimateVertexSBAXYZ(ref, maxKFid + 7), rtol=1e-9)
        # fixed vertices did not move
        for i in np.nonzero(fixed)[0]:
            assert np.array_equal(op.estimateVertexSE2(pod, int(i)), twb[i].astype(np.float64))


def test_dataflow_timeout_falls_back_to_the_column_launches(synth):
    """k_chol_tiles makes progress only while workgroups are dispatched in task order.  Should that ever
    fail, a dependency spin times out after 2 s - and the run must go on with k_chol_step instead of ending localBA with
    an error.  SE2GPU_BA_CHOL_FAULT=1 (read once per process, hence the child) launches the first dataflow solve without
    its first task; the child's histories must equal an undisturbed run's."""
    import json
    import subprocess
    import sys
    code = ("import json, sys; sys.path.insert(0, %r)\n"
            "from se2lam_amd import synth\n"
            "from se2lam_amd.optimizer import SlamOptimizer\n"
            "g = synth.ba_graph(20, 600)\n"
            "o = SlamOptimizer(); o.load(g); o.initializeOptimization(0); o.optimize(6)\n"
            "print(json.dumps({'chi2': o.stats['chi2_hist'], 'trials': o.stats['trials_hist'], 'path': o.solver_path()}))\n") % ROOT
    env = dict(os.environ)
    env["SE2GPU_BA_CHOL_FAULT"] = "1"
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, timeout=120)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "k_chol_tiles timed out; continuing with k_chol_step" in r.stderr
    got = json.loads(r.stdout.strip().splitlines()[-1])
    assert got["path"] == 2                      # se2gpu_ba_debug_solver_path: column launches after a time-out
    ref = _opt(synth.ba_graph(20, 600))
    ref.optimize(6)
    assert ref.solver_path() == 0                # (everywhere else the harness raises if a handle fell back)
    assert got["trials"] == ref.stats["trials_hist"]
    assert np.allclose(got["chi2"], ref.stats["chi2_hist"], rtol=1e-9)


def test_repeated_optimize_replays_a_graph_with_identical_results(synth):
    """the same optimize(n) asked again of an initialised handle is captured as a hipGraph the second time and replayed
    from the third on (ba_run_begin): the first (direct launches), second (capture + launch) and later (replay) runs
    must give bit-identical histories and estimates - also on a start that rejects trials (the extra slots of a
    rejected trial are enqueued directly, after the graph) and for the SE3 / pose-graph models"""
    from se2lam_amd import optimizer as op
    case, trials = LM_REJECT_CASES[1]
    for g, loader in ((synth.ba_graph(30, 900), None), (_kidnapped(synth, *case), None),
                      (synth.ba3_graph(12, 300, 2), op.load_se3_graph), (synth.pose_graph(40), op.load_pose_graph)):
        o = op.SlamOptimizer()
        if loader:
            loader(o, g)
        else:
            o.load(g)
        o.initializeOptimization(0)
        runs = []
        for rep in range(5):
            o.reset_estimates()
            o.optimize(7 if rep < 4 else 3)              # the last one: another shape, direct again
            runs.append((o.stats["trials_hist"], o.stats["chi2_hist"], o.stats["lambda_hist"]))
        assert runs[0] == runs[1] == runs[2] == runs[3]
        assert runs[4][0] == runs[0][0][:3] and runs[4][1] == runs[0][1][:3]
