"""The way the reference runs (src/OdoSLAM.cpp:142-145): the Track thread (ORBextractor, ORBmatcher, removeOutliers,
doTriangulate - one frame at a time) and the LocalMapper thread (a fresh SlamOptimizer per localBA call, whose dense
solve is the spin-waiting dataflow kernel k_chol_tiles) share ONE GPU, each with its own handles and streams.
200 frames against 20 local bundle adjustments, concurrently: every result must equal the serial run's, and no
dataflow dependency may time out."""
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

N_FRAMES, N_BA = 200, 20


def _track_loop(synth, out):
    from se2lam_amd import orb
    from se2lam_amd.matcher import ORBmatcher
    from se2lam_amd.track import Track
    ex, tr = orb.ORBextractor(), Track()
    prev = None
    for t in range(N_FRAMES):
        k, d = ex(synth.frame(t))
        rec = [k.copy(), d.copy()]
        if prev is not None:
            mt = ORBmatcher(0.9)                      # Track.cpp:131: constructed on the stack of every frame
            pxy = np.ascontiguousarray(np.stack([prev[0]["x"], prev[0]["y"]], 1), np.float32)
            nm, m12 = mt.MatchByWindow(prev[0], prev[1], k, d, pxy, 20)
            m = np.ascontiguousarray(m12, np.int32).copy()
            ninl = tr.removeOutliers(prev[0], k, m)
            rec += [nm, np.asarray(m12).copy(), ninl, m]
        out.append(rec)
        prev = (k, d)


def _ba_loop(synth, out):
    from se2lam_amd.optimizer import SlamOptimizer
    graphs = [synth.ba_graph(50, 5000), synth.ba_graph(21, 800), synth.ba_graph(200, 20000)]
    for i in range(N_BA):
        g = graphs[i % 3] if i % 7 else graphs[2]
        o = SlamOptimizer()                           # LocalMapper.cpp:239: a new optimizer per localBA
        o.load(g)
        o.initializeOptimization(0)
        o.optimize(10)
        p, l = o.estimates()
        out.append((o.stats, p.copy(), l.copy()))
        del o


def test_track_and_local_mapper_threads_share_the_gpu(synth):
    ser_t, ser_b = [], []
    _track_loop(synth, ser_t)
    _ba_loop(synth, ser_b)
    con_t, con_b, errors = [], [], []

    def guard(fn, out):
        try:
            fn(synth, out)
        except Exception as exc:   # pragma: no cover
            errors.append(exc)

    th = [threading.Thread(target=guard, args=(_track_loop, con_t)), threading.Thread(target=guard, args=(_ba_loop, con_b))]
    [t.start() for t in th]
    [t.join() for t in th]
    assert not errors, errors
    assert len(con_t) == N_FRAMES and len(con_b) == N_BA
    for a, b in zip(ser_t, con_t):
        assert len(a) == len(b)
        for x, y in zip(a, b):
            assert np.array_equal(x, y)
    for (sa, pa, la), (sb, pb, lb) in zip(ser_b, con_b):
        assert sa == sb
        assert np.array_equal(pa, pb) and np.array_equal(la, lb)


def test_short_lived_threads_do_not_grow_the_lease_pools(synth):
    """the plan caches of the lock-step driver and the staging of the batched reset used to be thread_local and
    never freed - a mapper on short-lived threads left a set of device buffers, pinned memory and events behind with every
    thread that exited.  They are leased from process-wide pools now: twelve threads, one after the other, each running a
    batched reset + a window batch, leave exactly one idle set in each pool, and their results equal the main thread's."""
    import ctypes as C
    from se2lam_amd import capi
    from se2lam_amd.optimizer import SlamOptimizer, optimize_batch, reset_estimates_batch
    graphs = [synth.ba_graph(9 + k, 80 + 20 * k, seed=900 + k) for k in range(5)]

    def run(out):
        opts = []
        for g in graphs:
            o = SlamOptimizer(); o.load(g); o.initializeOptimization(0); opts.append(o)
        optimize_batch(opts, 4)
        reset_estimates_batch(opts)
        optimize_batch(opts, 4)
        out.append([(o.stats, o.estimates()[0].copy()) for o in opts])

    ref = []
    run(ref)
    sizes = (C.c_int * 2)()
    capi.check(capi.lib().se2gpu_ba_debug_pool_sizes(sizes))
    base = (sizes[0], sizes[1])
    assert base[0] >= 1 and base[1] >= 1
    for _ in range(12):
        got, errors = [], []

        def guard():
            try:
                run(got)
            except Exception as exc:   # pragma: no cover
                errors.append(exc)
        t = threading.Thread(target=guard)
        t.start(); t.join()
        assert not errors, errors
        for (sa, pa), (sb, pb) in zip(ref[0], got[0]):
            assert sa == sb and np.array_equal(pa, pb)
    capi.check(capi.lib().se2gpu_ba_debug_pool_sizes(sizes))
    assert (sizes[0], sizes[1]) == base, ((sizes[0], sizes[1]), base)
