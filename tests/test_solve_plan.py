"""The plan of the dense pose solve (csrc/ba.hip: solve_plan_choose / solve_plan_build) - host code, no device needed.
What CHOLMOD's symbolic analysis is to the reference (include/se2lam/optimizer.h:31: LinearSolverCholmod), this is to the
dataflow solve: a fill-reducing order of the poses (nested dissection of a band or ring of key frames), the non-zero tiles
of L and of R = L^-T by a symbolic factorisation, and for every tile task the list of block columns it waits for.
The test runs the tile algorithm of k_chol_tiles in numpy FROM THOSE LISTS (a tile without a task is never computed, an
update without a list entry never applied) and compares x with a dense solve: a missing tile or dependency shows up as a
wrong solution, a superfluous one only costs time."""
import ctypes as C

import numpy as np
import pytest

NB = 32      # module default; the tests below also run at 64 (the wide block column) through se2gpu_ba_debug_solve_plan_tile


def _plan(P, D, pattern, allow_nd=True, tile=32):
    from se2lam_amd import capi
    lib = capi.lib()
    out = [C.c_int() for _ in range(5)]
    pat = None if pattern is None else np.ascontiguousarray(pattern, np.uint8)
    pp = None if pat is None else pat.ctypes.data
    capi.check(lib.se2gpu_ba_debug_solve_plan_tile(P, D, pp, int(allow_nd), tile, *[C.byref(o) for o in out], None, None, 0, None, 0))
    nsys, nbc, depth, ntask, ndep = [o.value for o in out]
    off = np.zeros(P, np.int32)
    tasks = np.zeros((ntask, 4), np.int32)
    deps = np.zeros(max(ndep, 1), np.int32)
    capi.check(lib.se2gpu_ba_debug_solve_plan_tile(P, D, pp, int(allow_nd), tile, *[C.byref(o) for o in out], off.ctypes.data,
                                              tasks.ctypes.data, ntask, deps.ctypes.data, len(deps)))
    return dict(nsys=nsys, nbc=nbc, depth=depth, off=off, tasks=tasks, deps=deps[:ndep])


def _random_spd(rng, P, D, pattern):
    n = D * P
    S = np.zeros((n, n))
    for a in range(P):
        for b in range(a):
            if pattern is None or pattern[a, b] or pattern[b, a]:
                blk = rng.normal(size=(D, D))
                S[D * a:D * a + D, D * b:D * b + D] = blk
                S[D * b:D * b + D, D * a:D * a + D] = blk.T
    S += np.diag(np.abs(S).sum(1) + rng.uniform(1, 2, n))      # strictly diagonally dominant: SPD
    return S


def _solve_by_tasks(plan, P, D, S, b):
    """the dataflow of k_chol_tiles in exact arithmetic order-of-magnitude (LL^T per tile instead of the stacked LDL^T)"""
    nsys, nbc, off = plan["nsys"], plan["nbc"], plan["off"]
    ld = -(-(nsys + 1) // NB) * NB
    A = np.zeros((ld, ld))
    cols = np.concatenate([off[p] + np.arange(D) for p in range(P)])
    A[np.ix_(cols, cols)] = S
    pad = np.setdiff1d(np.arange(nsys), cols)
    A[pad, pad] = 1.0
    A[nsys, cols] = b
    it = nsys // NB
    Lt, Rt, yt, x = {}, {}, {}, np.zeros(ld)
    tile = lambda M, i, j: M[NB * i:NB * i + NB, NB * j:NB * j + NB]
    for ti, tj, d0, d1 in plan["tasks"]:
        kind, i = ti >> 16, ti & 0xffff
        dl = plan["deps"][d0:d1]
        if kind == 2:                                   # x(r) = sum_j R(r, j) y(j)
            r = i
            acc = np.zeros(NB)
            for j in dl:
                acc += Rt[(r, int(j))] @ yt[int(j)]
            x[NB * r:NB * r + NB] = acc
            continue
        j = tj
        w = min(NB, nsys - NB * j)                      # columns of the last tile of an unpadded system
        Dj = tile(A, j, j).copy()
        T = tile(A, i, j).copy() if kind == 0 else np.zeros((NB, NB))
        for dep in dl:
            m, has = int(dep) & 0x7fff, int(dep) >> 15
            assert (j, m) in Lt, f"task ({i},{j}) lists column {m} but L({j},{m}) has no task"
            Ljm = Lt[(j, m)]
            Dj -= Ljm @ Ljm.T
            if has:
                src = Lt[(i, m)] if kind == 0 else Rt[(i, m)]
                T -= src @ Ljm.T
        Dw = Dj[:w, :w]
        Lw = np.linalg.cholesky(Dw)
        if kind == 0 and i == j:
            full = np.zeros((NB, NB)); full[:w, :w] = Lw
            Lt[(j, j)] = full
            Rjj = np.zeros((NB, NB)); Rjj[:w, :w] = np.linalg.inv(Lw).T
            Rt[(j, j)] = Rjj
            if it == j:                                 # the rhs row lives inside the last diagonal tile
                yrow = Dj[nsys - NB * j, :w]
                yj = np.zeros(NB); yj[:w] = np.linalg.solve(Lw, yrow)
                yt[j] = yj
        else:
            out = np.zeros((NB, NB))
            out[:, :w] = np.linalg.solve(Lw, T[:, :w].T).T
            (Lt if kind == 0 else Rt)[(i, j)] = out
            if kind == 0 and i == it:
                yt[j] = out[nsys - NB * it].copy()
    xs = np.zeros(D * P)
    for p in range(P):
        xs[D * p:D * p + D] = x[off[p]:off[p] + D]
    return xs


def _band(P, w, ring):
    d = np.abs(np.subtract.outer(np.arange(P), np.arange(P)))
    if ring:
        d = np.minimum(d, P - d)
    return (d <= w).astype(np.uint8)


@pytest.mark.parametrize("name,P,pattern", [
    ("dense 40", 40, None),
    ("dense 11 (one tile + rhs inside it)", 10, None),
    ("ring 200 / 41", 200, _band(200, 41, True)),
    ("ring 200 / 43 (widened separators)", 200, _band(200, 43, True)),
    ("ring 50 / 10", 50, _band(50, 10, True)),
    ("open band 120 / 12", 120, _band(120, 12, False)),
    ("open band 64 / 30 (too wide to cut)", 64, _band(64, 30, False)),
])
def test_tile_tasks_solve_the_system(name, P, pattern):
    rng = np.random.default_rng(P)
    D = 3
    if pattern is not None:                            # a few poses that share nothing with anybody (fixed key frames)
        pattern = pattern.copy()
        for p in (3, P // 2):
            pattern[p, :] = 0; pattern[:, p] = 0; pattern[p, p] = 1
    S = _random_spd(rng, P, D, pattern)
    b = rng.normal(size=D * P)
    want = np.linalg.solve(S, b)
    for nd in (False, True):
        plan = _plan(P, D, pattern, nd)
        got = _solve_by_tasks(plan, P, D, S, b)
        assert np.allclose(got, want, rtol=1e-9, atol=1e-11), (name, nd, np.abs(got - want).max())
        if not nd:
            assert plan["nsys"] == D * P and np.array_equal(plan["off"], D * np.arange(P))     # natural order, no padding


def test_nested_dissection_shortens_the_chain_of_block_columns():
    """the point of the order: fewer block columns on the longest dependency chain (7 us each on the device)"""
    nat, nd = _plan(200, 3, _band(200, 41, True), False), _plan(200, 3, _band(200, 41, True), True)
    assert nat["depth"] == nat["nbc"] == 19 and nd["depth"] <= 15 and nd["nsys"] % NB == 0
    # the bench loop (band 43): separators widened from 43 to 48 key frames so that the arcs end on a tile boundary -
    # 5 + 9 block columns on the chain instead of 6 + 9
    nd = _plan(200, 3, _band(200, 43, True), True)
    assert nd["depth"] <= 14 and nd["nsys"] == 608
    nat, nd = _plan(50, 3, _band(50, 10, True), False), _plan(50, 3, _band(50, 10, True), True)
    assert nat["depth"] == 5 and nd["depth"] <= 4
    nat, nd = _plan(120, 3, _band(120, 12, False), False), _plan(120, 3, _band(120, 12, False), True)
    assert nat["depth"] == 12 and nd["depth"] <= 8
    # nothing to gain: the order stays natural (no padding, bit-identical to the solve without the analysis)
    for P, pat in ((40, None), (64, _band(64, 30, False)), (30, _band(30, 3, False))):
        p = _plan(P, 3, pat, True)
        assert p["nsys"] == 3 * P and p["depth"] == p["nbc"]
    # the bench graphs themselves (synth.ba_graph: a loop of key frames)
    from se2lam_amd import synth
    for P, L, dmax in ((200, 20000, 15), (50, 5000, 4)):
        g = synth.ba_graph(P, L)
        pat = np.eye(P, dtype=np.uint8)
        order = np.argsort(g.e_lm, kind="stable")
        lm, kf = g.e_lm[order], g.e_kf[order]
        ptr = np.searchsorted(lm, np.arange(g.L + 1))
        for l in range(g.L):
            k = kf[ptr[l]:ptr[l + 1]]
            pat[np.ix_(k, k)] = 1
        pat[g.o_i, g.o_j] = 1; pat[g.o_j, g.o_i] = 1
        fx = np.asarray(g.fixed, bool)
        pat[fx, :] = 0; pat[:, fx] = 0
        np.fill_diagonal(pat, 1)
        assert _plan(P, 3, pat, True)["depth"] <= dmax < _plan(P, 3, pat, False)["depth"]


def test_one_sided_pattern_gives_the_plan_of_its_symmetric_closure():
    """The chooser works on pattern | pattern' (initialize hands over a symmetric one; the debug entry point need not)."""
    rng = np.random.default_rng(3)
    P = 90
    full = np.zeros((P, P), np.uint8)
    for a in range(P):
        for d in range(1, 14):
            full[a, (a + d) % P] = full[(a + d) % P, a] = 1
    upper = np.triu(full)
    lower = np.tril(full)
    mixed = np.where(rng.random((P, P)) < 0.5, upper, lower).astype(np.uint8)
    mixed = np.maximum(mixed, np.where((mixed + mixed.T) == 0, full, 0)).astype(np.uint8)   # every edge on at least one side
    want = _plan(P, 3, full)
    for pat in (upper, lower, mixed):
        got = _plan(P, 3, pat)
        assert got["nsys"] == want["nsys"] and got["depth"] == want["depth"]
        assert np.array_equal(got["tasks"], want["tasks"]) and np.array_equal(got["deps"], want["deps"])


@pytest.mark.parametrize("name,P,pattern", [
    ("dense 40", 40, None),
    ("dense 21 (one tile + rhs inside it)", 21, None),
    ("ring 200 / 41", 200, _band(200, 41, True)),
    ("ring 200 / 43", 200, _band(200, 43, True)),
    ("open band 120 / 12", 120, _band(120, 12, False)),
    ("open band 64 / 30 (too wide to cut)", 64, _band(64, 30, False)),
])
def test_tile_tasks_solve_the_system_at_tile_64(monkeypatch, name, P, pattern):
    """the plan for the wide block column (docs/history/DESIGN_rounds_1-5.md 8.1): the same numpy executor, 64 x 64 tiles"""
    import sys
    monkeypatch.setattr(sys.modules[__name__], "NB", 64)
    rng = np.random.default_rng(P)
    D = 3
    if pattern is not None:
        pattern = pattern.copy()
        for p in (3, P // 2):
            pattern[p, :] = 0; pattern[:, p] = 0; pattern[p, p] = 1
    S = _random_spd(rng, P, D, pattern)
    b = rng.normal(size=D * P)
    want = np.linalg.solve(S, b)
    for nd in (False, True):
        plan = _plan(P, D, pattern, nd, tile=64)
        got = _solve_by_tasks(plan, P, D, S, b)
        assert np.allclose(got, want, rtol=1e-9, atol=1e-11), (name, nd, np.abs(got - want).max())
        assert plan["nsys"] % 64 == 0 or plan["nsys"] == D * P


def test_nested_dissection_at_tile_64_puts_eight_block_columns_on_the_chain():
    nat, nd = _plan(200, 3, _band(200, 43, True), False, tile=64), _plan(200, 3, _band(200, 43, True), True, tile=64)
    assert nat["depth"] == nat["nbc"] == 10 and nd["depth"] <= 8 and nd["nsys"] % 64 == 0
