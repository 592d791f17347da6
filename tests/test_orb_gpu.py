"""GPU parity tests of the HIP ORB extractor (through the C ABI) against the CPU oracle: BIT-EXACT
pyramid, FAST score map, blurred pyramid, key points (all 7 cv::KeyPoint fields) and descriptors."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ex():
    from se2lam_amd.orb import ORBextractor
    return ORBextractor(max_batch=8)


def test_pyramid_score_blur_bit_exact(ex, oracle, synth):
    img = synth.frame(3)
    ex(img)
    for lv in range(8):
        assert np.array_equal(ex.debug_level(0, lv), oracle.orb_level(img, lv)), f"level {lv}"
        assert np.array_equal(ex.debug_level(0, lv, blurred=True), oracle.orb_level(img, lv, blurred=True)), f"blur {lv}"
        # ... and with the 16 px reflect-101 frame (the blurred pyramid keeps the un-blurred frame)
        assert np.array_equal(ex.debug_level(0, lv, bordered=True), oracle.orb_level(img, lv, bordered=True)), f"frame {lv}"
        assert np.array_equal(ex.debug_level(0, lv, blurred=True, bordered=True),
                              oracle.orb_level(img, lv, blurred=True, bordered=True)), f"blur frame {lv}"
        # the device score plane holds S' = S where S > 7 and S is a strict maximum among the 8-neighbours of the
        # SAME cell (cv::FAST's in-call non-max suppression), else 0
        s_gpu, s_ref = ex.debug_score(0, lv), oracle.orb_score(img, lv).astype(np.int32)
        w, h, gcols, grows, cellW, cellH, _ = oracle.orb_geometry(480, 640)[lv]
        exp = np.zeros_like(s_ref)
        ys, xs = np.nonzero(s_ref > 7)
        for y, x in zip(ys, xs):
            if not (16 <= x < w - 16 and 16 <= y < h - 16):
                continue
            cj, ci = min((x - 16) // cellW, gcols - 1), min((y - 16) // cellH, grows - 1)
            xa, ya = 16 + cj * cellW, 16 + ci * cellH
            xb = w - 16 if cj == gcols - 1 else xa + cellW
            yb = h - 16 if ci == grows - 1 else ya + cellH
            nb = s_ref[max(y - 1, ya):min(y + 2, yb), max(x - 1, xa):min(x + 2, xb)]
            if (nb >= s_ref[y, x]).sum() == 1:
                exp[y, x] = s_ref[y, x]
        assert np.array_equal(s_gpu.astype(np.int32), exp), f"score {lv}"


@pytest.mark.parametrize("t", range(10))
def test_extract_bit_exact_config1_frames(ex, oracle, synth, t):
    """config 1/2: the ten 640x480 synthetic frames, 8 levels, 1000 features - every key point field and
    every descriptor byte equal to the oracle's."""
    img = synth.frame(t)
    k, d = ex(img)
    ko, do = oracle.orb_extract(img)
    assert len(k) == len(ko) == 1000
    for field in ("octave", "x", "y", "response", "size", "angle", "class_id"):
        assert np.array_equal(k[field], ko[field]), field
    assert np.array_equal(d, do)


def test_batch_equals_single(ex, oracle, synth):
    imgs = synth.frames(8, start=20)
    out = ex.extract_batch(imgs)
    for b in range(8):
        ko, do = oracle.orb_extract(imgs[b])
        assert np.array_equal(out[b][0], ko) and np.array_equal(out[b][1], do), b


def test_other_image_sizes_and_parameters(oracle, synth):
    from se2lam_amd.orb import ORBextractor
    tex = synth.texture()
    for (rows, cols, nf, nl, th) in ((480, 640, 500, 8, 20), (376, 1241 // 2, 1000, 6, 12), (240, 320, 300, 4, 20)):
        img = np.ascontiguousarray(tex[50:50 + rows, 70:70 + cols])
        ex = ORBextractor(nfeatures=nf, nlevels=nl, fastTh=th, max_rows=rows, max_cols=cols)
        k, d = ex(img)
        ko, do = oracle.orb_extract(img, oracle.orb_params(nf, 1.2, nl, th))
        assert len(k) == len(ko) > 0
        assert np.array_equal(k, ko) and np.array_equal(d, do), (rows, cols)


def test_edge_cases(ex, oracle, synth):
    # empty image: silent return (ORBextractor.cpp:730-731)
    k, d = ex(np.zeros((0, 0), np.uint8))
    assert len(k) == 0
    # flat image: no corners at either threshold
    k, d = ex(np.full((480, 640), 128, np.uint8))
    assert len(k) == 0
    # low contrast: only the threshold-7 fallback fires (cells with <= 3 corners at 20)
    img = (128 + (synth.frame(0).astype(np.int32) - 128) // 8).astype(np.uint8)
    k, d = ex(img)
    ko, do = oracle.orb_extract(img)
    assert len(ko) > 0 and np.array_equal(k, ko) and np.array_equal(d, do)
    # salt-and-pepper noise: thousands of candidates per cell, many ties at the retain boundary
    rng = np.random.default_rng(5)
    img = rng.integers(0, 256, (480, 640)).astype(np.uint8)
    k, d = ex(img)
    ko, do = oracle.orb_extract(img)
    assert len(ko) == 1000 and np.array_equal(k, ko) and np.array_equal(d, do)
    # a textured half and a flat half: quota redistribution between cells (ORBextractor.cpp:653-679)
    img = synth.frame(2).copy()
    img[:, 320:] = 90
    k, d = ex(img)
    ko, do = oracle.orb_extract(img)
    assert np.array_equal(k, ko) and np.array_equal(d, do)
    # strided input (cv::Mat ROI with step > cols)
    big = synth.texture()
    view = big[100:580, 200:840]
    k, d = ex(view)
    ko, do = oracle.orb_extract(np.ascontiguousarray(view))
    assert np.array_equal(k, ko) and np.array_equal(d, do)


def test_errors(ex):
    from se2lam_amd import capi
    from se2lam_amd.orb import ORBextractor
    with pytest.raises(capi.Se2GpuError) as e:
        ORBextractor(scoreType=2)                      # neither ORB::HARRIS_SCORE (0) nor ORB::FAST_SCORE (1)
    assert e.value.code == capi.ERR_INVALID
    # a mask is accepted and changes nothing: the reference builds a mask pyramid (ORBextractor.cpp:797-828) that its
    # cv::FAST calls (:616, :622) are never given
    from se2lam_amd import synth as sy
    f0 = sy.frame(0)
    half = np.zeros((480, 640), np.uint8)
    half[:, :320] = 255
    (k0, d0), (k1, d1) = ex(f0), ex(f0, mask=half)
    assert len(k0) > 0 and np.array_equal(k0, k1) and np.array_equal(d0, d1)
    k, _ = ex(np.zeros((481, 640), np.uint8))          # larger than the handle was created for: it grows, as
    assert len(k) == 0                                 # ORBextractor::operator() takes whatever image it is given


@pytest.mark.parametrize("t", [0, 4, 9])
def test_harris_score_bit_exact(oracle, synth, t):
    """scoreType = ORB::HARRIS_SCORE (ORBextractor.cpp:85-126, 625-629; dormant in the reference's own callers): the FAST
    key points of every cell are re-scored with the 7x7 Harris response and retained by it - every key point field
    (response = the float Harris value) and every descriptor byte equal to the oracle's."""
    from se2lam_amd.orb import ORBextractor, HARRIS_SCORE
    ex = ORBextractor(scoreType=HARRIS_SCORE)
    img = synth.frame(t)
    k, d = ex(img)
    kr, dr = oracle.orb_extract(img, oracle.orb_params(score_type=oracle.HARRIS_SCORE))
    assert len(k) == len(kr) > 500
    for name in k.dtype.names:
        assert np.array_equal(k[name], kr[name]), name
    assert np.array_equal(d, dr)
    assert not np.array_equal(kr["response"], np.round(kr["response"]))   # really Harris floats, not FAST scores


def test_harris_score_on_noise_and_half_flat(oracle, synth):
    """HARRIS_SCORE where cells hold thousands of candidates (bitonic 64-bit sort path, negative responses) and where
    the quota is redistributed between cells."""
    from se2lam_amd.orb import ORBextractor, HARRIS_SCORE
    ex = ORBextractor(scoreType=HARRIS_SCORE)
    par = oracle.orb_params(score_type=oracle.HARRIS_SCORE)
    rng = np.random.default_rng(11)
    imgs = [rng.integers(0, 256, (480, 640)).astype(np.uint8), synth.frame(5).copy()]
    imgs[1][:, :300] = 40
    for img in imgs:
        k, d = ex(img)
        ko, do = oracle.orb_extract(img, par)
        assert len(ko) > 0 and np.array_equal(k, ko) and np.array_equal(d, do)


def test_dense_and_sparse_score_kernels_agree(synth, monkeypatch):
    """k_fast_score (every pixel, the default) and k_fast_score_sparse (compass pre-test + candidates only,
    SE2GPU_ORB_SCORE=sparse) must produce the same score planes and the same features, also where nearly every pixel is
    a candidate (noise) and where none is (flat)."""
    from se2lam_amd.orb import ORBextractor
    monkeypatch.setenv("SE2GPU_ORB_SCORE", "dense")
    dense = ORBextractor()
    monkeypatch.setenv("SE2GPU_ORB_SCORE", "sparse")
    sparse = ORBextractor()
    monkeypatch.delenv("SE2GPU_ORB_SCORE")
    assert dense.score_kernel()[0] == "dense" and sparse.score_kernel()[0] == "sparse"
    rng = np.random.default_rng(9)
    half = synth.frame(1).copy(); half[:240] = 77
    for img in (synth.frame(4), rng.integers(0, 256, (480, 640)).astype(np.uint8), np.full((480, 640), 31, np.uint8), half):
        ks, ds = sparse(img)
        kd, dd = dense(img)
        assert np.array_equal(ks, kd) and np.array_equal(ds, dd)
        for lv in range(8):
            assert np.array_equal(sparse.debug_score(0, lv), dense.debug_score(0, lv)), lv


def test_batch_sizes_not_a_multiple_of_eight(oracle, synth):
    """frames are dealt to the 8 XCDs through the grid's fast axis (rounded up to 8): batches of 1, 3, 19 frames must
    give the single-frame results, with stale frames of a larger earlier batch left alone"""
    from se2lam_amd.orb import ORBextractor
    ex = ORBextractor(max_batch=19)
    one = ORBextractor()
    for B, start in ((19, 40), (3, 7), (1, 90)):
        imgs = synth.frames(B, start=start)
        out = ex.extract_batch(imgs)
        assert len(out) == B
        for b in range(B):
            k, d = one(imgs[b])
            assert np.array_equal(out[b][0], k) and np.array_equal(out[b][1], d), (B, b)
        ko, do = oracle.orb_extract(imgs[B - 1])
        assert np.array_equal(out[B - 1][0], ko) and np.array_equal(out[B - 1][1], do)


def test_pipelined_batches_equal_single_frames(synth, monkeypatch):
    """batches of 16 frames and more are pipelined by level (pyramid chain on its own stream, score / blur of a level as
    soon as it exists, orb_run): same features as the serial single-frame sequence, with either score kernel, twice in a
    row on the same handle (the second batch starts while the first one's key-point chain may still be running)"""
    from se2lam_amd.orb import ORBextractor
    one = ORBextractor()
    for mode in ("dense", "sparse"):
        monkeypatch.setenv("SE2GPU_ORB_SCORE", mode)
        ex = ORBextractor(max_batch=20)
        monkeypatch.delenv("SE2GPU_ORB_SCORE")
        assert ex.score_kernel()[0] == mode
        for start in (3, 50):
            imgs = synth.frames(20, start=start)
            out = ex.extract_batch(imgs)
            for b in (0, 7, 8, 19):
                k, d = one(imgs[b])
                assert np.array_equal(out[b][0], k) and np.array_equal(out[b][1], d), (mode, start, b)


def test_large_batch_pyramids_equal_the_single_frame_pyramids(oracle, synth):
    """every pyramid level (with its 16 px frame) and every blurred level of frames 0, 9 and B-1 of a batch of 43 and of 32
    frames equals the single-frame extractor's bit for bit, and so do the features - batch sizes the other tests do not
    reach (written for a fused upper-level kernel that was measured slower and dropped)"""
    from se2lam_amd.orb import ORBextractor
    one = ORBextractor()
    ex = ORBextractor(max_batch=43)
    for B, start in ((43, 11), (32, 70)):
        imgs = synth.frames(B, start=start)
        out = ex.extract_batch(imgs)
        levels = {b: [ex.debug_level(b, lv, bordered=True) for lv in range(8)] + [ex.debug_level(b, lv, blurred=True) for lv in range(8)]
                  for b in (0, 9, B - 1)}
        for b in (0, 9, B - 1):
            k, d = one(imgs[b])
            assert np.array_equal(out[b][0], k) and np.array_equal(out[b][1], d), (B, b)
            ref = [one.debug_level(0, lv, bordered=True) for lv in range(8)] + [one.debug_level(0, lv, blurred=True) for lv in range(8)]
            for q in range(16):
                assert np.array_equal(levels[b][q], ref[q]), (B, b, q)
        ko, do = oracle.orb_extract(imgs[B - 1])
        assert np.array_equal(out[B - 1][0], ko) and np.array_equal(out[B - 1][1], do)


def test_score_kernel_follows_the_candidate_density(oracle, synth):
    """default (auto) mode: the first frame goes through the candidate kernel, which counts the pixels passing its compass
    test; busy imagery (the benchmark texture: ~15 % candidates) switches to the dense kernel, quiet imagery back to the
    candidate kernel - with identical features either way"""
    from se2lam_amd.orb import ORBextractor
    ex = ORBextractor()
    busy = synth.frame(6)
    quiet = (128 + (synth.frame(6).astype(np.int32) - 128) // 8).astype(np.uint8)
    ko, do = oracle.orb_extract(busy)
    kq, dq = oracle.orb_extract(quiet)
    for _ in range(3):
        k, d = ex(busy)
        assert np.array_equal(k, ko) and np.array_equal(d, do)
    kind, dens = ex.score_kernel()
    assert kind == "dense" and 0.10 < dens < 0.30, (kind, dens)
    for _ in range(40):                       # the dense kernel does not count: the next look comes after 32 batches
        k, d = ex(quiet)
        assert np.array_equal(k, kq) and np.array_equal(d, dq)
    kind, dens = ex.score_kernel()
    assert kind == "sparse" and dens < 0.10, (kind, dens)
    for _ in range(3):                        # ... and the candidate kernel counts every time
        k, d = ex(busy)
        assert np.array_equal(k, ko) and np.array_equal(d, do)
    assert ex.score_kernel()[0] == "dense"


def test_huge_cells_full_of_corners(oracle):
    """few features on a large, busy image: one cell per level (e.g. 768 x 568 pixels at level 0) meets more than 64
    candidate lists and holds far more FAST corners than the LDS buffers (4096 to sort, 1024 to select) - the cell's list is
    then sorted into row-major order and cut by introselect in global memory instead of failing; results equal the oracle's"""
    from se2lam_amd.orb import ORBextractor
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (600, 800)).astype(np.uint8)
    smooth = (0.5 * noise + 0.5 * np.roll(noise, 1, 1)).astype(np.uint8)
    for img, nf, nl in ((noise, 150, 4), (smooth, 150, 4), (noise[:368, :], 150, 6)):
        ex = ORBextractor(nfeatures=nf, nlevels=nl, max_rows=img.shape[0], max_cols=img.shape[1])
        k, d = ex(np.ascontiguousarray(img))
        ko, do = oracle.orb_extract(np.ascontiguousarray(img), oracle.orb_params(nfeatures=nf, nlevels=nl))
        assert len(ko) > 0 and np.array_equal(k, ko) and np.array_equal(d, do)
        assert int((ex.debug_score(0, 0) > 0).sum()) > 4096        # the cell's sort buffer is far too small for all of them


def test_huge_cells_full_of_corners_with_harris_score(oracle):
    """the same images with scoreType = HARRIS_SCORE: the retention goes by the 7x7 Harris response of every FAST corner of the
    cell (64-bit list entries); same global-memory paths.  Used to be SE2GPU_ERR_CAPACITY."""
    from se2lam_amd.orb import ORBextractor
    rng = np.random.default_rng(5)
    noise = rng.integers(0, 256, (600, 800)).astype(np.uint8)
    smooth = (0.5 * noise + 0.5 * np.roll(noise, 1, 1)).astype(np.uint8)
    for img, nf, nl in ((noise, 150, 4), (smooth, 300, 3)):
        ex = ORBextractor(nfeatures=nf, nlevels=nl, scoreType=0, max_rows=img.shape[0], max_cols=img.shape[1])
        k, d = ex(np.ascontiguousarray(img))
        ko, do = oracle.orb_extract(np.ascontiguousarray(img), oracle.orb_params(nfeatures=nf, nlevels=nl, score_type=0))
        assert len(ko) > 0 and np.array_equal(k, ko) and np.array_equal(d, do)


def test_device_nth_element_equals_libstdcxx(oracle):
    """introselect_wave (csrc/orb.hip) - the routine behind both retainBest cuts - on caller data through
    se2gpu_orb_debug_nth_element, against libstdc++'s std::nth_element as restated in oracle/stl_nth.h: the whole permutation,
    LDS path and in-place global-memory path, from all keys tied to no ties, presorted / reversed inputs, every nth of small
    arrays, and median-of-three killers that reach the depth limit (the heap-select branch, run by one lane)."""
    import ctypes as C
    from se2lam_amd import capi
    f = capi.lib().se2gpu_orb_debug_nth_element

    def dev(e, nth, glob):
        a = np.ascontiguousarray(e, np.uint64).copy()
        capi.check(f(a.ctypes.data, len(a), int(nth), int(glob)))
        return a

    rng = np.random.default_rng(12)
    ncase = 0
    for n in list(range(1, 12)) + [63, 64, 65, 127, 128, 129, 200, 257, 1000, 1024, 1025, 3000, 20000]:
        for alpha in (1, 2, 3, 60, 1 << 20):
            keys = rng.integers(0, alpha, n).astype(np.uint64)
            for order in range(3):
                k = keys if order == 0 else np.sort(keys) if order == 1 else np.sort(keys)[::-1]
                e = (k << np.uint64(32)) | np.arange(n, dtype=np.uint64)
                nths = range(n) if n <= 11 else (0, 1, n // 3, n // 2, n - 2, n - 1, int(rng.integers(0, n)))
                for nth in nths:
                    want = oracle.nth_element(e, nth)
                    for glob in ((0, 1) if n <= 1024 else (1,)):
                        if n > 3000 and (order or alpha in (2, 3)):
                            continue
                        got = dev(e, nth, glob)
                        assert np.array_equal(got, want), (n, alpha, order, nth, glob)
                        ncase += 1
    h0 = oracle.nth_heap_selects()
    for n, nth in ((64, 63), (200, 100), (1000, 999), (1000, 750), (5000, 4999)):
        k = oracle.nth_killer(n, nth).astype(np.uint64)
        for fold in (1, 3):
            e = ((k // np.uint64(fold)) << np.uint64(32)) | np.arange(n, dtype=np.uint64)
            want = oracle.nth_element(e, nth)
            for glob in ((0, 1) if n <= 1024 else (1,)):
                assert np.array_equal(dev(e, nth, glob), want), ("killer", n, nth, fold, glob)
    assert oracle.nth_heap_selects() - h0 >= 4 and ncase > 1000


@pytest.mark.parametrize("rows,cols,nf,nl", [(142, 229, 1327, 3), (395, 182, 2135, 3), (506, 174, 1606, 5), (201, 423, 2170, 2),
                                             (143, 678, 878, 2), (679, 162, 400, 1), (198, 193, 2549, 2), (161, 188, 2148, 2),
                                             (137, 184, 2162, 2), (176, 671, 1623, 5)])
def test_cell_grids_whose_last_column_or_row_starts_beyond_the_scan_area(oracle, synth, rows, cols, nf, nl):
    """cellW = ceil(W / levelCols) can put the last cell column at or beyond the end of the scan area: its own window is then
    empty (no key points, ORBextractor.cpp:598-603) or not even visited (hX <= 0: `continue`, which also changes the quota
    redistribution), and the column before it scans up to 13 px into the level's reflect frame - key points with x > w - 16.
    Same for rows.  About one random (size, nfeatures, levels) combination in five is like that; rounds 1-4 refused them
    ("degenerate cell grid") although the reference runs them.  Both score kernels, FAST and Harris responses, batches."""
    from se2lam_amd.orb import ORBextractor
    tex = synth.texture()
    img = np.ascontiguousarray(tex[33:33 + rows, 61:61 + cols])
    noise = np.random.default_rng(rows * cols).integers(0, 256, (rows, cols)).astype(np.uint8)
    for st in (1, 0):
        p = oracle.orb_params(nfeatures=nf, nlevels=nl, score_type=st)
        ex = ORBextractor(nfeatures=nf, nlevels=nl, scoreType=st, max_rows=rows, max_cols=cols, max_batch=2)
        for im in (img, noise):
            ko, do = oracle.orb_extract(im, p, cap=4 * nf + 64)
            k, d = ex(im)
            assert len(ko) > 0 and np.array_equal(k, ko) and np.array_equal(d, do), (st, len(k), len(ko))
        out = ex.extract_batch(np.stack([img, noise]))
        ko, do = oracle.orb_extract(noise, p, cap=4 * nf + 64)
        assert np.array_equal(out[1][0], ko) and np.array_equal(out[1][1], do)


def test_both_score_kernels_on_a_grid_with_an_empty_last_column(oracle, synth, monkeypatch):
    rows, cols, nf, nl = 395, 182, 2135, 3
    img = np.ascontiguousarray(synth.texture()[100:100 + rows, 200:200 + cols])
    ko, do = oracle.orb_extract(img, oracle.orb_params(nfeatures=nf, nlevels=nl), cap=4 * nf + 64)
    sc = 1.2 ** ko["octave"].astype(np.float64)
    assert (ko["x"] / sc > np.round(cols / sc) - 16 + 0.5).any()               # key points inside the reflect frame
    from se2lam_amd.orb import ORBextractor
    for mode in ("dense", "sparse"):
        monkeypatch.setenv("SE2GPU_ORB_SCORE", mode)
        ex = ORBextractor(nfeatures=nf, nlevels=nl, max_rows=rows, max_cols=cols)
        k, d = ex(img)
        assert ex.score_kernel()[0] == mode and np.array_equal(k, ko) and np.array_equal(d, do), mode


@pytest.mark.parametrize("rows,cols,nf,nl", [(136, 899, 150, 7), (396, 882, 150, 7), (480, 640, 150, 8), (500, 300, 100, 6), (480, 640, 20, 8)])
def test_levels_whose_quota_is_too_small_for_one_cell(oracle, synth, rows, cols, nf, nl):
    """levelCols = (int)sqrt(quota / (5 ratio)) is 0 for the top levels of a small feature budget (150 features over 8 levels:
    the reference's loops over levelRows x levelCols visit nothing, ORBextractor.cpp:541-716) - such a level contributes no key
    point, the others are unaffected.  Rounds 1-4 refused the whole extractor ("unsupported cell grid 0x0")."""
    from se2lam_amd.orb import ORBextractor
    img = np.ascontiguousarray(synth.texture()[10:10 + rows, 20:20 + cols])
    p = oracle.orb_params(nfeatures=nf, nlevels=nl)
    ko, do = oracle.orb_extract(img, p)
    ex = ORBextractor(nfeatures=nf, nlevels=nl, max_rows=rows, max_cols=cols, max_batch=3)
    k, d = ex(img)
    assert np.array_equal(k, ko) and np.array_equal(d, do), (len(k), len(ko))
    assert (np.bincount(ko["octave"], minlength=nl) == 0).any() or nf == 150 and (rows, cols) == (480, 640)
    out = ex.extract_batch(np.stack([img, img[::-1].copy(), img]))
    assert np.array_equal(out[2][0], ko) and np.array_equal(out[2][1], do)


@pytest.mark.parametrize("rows,cols,nf,nl", [(653, 1005, 5135, 1), (501, 715, 2786, 2), (884, 253, 3225, 1)])
def test_cell_grids_of_more_than_256_cells(oracle, synth, rows, cols, nf, nl):
    """many features on few levels: 950, 266 and 611 cells at level 0 (rounds 1-4: "unsupported cell grid", limit 256; now 1024)"""
    from se2lam_amd.orb import ORBextractor
    img = np.ascontiguousarray(synth.texture()[0:rows, 0:cols])
    p = oracle.orb_params(nfeatures=nf, nlevels=nl)
    ko, do = oracle.orb_extract(img, p, cap=16384)
    k, d = ORBextractor(nfeatures=nf, nlevels=nl, max_rows=rows, max_cols=cols)(img)
    assert len(ko) > 1000 and np.array_equal(k, ko) and np.array_equal(d, do)
