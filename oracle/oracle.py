"""ctypes loader for the CPU oracle (oracle/_build/liboracle.so).

ORACLE - TEST INFRASTRUCTURE ONLY.  Importable from tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg; never from se2lam_amd/ (the product path).
PARITY: front end pinned against the reference's own compiled sources (oracle/ref.py, oracle/_ref), third-party arithmetic
(OpenCV, g2o) and the bundle adjustments unpinned - see the headers of oracle/*.cpp and DESIGN.md section 3.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, "_build", "liboracle.so")


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in os.listdir(_HERE) if f.endswith((".cpp", ".h"))]
    stale = (not os.path.exists(_LIB)) or any(os.path.getmtime(s) > os.path.getmtime(_LIB) for s in srcs)
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-s"])
    return _LIB


_lib = None


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        _lib = C.CDLL(build())
        _declare(_lib)
    return _lib


def _p(a, ty):
    return a.ctypes.data_as(C.POINTER(ty))


# --------------------------------------------------------------------------------------
# BA
# --------------------------------------------------------------------------------------
class BaProblem(C.Structure):
    _fields_ = [
        ("P", C.c_int32), ("L", C.c_int32), ("E", C.c_int32), ("O", C.c_int32),
        ("poses", C.POINTER(C.c_double)), ("fixed", C.POINTER(C.c_uint8)), ("lms", C.POINTER(C.c_double)),
        ("e_kf", C.POINTER(C.c_int32)), ("e_lm", C.POINTER(C.c_int32)),
        ("e_uv", C.POINTER(C.c_double)), ("e_info", C.POINTER(C.c_double)),
        ("o_i", C.POINTER(C.c_int32)), ("o_j", C.POINTER(C.c_int32)),
        ("o_meas", C.POINTER(C.c_double)), ("o_info", C.POINTER(C.c_double)),
        ("fx", C.c_double), ("cx", C.c_double), ("cy", C.c_double),
        ("Rbc", C.c_double * 9), ("tbc", C.c_double * 3), ("huber", C.c_double),
    ]


class BaStats(C.Structure):
    _fields_ = [
        ("iterations", C.c_int32), ("trials", C.c_int32), ("terminated", C.c_int32),
        ("chi2_init", C.c_double), ("chi2_final", C.c_double), ("lambda_final", C.c_double),
        ("chi2_hist", C.c_double * 64), ("lambda_hist", C.c_double * 64), ("trials_hist", C.c_int32 * 64),
        ("n_rho", C.c_int32), ("rho_log", C.c_double * 256),
    ]


def _declare(l):
    l.ba_ref_chi2.restype = C.c_double
    l.ba_ref_chi2.argtypes = [C.POINTER(BaProblem), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    l.ba_ref_optimize.restype = C.c_int
    l.ba_ref_optimize.argtypes = [C.POINTER(BaProblem), C.c_int, C.c_int, C.POINTER(C.c_uint8),
                                  C.POINTER(C.c_double), C.POINTER(C.c_double), C.POINTER(BaStats)]
    l.ba_ref_reduced_system.restype = None
    l.ba_ref_reduced_system.argtypes = [C.POINTER(BaProblem)] + [C.POINTER(C.c_double)] * 2 + [C.c_double] + \
        [C.POINTER(C.c_double)] * 5
    l.ba_ref_edge_se2xyz.restype = None
    l.ba_ref_edge_se2xyz.argtypes = [C.POINTER(BaProblem)] + [C.POINTER(C.c_double)] * 6
    l.ba_ref_edge_pre_se2.restype = None
    l.ba_ref_edge_pre_se2.argtypes = [C.POINTER(C.c_double)] * 6
    if hasattr(l, "orb_ref_extract"):
        _declare_orb(l)
    if hasattr(l, "match_ref_window"):
        _declare_match(l)


class _Keep:
    """Holds numpy arrays alive for the lifetime of a ctypes problem struct."""

    def __init__(self):
        self.arrs = []

    def arr(self, a, dtype):
        a = np.ascontiguousarray(a, dtype=dtype)
        self.arrs.append(a)
        return a


def ba_problem(g):
    """synth.BAGraph -> (BaProblem, keepalive)."""
    k = _Keep()
    pr = BaProblem()
    pr.P, pr.L, pr.E, pr.O = g.P, g.L, g.E, g.O
    pr.poses = _p(k.arr(g.poses, np.float64), C.c_double)
    pr.fixed = _p(k.arr(g.fixed, np.uint8), C.c_uint8)
    pr.lms = _p(k.arr(g.lms, np.float64), C.c_double)
    pr.e_kf = _p(k.arr(g.e_kf, np.int32), C.c_int32)
    pr.e_lm = _p(k.arr(g.e_lm, np.int32), C.c_int32)
    pr.e_uv = _p(k.arr(g.e_uv, np.float64), C.c_double)
    pr.e_info = _p(k.arr(g.e_info, np.float64), C.c_double)
    pr.o_i = _p(k.arr(g.o_i, np.int32), C.c_int32)
    pr.o_j = _p(k.arr(g.o_j, np.int32), C.c_int32)
    pr.o_meas = _p(k.arr(g.o_meas, np.float64), C.c_double)
    pr.o_info = _p(k.arr(g.o_info, np.float64), C.c_double)
    pr.fx, pr.cx, pr.cy = g.fx, g.cx, g.cy
    pr.Rbc = (C.c_double * 9)(*np.asarray(g.Rbc, dtype=np.float64).reshape(-1))
    pr.tbc = (C.c_double * 3)(*np.asarray(g.tbc, dtype=np.float64).reshape(-1))
    pr.huber = g.huber
    return pr, k


def ba_chi2(g, poses=None, lms=None) -> float:
    pr, k = ba_problem(g)
    poses = k.arr(g.poses if poses is None else poses, np.float64)
    lms = k.arr(g.lms if lms is None else lms, np.float64)
    return float(lib().ba_ref_chi2(C.byref(pr), _p(poses, C.c_double), _p(lms, C.c_double)))


def ba_optimize(g, iters=10, mode=0):
    """-> (poses (P,3), lms (L,3), stats dict)"""
    pr, k = ba_problem(g)
    poses = np.zeros((g.P, 3))
    lms = np.zeros((g.L, 3))
    st = BaStats()
    rc = lib().ba_ref_optimize(C.byref(pr), iters, mode, None, _p(poses, C.c_double), _p(lms, C.c_double), C.byref(st))
    assert rc == 0
    n = min(st.iterations, 64)
    stats = dict(iterations=st.iterations, trials=st.trials, terminated=bool(st.terminated),
                 chi2_init=st.chi2_init, chi2_final=st.chi2_final, lambda_final=st.lambda_final,
                 chi2_hist=list(st.chi2_hist[:n]), lambda_hist=list(st.lambda_hist[:n]),
                 trials_hist=list(st.trials_hist[:n]), rho_log=list(st.rho_log[:st.n_rho]))
    return poses, lms, stats


def _lapack_pointers():
    """(dpotrf, dpotrs) function pointers of scipy's LAPACK (OpenBLAS, threaded) or (None, None)."""
    try:
        import scipy.linalg.cython_lapack as cl
        C.pythonapi.PyCapsule_GetPointer.restype = C.c_void_p
        C.pythonapi.PyCapsule_GetPointer.argtypes = [C.py_object, C.c_char_p]
        C.pythonapi.PyCapsule_GetName.restype = C.c_char_p
        C.pythonapi.PyCapsule_GetName.argtypes = [C.py_object]
        out = []
        for name in ("dpotrf", "dpotrs"):
            cap = cl.__pyx_capi__[name]
            out.append(C.pythonapi.PyCapsule_GetPointer(cap, C.pythonapi.PyCapsule_GetName(cap)))
        return tuple(out)
    except Exception:
        return (None, None)


def ba_mt_threads() -> int:
    f = lib().ba_ref_mt_threads
    f.restype = C.c_int
    return int(f())


def ba_optimize_mt(g, iters=10, threads=0, lapack=True):
    """ALL-CORE CPU baseline (oracle/ba_ref_mt.cpp: OpenMP over landmarks / pose rows + LAPACK dpotrf for the pose
    solve) -> (poses, lms, stats); same algorithm and LM policy as ba_optimize."""
    pr, k = ba_problem(g)
    poses = np.zeros((g.P, 3))
    lms = np.zeros((g.L, 3))
    st = BaStats()
    f = lib().ba_ref_optimize_mt
    f.restype = C.c_int
    f.argtypes = [C.POINTER(BaProblem), C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_double),
                  C.POINTER(C.c_double), C.POINTER(BaStats)]
    pf, ps = _lapack_pointers() if lapack else (None, None)
    rc = f(C.byref(pr), iters, threads, pf, ps, _p(poses, C.c_double), _p(lms, C.c_double), C.byref(st))
    assert rc == 0
    n = min(st.iterations, 64)
    stats = dict(iterations=st.iterations, trials=st.trials, terminated=bool(st.terminated),
                 chi2_init=st.chi2_init, chi2_final=st.chi2_final, lambda_final=st.lambda_final,
                 chi2_hist=list(st.chi2_hist[:n]), lambda_hist=list(st.lambda_hist[:n]),
                 trials_hist=list(st.trials_hist[:n]), rho_log=list(st.rho_log[:st.n_rho]), lapack=pf is not None)
    return poses, lms, stats


def ba_reduced_system(g, lam, poses=None, lms=None):
    """-> dict(S (3P,3P), bs, bp, bl, Hll (L,3,3)) at the given state and damping."""
    pr, k = ba_problem(g)
    poses = k.arr(g.poses if poses is None else poses, np.float64)
    lms = k.arr(g.lms if lms is None else lms, np.float64)
    n = 3 * g.P
    S = np.zeros((n, n)); bs = np.zeros(n); bp = np.zeros(n)
    bl = np.zeros((g.L, 3)); Hll = np.zeros((g.L, 3, 3))
    lib().ba_ref_reduced_system(C.byref(pr), _p(poses, C.c_double), _p(lms, C.c_double), float(lam),
                                _p(S, C.c_double), _p(bs, C.c_double), _p(bp, C.c_double),
                                _p(bl, C.c_double), _p(Hll, C.c_double))
    return dict(S=S, bs=bs, bp=bp, bl=bl, Hll=Hll)


def ba_edge_se2xyz(g, pose, lw, uv):
    pr, k = ba_problem(g)
    pose = k.arr(pose, np.float64); lw = k.arr(lw, np.float64); uv = k.arr(uv, np.float64)
    e = np.zeros(2); Jp = np.zeros((2, 3)); Jl = np.zeros((2, 3))
    lib().ba_ref_edge_se2xyz(C.byref(pr), _p(pose, C.c_double), _p(lw, C.c_double), _p(uv, C.c_double),
                             _p(e, C.c_double), _p(Jp, C.c_double), _p(Jl, C.c_double))
    return e, Jp, Jl


def ba_edge_information(lc, lw, e_kf, sigma2, Rcw, twb_xy, fx, xrot_info=1e6, z_info=1.0):
    lc = np.ascontiguousarray(lc, np.float32); lw = np.ascontiguousarray(lw, np.float32)
    e_kf = np.ascontiguousarray(e_kf, np.int32); sigma2 = np.ascontiguousarray(sigma2, np.float32)
    Rcw = np.ascontiguousarray(Rcw, np.float32).reshape(-1, 9); twb_xy = np.ascontiguousarray(twb_xy, np.float32)
    E = len(e_kf)
    out = np.zeros((max(E, 1), 2, 2))
    f = lib().ba_ref_edge_information
    f.restype = None
    f.argtypes = [C.c_int] + [C.c_void_p] * 4 + [C.c_int] + [C.c_void_p] * 2 + [C.c_float] * 3 + [C.c_void_p]
    f(E, lc.ctypes.data, lw.ctypes.data, e_kf.ctypes.data, sigma2.ctypes.data, len(Rcw), Rcw.ctypes.data,
      twb_xy.ctypes.data, fx, xrot_info, z_info, out.ctypes.data)
    return out[:E]


def ba_edge_pre_se2(pi, pj, z):
    pi = np.ascontiguousarray(pi, np.float64); pj = np.ascontiguousarray(pj, np.float64)
    z = np.ascontiguousarray(z, np.float64)
    e = np.zeros(3); Ji = np.zeros((3, 3)); Jj = np.zeros((3, 3))
    lib().ba_ref_edge_pre_se2(_p(pi, C.c_double), _p(pj, C.c_double), _p(z, C.c_double),
                              _p(e, C.c_double), _p(Ji, C.c_double), _p(Jj, C.c_double))
    return e, Ji, Jj


# --------------------------------------------------------------------------------------
# ORB extractor
# --------------------------------------------------------------------------------------
class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("fast_th", C.c_int32), ("score_type", C.c_int32)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])


def _declare_orb(l):
    VP = C.c_void_p
    PI = C.POINTER(C.c_int)
    l.orb_ref_extract.restype = C.c_int
    l.orb_ref_extract.argtypes = [C.POINTER(OrbParams), VP, C.c_int, C.c_int, C.c_int, VP, VP, C.c_int, PI]
    l.orb_ref_tables.restype = None
    l.orb_ref_tables.argtypes = [C.POINTER(OrbParams), VP, VP, VP, VP]
    l.orb_ref_geometry.restype = None
    l.orb_ref_geometry.argtypes = [C.POINTER(OrbParams), C.c_int, C.c_int, VP]
    l.orb_ref_level.restype = C.c_int
    l.orb_ref_level.argtypes = [C.POINTER(OrbParams), VP, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, VP, PI, PI]
    l.orb_ref_score.restype = C.c_int
    l.orb_ref_score.argtypes = [C.POINTER(OrbParams), VP, C.c_int, C.c_int, C.c_int, C.c_int, VP]
    l.orb_ref_pattern.restype = None
    l.orb_ref_pattern.argtypes = [VP]
    l.orb_ref_gaussian_taps.restype = None
    l.orb_ref_gaussian_taps.argtypes = [VP]
    l.orb_ref_fast_atan2.restype = C.c_float
    l.orb_ref_fast_atan2.argtypes = [C.c_float, C.c_float]
    l.orb_ref_cv_round.restype = C.c_int
    l.orb_ref_cv_round.argtypes = [C.c_float]
    l.orb_ref_fast_score.restype = C.c_int
    l.orb_ref_fast_score.argtypes = [VP, C.c_int]


FAST_SCORE, HARRIS_SCORE = 1, 0   # cv::ORB enum values the reference's constructor takes (ORBextractor.h:44)


def orb_params(nfeatures=1000, scale_factor=1.2, nlevels=8, fast_th=20, score_type=FAST_SCORE):
    return OrbParams(nfeatures, scale_factor, nlevels, fast_th, score_type)


def orb_extract(img, params=None, cap=4096):
    """-> (keypoints structured array (n,), descriptors (n,32) u8)"""
    params = params or orb_params()
    img = np.ascontiguousarray(img, np.uint8)
    rows, cols = (img.shape if img.ndim == 2 else (0, 0))
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = C.c_int(0)
    rc = lib().orb_ref_extract(C.byref(params), img.ctypes.data, rows, cols, cols, kps.ctypes.data,
                               desc.ctypes.data, cap, C.byref(n))
    assert rc == 0, "orb_ref_extract: capacity exceeded"
    return kps[:n.value].copy(), desc[:n.value].copy()


def orb_trig_libm(mode):
    """descriptor steering angle: 0 / False = the rounded double cosine / sine (default), 1 / True = libm's cosf / sinf (what the
    reference's C++ resolves to), 2 = glibc's sinf / cosf algorithm written out in double arithmetic - see orb_ref.cpp"""
    f = lib().orb_ref_set_trig_libm
    f.restype = None
    f.argtypes = [C.c_int]
    f(int(mode))


def orb_retain_stable(on):
    """retainBest order: False (default) = libstdc++'s nth_element restated (stl_nth.h) = what a GCC build of the reference keeps;
    True = the order rounds 1-4 defined (best n by response, ties by list position, list order kept)"""
    f = lib().orb_ref_set_retain_stable
    f.restype = None
    f.argtypes = [C.c_int]
    f(int(bool(on)))


def nth_element(entries, nth, std=False):
    """entries: uint64, high 32 bits = key (larger is better).  The permutation libstdc++'s std::nth_element leaves, by the
    restatement in stl_nth.h (std=True: by this machine's std::nth_element itself)"""
    e = np.ascontiguousarray(entries, np.uint64).copy()
    f = lib().orb_ref_std_nth_element if std else lib().orb_ref_nth_element
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int]
    f(e.ctypes.data, len(e), int(nth))
    return e


def nth_heap_selects():
    f = lib().orb_ref_nth_heap_selects
    f.restype = C.c_long
    return f()


def nth_killer(n, nth):
    """keys (uint32, larger is better) on which libstdc++'s introselect runs out of its depth limit (median-of-three killer)"""
    k = np.zeros(n, np.uint32)
    f = lib().orb_ref_nth_killer
    f.restype = None
    f.argtypes = [C.c_int, C.c_int, C.c_void_p]
    f(int(n), int(nth), k.ctypes.data)
    return k


def glibc_sincosf(y, cosine: bool) -> float:
    f = lib().orb_ref_glibc_sincosf
    f.restype = C.c_float
    f.argtypes = [C.c_float, C.c_int]
    return f(float(y), 1 if cosine else 0)


def orb_tables(params=None):
    params = params or orb_params()
    L = params.nlevels
    scale = np.zeros(L, np.float32); inv = np.zeros(L, np.float32)
    quota = np.zeros(L, np.int32); umax = np.zeros(16, np.int32)
    lib().orb_ref_tables(C.byref(params), scale.ctypes.data, inv.ctypes.data, quota.ctypes.data, umax.ctypes.data)
    return dict(scale=scale, inv_scale=inv, quota=quota, umax=umax)


def orb_geometry(rows, cols, params=None):
    params = params or orb_params()
    out = np.zeros((params.nlevels, 7), np.int32)
    lib().orb_ref_geometry(C.byref(params), rows, cols, out.ctypes.data)
    return out


def orb_level(img, level, blurred=False, params=None, bordered=False):
    params = params or orb_params()
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros((img.shape[0] + 32) * (img.shape[1] + 32), np.uint8)
    w = C.c_int(); h = C.c_int()
    lib().orb_ref_level(C.byref(params), img.ctypes.data, img.shape[0], img.shape[1], img.shape[1], level,
                        int(blurred) | (2 if bordered else 0), out.ctypes.data, C.byref(w), C.byref(h))
    return out[:w.value * h.value].reshape(h.value, w.value).copy()


def orb_score(img, level, params=None):
    params = params or orb_params()
    lv = orb_level(img, level, False, params)
    img = np.ascontiguousarray(img, np.uint8)
    out = np.zeros(lv.shape, np.uint8)
    lib().orb_ref_score(C.byref(params), img.ctypes.data, img.shape[0], img.shape[1], img.shape[1], level,
                        out.ctypes.data)
    return out


def orb_pattern():
    out = np.zeros(1024, np.int32)
    lib().orb_ref_pattern(out.ctypes.data)
    return out


def orb_gaussian_taps():
    out = np.zeros(7, np.int32)
    lib().orb_ref_gaussian_taps(out.ctypes.data)
    return out


# --------------------------------------------------------------------------------------
# matcher
# --------------------------------------------------------------------------------------
class Bounds(C.Structure):
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float)]


def _declare_match(l):
    VP = C.c_void_p
    l.match_ref_hamming.restype = C.c_int
    l.match_ref_hamming.argtypes = [VP, VP]
    l.match_ref_features_in_area.restype = C.c_int
    l.match_ref_features_in_area.argtypes = [C.POINTER(Bounds), VP, C.c_int, C.c_float, C.c_float, C.c_float,
                                             C.c_int, C.c_int, VP, C.c_int]
    l.match_ref_window.restype = C.c_int
    l.match_ref_window.argtypes = [C.POINTER(Bounds), VP, VP, C.c_int, VP, VP, C.c_int, VP, C.c_int, C.c_int,
                                   C.c_int, C.c_int, C.c_float, VP]
    l.match_ref_projection.restype = C.c_int
    l.match_ref_projection.argtypes = [C.POINTER(Bounds), VP, VP, VP, VP, C.c_int, VP, C.c_float, C.c_float,
                                       C.c_float, C.c_float, VP, VP, VP, C.c_int, C.c_int, C.c_int, C.c_float, VP]


def search_by_bow(kps1, desc1, fv1, has_mp1, kps2, desc2, fv2, has_mp2, mp_only=False, nnratio=0.6, check_ori=True):
    """fv = (nodes, ptr, idx) CSR int32 arrays -> (matches12 (n1,), nmatches)"""
    kps1 = np.ascontiguousarray(kps1); kps2 = np.ascontiguousarray(kps2)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    a = [np.ascontiguousarray(x, np.int32) for x in fv1]; b = [np.ascontiguousarray(x, np.int32) for x in fv2]
    h1 = np.ascontiguousarray(has_mp1, np.uint8); h2 = np.ascontiguousarray(has_mp2, np.uint8)
    out = np.full(max(len(kps1), 1), -1, np.int32)
    f = lib().match_ref_search_by_bow
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p] + [C.c_void_p] * 2 + \
        [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    nm = f(kps1.ctypes.data, desc1.ctypes.data, len(kps1), a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data,
           len(a[0]), h1.ctypes.data, kps2.ctypes.data, desc2.ctypes.data, len(kps2), b[0].ctypes.data,
           b[1].ctypes.data, b[2].ctypes.data, len(b[0]), h2.ctypes.data, int(mp_only), nnratio, int(check_ori),
           out.ctypes.data)
    return out[:len(kps1)].copy(), int(nm)


def default_bounds(cols=640, rows=480):
    return Bounds(0.0, 0.0, float(cols), float(rows))


def hamming(a, b) -> int:
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    return int(lib().match_ref_hamming(a.ctypes.data, b.ctypes.data))


def features_in_area(kps, x, y, r, min_level, max_level, bounds=None):
    bounds = bounds or default_bounds()
    kps = np.ascontiguousarray(kps)
    out = np.zeros(max(len(kps), 1), np.int32)
    n = lib().match_ref_features_in_area(C.byref(bounds), kps.ctypes.data, len(kps), x, y, r, min_level, max_level,
                                         out.ctypes.data, out.size)
    return out[:n].copy()


def match_window(kps1, desc1, kps2, desc2, prev_xy=None, win=20, level_offset=1, min_level=0, max_level=8,
                 nnratio=0.9, bounds=None):
    """-> (matches12 (n1,), nmatches, prev_xy updated)"""
    bounds = bounds or default_bounds()
    kps1 = np.ascontiguousarray(kps1); kps2 = np.ascontiguousarray(kps2)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    n1, n2 = len(kps1), len(kps2)
    if prev_xy is None:
        prev_xy = np.stack([kps1["x"], kps1["y"]], axis=1)
    prev = np.ascontiguousarray(prev_xy, np.float32).copy()
    m12 = np.full(max(n1, 1), -1, np.int32)
    nm = lib().match_ref_window(C.byref(bounds), kps1.ctypes.data, desc1.ctypes.data, n1, kps2.ctypes.data,
                                desc2.ctypes.data, n2, prev.ctypes.data, win, level_offset, min_level, max_level,
                                nnratio, m12.ctypes.data)
    return m12[:n1].copy(), int(nm), prev


def match_projection(mp_pos, mp_desc, mp_octave, mp_skip, Tcw, K4, kps, desc, kf_observed, win=15, level_offset=2,
                     nnratio=0.6, bounds=None):
    """-> (match_idx_mp (n,), nmatches)"""
    bounds = bounds or default_bounds()
    mp_pos = np.ascontiguousarray(mp_pos, np.float32); mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
    mp_octave = np.ascontiguousarray(mp_octave, np.int32); mp_skip = np.ascontiguousarray(mp_skip, np.uint8)
    Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(-1)[:12].copy()
    kps = np.ascontiguousarray(kps); desc = np.ascontiguousarray(desc, np.uint8)
    kf_observed = np.ascontiguousarray(kf_observed, np.uint8)
    n, m = len(kps), len(mp_octave)
    out = np.full(max(n, 1), -1, np.int32)
    fx, fy, cx, cy = [float(v) for v in K4]
    nm = lib().match_ref_projection(C.byref(bounds), mp_pos.ctypes.data, mp_desc.ctypes.data, mp_octave.ctypes.data,
                                    mp_skip.ctypes.data, m, Tcw.ctypes.data, fx, fy, cx, cy, kps.ctypes.data,
                                    desc.ctypes.data, kf_observed.ctypes.data, n, win, level_offset, nnratio,
                                    out.ctypes.data)
    return out[:n].copy(), int(nm)


def triangulate(kps_ref, kps_cur, match_idx, has_obs, P_ref, P_cur, Ocam, lower, upper, min_degree=2):
    """Track::doTriangulate over all matches -> (pos (n,3) f32, good (n,) u8, match_idx updated, n_good, n_tracked_old)"""
    kps_ref = np.ascontiguousarray(kps_ref); kps_cur = np.ascontiguousarray(kps_cur)
    n = len(kps_ref)
    m = np.ascontiguousarray(match_idx, np.int32).copy()
    ho = None if has_obs is None else np.ascontiguousarray(has_obs, np.uint8)
    P1 = np.ascontiguousarray(P_ref, np.float32).reshape(-1); P2 = np.ascontiguousarray(P_cur, np.float32).reshape(-1)
    oc = np.ascontiguousarray(Ocam, np.float32)
    pos = np.zeros((max(n, 1), 3), np.float32)
    good = np.zeros(max(n, 1), np.uint8)
    nold = C.c_int(0)
    f = lib().match_ref_triangulate
    f.restype = C.c_int
    f.argtypes = [C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                  C.c_float, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
    ng = f(n, kps_ref.ctypes.data, kps_cur.ctypes.data, len(kps_cur), m.ctypes.data,
           None if ho is None else ho.ctypes.data, P1.ctypes.data, P2.ctypes.data, oc.ctypes.data, lower, upper,
           min_degree, pos.ctypes.data, good.ctypes.data, C.byref(nold))
    return pos[:n], good[:n], m, int(ng), int(nold.value)


def fundamental_mask(pt1, pt2):
    """cv::findFundamentalMat(pt1, pt2, mask) with the defaults (FM_RANSAC, 3 px, 0.99) -> (mask (n,) u8, n_inliers)"""
    p1 = np.ascontiguousarray(pt1, np.float32).reshape(-1, 2)
    p2 = np.ascontiguousarray(pt2, np.float32).reshape(-1, 2)
    n = len(p1)
    mask = np.zeros(max(n, 1), np.uint8)
    f = lib().match_ref_fundamental_mask
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    ni = f(p1.ctypes.data, p2.ctypes.data, n, mask.ctypes.data)
    return mask[:n], int(ni)


def remove_outliers(kps1, kps2, matches):
    """Track::removeOutliers -> (matches with the outliers set to -1, n_inliers)"""
    k1 = np.ascontiguousarray(kps1); k2 = np.ascontiguousarray(kps2)
    m = np.ascontiguousarray(matches, np.int32).copy()
    f = lib().match_ref_remove_outliers
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p]
    ni = f(k1.ctypes.data, len(k1), k2.ctypes.data, len(k2), m.ctypes.data)
    return m, int(ni)


def _pose12(T):
    T = np.asarray(T, np.float64)
    if T.shape == (12,):
        return np.ascontiguousarray(T)
    return np.ascontiguousarray(np.concatenate([T[:3, :3].reshape(-1), T[:3, 3]]))


def pose12_to_matrix(p):
    T = np.eye(4)
    T[:3, :3] = np.asarray(p[:9]).reshape(3, 3)
    T[:3, 3] = p[9:12]
    return T


def plane_motion_prior(Tcw, Tbc, xrot_info=1e6, yrot_info=1e6, z_info=1.0):
    """addPlaneMotionSE3Expmap (optimizer.cpp:236-314) -> (measurement 4x4, information 6x6)"""
    meas = np.zeros(12); info = np.zeros(36)
    f = lib().ba_ref_plane_motion_prior
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    a, b = _pose12(Tcw), _pose12(Tbc)
    f(a.ctypes.data, b.ctypes.data, xrot_info, yrot_info, z_info, meas.ctypes.data, info.ctypes.data)
    return pose12_to_matrix(meas), info.reshape(6, 6)


def pose_only_ba(Tcw, prior_meas, prior_info, xyz, uv, inv_sigma2, f, cx, cy, delta, iters=30):
    """Localizer::DoLocalBA (Localizer.cpp:233-302) -> (Tcw 4x4, stats dict)"""
    xyz = np.ascontiguousarray(xyz, np.float64); uv = np.ascontiguousarray(uv, np.float64)
    w = np.ascontiguousarray(inv_sigma2, np.float64)
    out = np.zeros(12)
    st = BaStats()
    fn = lib().ba_ref_pose_only
    fn.restype = C.c_int
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double,
                   C.c_double, C.c_double, C.c_int, C.c_void_p, C.POINTER(BaStats)]
    a, m = _pose12(Tcw), _pose12(prior_meas)
    pi = np.ascontiguousarray(prior_info, np.float64).reshape(-1)
    rc = fn(a.ctypes.data, m.ctypes.data, pi.ctypes.data, len(xyz), xyz.ctypes.data, uv.ctypes.data, w.ctypes.data, f, cx, cy,
            delta, iters, out.ctypes.data, C.byref(st))
    assert rc == 0
    n = min(st.iterations, 64)
    stats = dict(iterations=st.iterations, trials=st.trials, terminated=bool(st.terminated), chi2_init=st.chi2_init,
                 chi2_final=st.chi2_final, lambda_final=st.lambda_final, chi2_hist=list(st.chi2_hist[:n]),
                 lambda_hist=list(st.lambda_hist[:n]), trials_hist=list(st.trials_hist[:n]))
    return pose12_to_matrix(out), stats


def se3_exp(update6):
    out = np.zeros(12); u = np.ascontiguousarray(update6, np.float64)
    f = lib().ba_ref_se3_exp; f.restype = None; f.argtypes = [C.c_void_p, C.c_void_p]
    f(u.ctypes.data, out.ctypes.data)
    return pose12_to_matrix(out)


def se3_log(T):
    out = np.zeros(6); a = _pose12(T)
    f = lib().ba_ref_se3_log; f.restype = None; f.argtypes = [C.c_void_p, C.c_void_p]
    f(a.ctypes.data, out.ctypes.data)
    return out


def se3_mul(A, B):
    out = np.zeros(12); a, b = _pose12(A), _pose12(B)
    f = lib().ba_ref_se3_mul; f.restype = None; f.argtypes = [C.c_void_p] * 3
    f(a.ctypes.data, b.ctypes.data, out.ctypes.data)
    return pose12_to_matrix(out)


def project_edge(T, X, uv, f, cx, cy):
    """EdgeProjectXYZ2UV -> (error (2,), Jacobian (2,6) w.r.t. the pose update)"""
    e = np.zeros(2); J = np.zeros(12); a = _pose12(T)
    X = np.ascontiguousarray(X, np.float64); uv = np.ascontiguousarray(uv, np.float64)
    fn = lib().ba_ref_project_edge; fn.restype = None
    fn.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    fn(a.ctypes.data, X.ctypes.data, uv.ctypes.data, f, cx, cy, e.ctypes.data, J.ctypes.data)
    return e, J.reshape(2, 6)


def seven_point(pt1, pt2, idx7):
    """run7Point on the correspondences idx7 -> list of 3x3 matrices"""
    p1 = np.ascontiguousarray(pt1, np.float32).reshape(-1, 2); p2 = np.ascontiguousarray(pt2, np.float32).reshape(-1, 2)
    idx = np.ascontiguousarray(idx7, np.int32)
    F = np.zeros(27)
    fn = lib().match_ref_seven_point; fn.restype = C.c_int; fn.argtypes = [C.c_void_p] * 4
    n = fn(p1.ctypes.data, p2.ctypes.data, idx.ctypes.data, F.ctypes.data)
    return [F[9 * k:9 * k + 9].reshape(3, 3).copy() for k in range(max(n, 0))]


def ransac_subsets(n, count):
    out = np.zeros((count, 7), np.int32)
    fn = lib().match_ref_ransac_subsets; fn.restype = None; fn.argtypes = [C.c_int, C.c_int, C.c_void_p]
    fn(n, count, out.ctypes.data)
    return out


# --------------------------------------------------------------------------------------
# SE3-expmap bundle adjustment with marginalised points (oracle/ba3_ref.cpp)
# --------------------------------------------------------------------------------------
class Ba3Problem(C.Structure):
    _fields_ = [("P", C.c_int32), ("L", C.c_int32), ("E", C.c_int32), ("O", C.c_int32),
                ("poses", C.c_void_p), ("fixed", C.c_void_p), ("lms", C.c_void_p), ("e_kf", C.c_void_p),
                ("e_lm", C.c_void_p), ("e_uv", C.c_void_p), ("e_w", C.c_void_p), ("has_prior", C.c_void_p),
                ("prior_meas", C.c_void_p), ("prior_info", C.c_void_p), ("o_i", C.c_void_p), ("o_j", C.c_void_p),
                ("o_meas", C.c_void_p), ("o_info", C.c_void_p),
                ("f", C.c_double), ("cx", C.c_double), ("cy", C.c_double), ("huber", C.c_double)]


def poses12(T):
    """(P,4,4) -> (P,12): rotation row-major then translation"""
    T = np.asarray(T, np.float64)
    return np.ascontiguousarray(np.concatenate([T[:, :3, :3].reshape(-1, 9), T[:, :3, 3]], axis=1))


def poses44(p12):
    p12 = np.asarray(p12).reshape(-1, 12)
    T = np.tile(np.eye(4), (len(p12), 1, 1))
    T[:, :3, :3] = p12[:, :9].reshape(-1, 3, 3)
    T[:, :3, 3] = p12[:, 9:]
    return T


def ba3_problem(g):
    k = _Keep()
    pr = Ba3Problem()
    pr.P, pr.L, pr.E, pr.O = g.P, g.L, g.E, g.O
    arrs = dict(poses=(poses12(g.poses), np.float64), fixed=(g.fixed, np.uint8), lms=(g.lms, np.float64),
                e_kf=(g.e_kf, np.int32), e_lm=(g.e_lm, np.int32), e_uv=(g.e_uv, np.float64), e_w=(g.e_w, np.float64),
                has_prior=(g.has_prior, np.uint8), prior_meas=(poses12(g.prior_meas), np.float64),
                prior_info=(g.prior_info.reshape(g.P, 36), np.float64), o_i=(g.o_i, np.int32), o_j=(g.o_j, np.int32),
                o_meas=(poses12(g.o_meas) if g.O else np.zeros((0, 12)), np.float64),
                o_info=(g.o_info.reshape(g.O, 36), np.float64))
    for name, (a, dt) in arrs.items():
        setattr(pr, name, k.arr(a, dt).ctypes.data)
    pr.f, pr.cx, pr.cy, pr.huber = g.fx, g.cx, g.cy, g.huber
    return pr, k


def ba3_chi2(g, poses=None, lms=None):
    """-> (robust chi2, per-edge chi2 (E,))"""
    pr, k = ba3_problem(g)
    p = k.arr(poses12(g.poses if poses is None else poses), np.float64)
    l = k.arr(g.lms if lms is None else lms, np.float64)
    ec = np.zeros(max(g.E, 1))
    f = lib().ba3_ref_chi2
    f.restype = C.c_double
    f.argtypes = [C.POINTER(Ba3Problem), C.c_void_p, C.c_void_p, C.c_void_p]
    return float(f(C.byref(pr), p.ctypes.data, l.ctypes.data, ec.ctypes.data)), ec[:g.E]


def ba3_reduced_system(g, lam):
    pr, k = ba3_problem(g)
    n = 6 * g.P
    S = np.zeros((n, n)); bs = np.zeros(n)
    p = k.arr(poses12(g.poses), np.float64); l = k.arr(g.lms, np.float64)
    f = lib().ba3_ref_reduced_system
    f.restype = None
    f.argtypes = [C.POINTER(Ba3Problem), C.c_void_p, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    f(C.byref(pr), p.ctypes.data, l.ctypes.data, float(lam), S.ctypes.data, bs.ctypes.data)
    return S, bs


def ba3_optimize(g, iters=10):
    """-> (poses (P,4,4), lms (L,3), per-edge chi2 (E,), stats)"""
    pr, k = ba3_problem(g)
    p = np.zeros((g.P, 12)); l = np.zeros((max(g.L, 1), 3)); ec = np.zeros(max(g.E, 1))
    st = BaStats()
    f = lib().ba3_ref_optimize
    f.restype = C.c_int
    f.argtypes = [C.POINTER(Ba3Problem), C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.POINTER(BaStats)]
    rc = f(C.byref(pr), iters, p.ctypes.data, l.ctypes.data, ec.ctypes.data, C.byref(st))
    assert rc == 0
    n = min(st.iterations, 64)
    stats = dict(iterations=st.iterations, trials=st.trials, terminated=bool(st.terminated), chi2_init=st.chi2_init,
                 chi2_final=st.chi2_final, lambda_final=st.lambda_final, chi2_hist=list(st.chi2_hist[:n]),
                 lambda_hist=list(st.lambda_hist[:n]), trials_hist=list(st.trials_hist[:n]), rho_log=list(st.rho_log[:st.n_rho]))
    return poses44(p), l[:g.L], ec[:g.E], stats


def ba3_odo_edge(Ti, Tj, Cm):
    e = np.zeros(6); Ji = np.zeros(36); Jj = np.zeros(36)
    a, b, c = [np.ascontiguousarray(poses12(np.asarray(x)[None])[0]) for x in (Ti, Tj, Cm)]
    f = lib().ba3_ref_odo_edge
    f.restype = None
    f.argtypes = [C.c_void_p] * 6
    f(a.ctypes.data, b.ctypes.data, c.ctypes.data, e.ctypes.data, Ji.ctypes.data, Jj.ctypes.data)
    return e, Ji.reshape(6, 6), Jj.reshape(6, 6)


# --------------------------------------------------------------------------------------
# pose graph of GlobalMapper::GlobalBA (oracle/pg_ref.cpp)
# --------------------------------------------------------------------------------------
class PgProblem(C.Structure):
    _fields_ = [("P", C.c_int32), ("O", C.c_int32), ("poses", C.c_void_p), ("fixed", C.c_void_p), ("has_prior", C.c_void_p),
                ("prior_meas", C.c_void_p), ("prior_info", C.c_void_p), ("o_i", C.c_void_p), ("o_j", C.c_void_p),
                ("o_meas", C.c_void_p), ("o_info", C.c_void_p)]


def pg_problem(g):
    k = _Keep()
    pr = PgProblem()
    pr.P, pr.O = g.P, g.O
    arrs = dict(poses=(poses12(g.poses), np.float64), fixed=(g.fixed, np.uint8), has_prior=(g.has_prior, np.uint8),
                prior_meas=(poses12(g.prior_meas), np.float64), prior_info=(g.prior_info.reshape(g.P, 36), np.float64),
                o_i=(g.o_i, np.int32), o_j=(g.o_j, np.int32), o_meas=(poses12(g.o_meas), np.float64),
                o_info=(g.o_info.reshape(g.O, 36), np.float64))
    for name, (a, dt) in arrs.items():
        setattr(pr, name, k.arr(a, dt).ctypes.data)
    return pr, k


def pg_chi2(g, poses=None):
    pr, k = pg_problem(g)
    p = k.arr(poses12(g.poses if poses is None else poses), np.float64)
    ec = np.zeros(max(g.O, 1))
    f = lib().pg_ref_chi2
    f.restype = C.c_double
    f.argtypes = [C.POINTER(PgProblem), C.c_void_p, C.c_void_p]
    return float(f(C.byref(pr), p.ctypes.data, ec.ctypes.data)), ec[:g.O]


def pg_system(g, lam=0.0):
    pr, k = pg_problem(g)
    n = 6 * g.P
    H = np.zeros((n, n)); b = np.zeros(n)
    p = k.arr(poses12(g.poses), np.float64)
    f = lib().pg_ref_system
    f.restype = None
    f.argtypes = [C.POINTER(PgProblem), C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
    f(C.byref(pr), p.ctypes.data, float(lam), H.ctypes.data, b.ctypes.data)
    return H, b


def pg_optimize(g, iters=10):
    """-> (poses (P,4,4), per-edge chi2 (O,), stats)"""
    pr, k = pg_problem(g)
    p = np.zeros((g.P, 12)); ec = np.zeros(max(g.O, 1))
    st = BaStats()
    f = lib().pg_ref_optimize
    f.restype = C.c_int
    f.argtypes = [C.POINTER(PgProblem), C.c_int, C.c_void_p, C.c_void_p, C.POINTER(BaStats)]
    assert f(C.byref(pr), iters, p.ctypes.data, ec.ctypes.data, C.byref(st)) == 0
    n = min(st.iterations, 64)
    stats = dict(iterations=st.iterations, trials=st.trials, terminated=bool(st.terminated), chi2_init=st.chi2_init,
                 chi2_final=st.chi2_final, lambda_final=st.lambda_final, chi2_hist=list(st.chi2_hist[:n]),
                 lambda_hist=list(st.lambda_hist[:n]), trials_hist=list(st.trials_hist[:n]), rho_log=list(st.rho_log[:st.n_rho]))
    return poses44(p), ec[:g.O], stats


def pg_edge(Xi, Xj, Z):
    e = np.zeros(6); Ji = np.zeros(36); Jj = np.zeros(36)
    a, b, c = [np.ascontiguousarray(poses12(np.asarray(x)[None])[0]) for x in (Xi, Xj, Z)]
    f = lib().pg_ref_edge
    f.restype = None
    f.argtypes = [C.c_void_p] * 6
    f(a.ctypes.data, b.ctypes.data, c.ctypes.data, e.ctypes.data, Ji.ctypes.data, Jj.ctypes.data)
    return e, Ji.reshape(6, 6), Jj.reshape(6, 6)


def pg_oplus(X, upd):
    out = np.zeros(12)
    a = np.ascontiguousarray(poses12(np.asarray(X)[None])[0]); u = np.ascontiguousarray(upd, np.float64)
    f = lib().pg_ref_oplus
    f.restype = None
    f.argtypes = [C.c_void_p] * 3
    f(a.ctypes.data, u.ctypes.data, out.ctypes.data)
    return poses44(out)[0]


def pg_plane_motion_prior(Twc, Tbc, xrot_info=1e6, yrot_info=1e6, z_info=1.0):
    meas = np.zeros(12); info = np.zeros(36)
    a = np.ascontiguousarray(poses12(np.asarray(Twc)[None])[0]); b = np.ascontiguousarray(poses12(np.asarray(Tbc)[None])[0])
    f = lib().pg_ref_plane_motion_prior
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    f(a.ctypes.data, b.ctypes.data, xrot_info, yrot_info, z_info, meas.ctypes.data, info.ctypes.data)
    return poses44(meas)[0], info.reshape(6, 6)


def sparsify(kf, mp, m_kf, m_mp, m_info):
    """Sparsifier::DoMarginalizeSE3XYZ: kf (2,4,4) T_w_c, mp (N,3), measurements (M,), info (M,3,3)
    -> (z_out 4x4 = KF0^-1 KF1, info_out 6x6, H_marginal 12x12)"""
    k12 = poses12(np.asarray(kf))
    mp = np.ascontiguousarray(mp, np.float64); mk = np.ascontiguousarray(m_kf, np.int32); mm = np.ascontiguousarray(m_mp, np.int32)
    mi = np.ascontiguousarray(m_info, np.float64).reshape(-1, 9)
    z = np.zeros(12); info = np.zeros(36); H = np.zeros(144)
    f = lib().sparsify_ref
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f(k12.ctypes.data, len(mp), mp.ctypes.data, len(mk), mk.ctypes.data, mm.ctypes.data, mi.ctypes.data, z.ctypes.data,
      info.ctypes.data, H.ctypes.data)
    return poses44(z)[0], info.reshape(6, 6), H.reshape(12, 12)
