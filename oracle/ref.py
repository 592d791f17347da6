"""ORACLE - TEST INFRASTRUCTURE ONLY.  ctypes loader of oracle/_ref/libse2lam_ref.so: the reference's OWN hot-path sources
(/root/reference/src/ORBextractor.cpp, ORBmatcher.cpp, Frame.cpp, Config.cpp, cvutil.cpp, DBoW2's FeatureVector) compiled
unmodified against oracle/_shim by `make -C oracle ref`.  The library can only be BUILT where /root/reference exists (this
container); it is git-ignored but travels to the GPU box with the snapshot, where the tests load the prebuilt file.  Same
call signatures as the restatement's wrappers in oracle/oracle.py, so a test runs either through one code path.

Only tests/ may import this module (and __graft_entry__.build() to compile it): never the product path."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from .oracle import KP_DTYPE, Bounds, OrbParams, default_bounds, orb_params  # noqa: F401  (same PODs as the restatement)

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "libse2lam_ref.so")
REFERENCE = os.environ.get("SE2LAM_REFERENCE", "/root/reference")
_lib = None


def can_build() -> bool:
    return os.path.isdir(os.path.join(REFERENCE, "src"))


def build(force: bool = False) -> str | None:
    """Compile the reference's sources where they lie; returns the path, or None when /root/reference is absent (GPU box)."""
    if not can_build():
        return LIB if os.path.exists(LIB) else None
    if force and os.path.exists(LIB):
        os.remove(LIB)
    subprocess.check_call(["make", "-C", HERE, "ref", "REF=" + REFERENCE], stdout=subprocess.DEVNULL)
    return LIB


def available() -> bool:
    return os.path.exists(LIB) or can_build()


def lib() -> C.CDLL:
    global _lib
    if _lib is None:
        if build() is None:
            raise FileNotFoundError("oracle/_ref/libse2lam_ref.so is absent and /root/reference is not here to build it from")
        _lib = C.CDLL(LIB)
    return _lib


def orb_extract(img, params=None, cap=8192):
    """se2lam::ORBextractor::operator() -> (keypoints structured array (n,), descriptors (n,32) u8)"""
    params = params or orb_params()
    img = np.ascontiguousarray(img, np.uint8)
    rows, cols = img.shape
    kps = np.zeros(cap, KP_DTYPE)
    desc = np.zeros((cap, 32), np.uint8)
    n = C.c_int(0)
    f = lib().ref_orb_extract
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_int)]
    rc = f(C.byref(params), img.ctypes.data, rows, cols, cols, kps.ctypes.data, desc.ctypes.data, cap, C.byref(n))
    if rc == -2:
        raise ValueError("the reference raises here (an OpenCV assertion: a cell rectangle outside its pyramid level)")
    assert rc == 0, "ref_orb_extract: capacity exceeded"
    return kps[:n.value].copy(), desc[:n.value].copy()


def hamming(a, b) -> int:
    a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
    f = lib().ref_hamming
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p]
    return int(f(a.ctypes.data, b.ctypes.data))


def three_maxima(counts):
    counts = np.ascontiguousarray(counts, np.int32)
    out = np.zeros(3, np.int32)
    f = lib().ref_three_maxima
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    f(counts.ctypes.data, len(counts), out.ctypes.data)
    return tuple(int(v) for v in out)


def features_in_area(kps, x, y, r, min_level, max_level, bounds=None):
    bounds = bounds or default_bounds()
    kps = np.ascontiguousarray(kps)
    out = np.zeros(max(len(kps), 1), np.int32)
    f = lib().ref_features_in_area
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_int, C.c_int, C.c_void_p, C.c_int]
    n = f(C.byref(bounds), kps.ctypes.data, len(kps), x, y, r, min_level, max_level, out.ctypes.data, out.size)
    return out[:n].copy()


def match_window(kps1, desc1, kps2, desc2, prev_xy=None, win=20, level_offset=1, min_level=0, max_level=8,
                 nnratio=0.9, bounds=None):
    """ORBmatcher::MatchByWindow -> (matches12 (n1,), nmatches, prev_xy updated)"""
    bounds = bounds or default_bounds()
    kps1 = np.ascontiguousarray(kps1); kps2 = np.ascontiguousarray(kps2)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    n1, n2 = len(kps1), len(kps2)
    if prev_xy is None:
        prev_xy = np.stack([kps1["x"], kps1["y"]], axis=1)
    prev = np.ascontiguousarray(prev_xy, np.float32).copy()
    m12 = np.full(max(n1, 1), -1, np.int32)
    f = lib().ref_match_window
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int,
                  C.c_int, C.c_int, C.c_float, C.c_void_p]
    nm = f(C.byref(bounds), kps1.ctypes.data, desc1.ctypes.data, n1, kps2.ctypes.data, desc2.ctypes.data, n2,
           prev.ctypes.data, win, level_offset, min_level, max_level, nnratio, m12.ctypes.data)
    return m12[:n1].copy(), int(nm), prev


def match_projection(mp_pos, mp_desc, mp_octave, mp_skip, Tcw, K4, kps, desc, kf_observed, win=15, level_offset=2,
                     nnratio=0.6, bounds=None):
    """ORBmatcher::MatchByProjection -> (match_idx_mp (n,), nmatches)"""
    bounds = bounds or default_bounds()
    mp_pos = np.ascontiguousarray(mp_pos, np.float32); mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
    mp_octave = np.ascontiguousarray(mp_octave, np.int32); mp_skip = np.ascontiguousarray(mp_skip, np.uint8)
    Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(-1)[:12].copy()
    kps = np.ascontiguousarray(kps); desc = np.ascontiguousarray(desc, np.uint8)
    kf_observed = np.ascontiguousarray(kf_observed, np.uint8)
    n, m = len(kps), len(mp_octave)
    out = np.full(max(n, 1), -1, np.int32)
    fx, fy, cx, cy = [float(v) for v in K4]
    f = lib().ref_match_projection
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_float, C.c_float, C.c_float,
                  C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p]
    nm = f(C.byref(bounds), mp_pos.ctypes.data, mp_desc.ctypes.data, mp_octave.ctypes.data, mp_skip.ctypes.data, m,
           Tcw.ctypes.data, fx, fy, cx, cy, kps.ctypes.data, desc.ctypes.data, kf_observed.ctypes.data, n, win, level_offset,
           nnratio, out.ctypes.data)
    return out[:n].copy(), int(nm)


def search_by_bow(kps1, desc1, fv1, has_mp1, kps2, desc2, fv2, has_mp2, mp_only=False, nnratio=0.6, check_ori=True):
    """ORBmatcher::SearchByBoW; fv = (nodes, ptr, idx) CSR int32 arrays -> (matches12 (n1,), nmatches)"""
    kps1 = np.ascontiguousarray(kps1); kps2 = np.ascontiguousarray(kps2)
    desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
    a = [np.ascontiguousarray(x, np.int32) for x in fv1]; b = [np.ascontiguousarray(x, np.int32) for x in fv2]
    h1 = np.ascontiguousarray(has_mp1, np.uint8); h2 = np.ascontiguousarray(has_mp2, np.uint8)
    out = np.full(max(len(kps1), 1), -1, np.int32)
    f = lib().ref_search_by_bow
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] * 2 + [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p] + [C.c_void_p] * 2 + \
        [C.c_int] + [C.c_void_p] * 3 + [C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_int, C.c_void_p]
    nm = f(kps1.ctypes.data, desc1.ctypes.data, len(kps1), a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data,
           len(a[0]), h1.ctypes.data, kps2.ctypes.data, desc2.ctypes.data, len(kps2), b[0].ctypes.data,
           b[1].ctypes.data, b[2].ctypes.data, len(b[0]), h2.ctypes.data, int(mp_only), nnratio, int(check_ori),
           out.ctypes.data)
    return out[:len(kps1)].copy(), int(nm)


def track_two_frames(img1, img2, params=None, K4=(400.0, 400.0, 320.0, 240.0), win=20, nnratio=0.9, cap=8192):
    """Frame::Frame on both images (undistort with D = 0, ORBextractor, grid), then MatchByWindow with the first frame's key
    points as vbPrevMatched -> ((kps1, desc1), (kps2, desc2), matches12, nmatches, prev_xy)"""
    params = params or orb_params()
    img1 = np.ascontiguousarray(img1, np.uint8); img2 = np.ascontiguousarray(img2, np.uint8)
    rows, cols = img1.shape
    k1 = np.zeros(cap, KP_DTYPE); k2 = np.zeros(cap, KP_DTYPE)
    d1 = np.zeros((cap, 32), np.uint8); d2 = np.zeros((cap, 32), np.uint8)
    n1 = C.c_int(0); n2 = C.c_int(0)
    m12 = np.full(cap, -1, np.int32)
    prev = np.zeros((cap, 2), np.float32)
    f = lib().ref_track_two_frames
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_float, C.c_float, C.c_float, C.c_int,
                  C.c_float, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_int,
                  C.c_void_p, C.c_void_p]
    nm = f(C.byref(params), img1.ctypes.data, img2.ctypes.data, rows, cols, *[float(v) for v in K4], win, nnratio,
           k1.ctypes.data, d1.ctypes.data, C.byref(n1), k2.ctypes.data, d2.ctypes.data, C.byref(n2), cap, m12.ctypes.data,
           prev.ctypes.data)
    assert nm >= 0, "ref_track_two_frames: capacity exceeded"
    a, b = n1.value, n2.value
    return (k1[:a].copy(), d1[:a].copy()), (k2[:b].copy(), d2[:b].copy()), m12[:a].copy(), int(nm), prev[:a].copy()


def triangulate_point(pt1, pt2, P1, P2):
    out = np.zeros(3, np.float32)
    a = np.ascontiguousarray(pt1, np.float32); b = np.ascontiguousarray(pt2, np.float32)
    p1 = np.ascontiguousarray(P1, np.float32).reshape(-1); p2 = np.ascontiguousarray(P2, np.float32).reshape(-1)
    f = lib().ref_triangulate_point
    f.restype = None
    f.argtypes = [C.c_void_p] * 5
    f(a.ctypes.data, b.ctypes.data, p1.ctypes.data, p2.ctypes.data, out.ctypes.data)
    return out


def check_parallax(o1, o2, p, min_degree=1) -> bool:
    a, b, c = [np.ascontiguousarray(v, np.float32) for v in (o1, o2, p)]
    f = lib().ref_check_parallax
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] * 3 + [C.c_int]
    return bool(f(a.ctypes.data, b.ctypes.data, c.ctypes.data, min_degree))


def se2_compose(a, b, minus=False):
    a = np.ascontiguousarray(a, np.float32); b = np.ascontiguousarray(b, np.float32)
    out = np.zeros(3, np.float32)
    f = lib().ref_se2_compose
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
    f(a.ctypes.data, b.ctypes.data, int(minus), out.ctypes.data)
    return out


def edge_se2xyz(g, pose, lw, uv):
    """g2o::EdgeSE2XYZ::computeError / linearizeOplus on one edge of graph g's camera and extrinsic -> (e (2,), Jp (2,3), Jl (2,3))"""
    D = C.POINTER(C.c_double)
    f = lib().ref_edge_se2xyz
    f.restype = None
    f.argtypes = [C.c_double] * 3 + [C.c_void_p] * 8
    R = np.ascontiguousarray(g.Rbc, np.float64); t = np.ascontiguousarray(g.tbc, np.float64)
    pose = np.ascontiguousarray(pose, np.float64); lw = np.ascontiguousarray(lw, np.float64); uv = np.ascontiguousarray(uv, np.float64)
    e = np.zeros(2); Jp = np.zeros((2, 3)); Jl = np.zeros((2, 3))
    f(float(g.fx), float(g.cx), float(g.cy), R.ctypes.data, t.ctypes.data, pose.ctypes.data, lw.ctypes.data, uv.ctypes.data,
      e.ctypes.data, Jp.ctypes.data, Jl.ctypes.data)
    return e, Jp, Jl


def edge_pre_se2(pi, pj, z):
    """g2o::PreEdgeSE2 -> (e (3,), Ji (3,3), Jj (3,3))"""
    pi = np.ascontiguousarray(pi, np.float64); pj = np.ascontiguousarray(pj, np.float64); z = np.ascontiguousarray(z, np.float64)
    e = np.zeros(3); Ji = np.zeros((3, 3)); Jj = np.zeros((3, 3))
    f = lib().ref_edge_pre_se2
    f.restype = None
    f.argtypes = [C.c_void_p] * 6
    f(pi.ctypes.data, pj.ctypes.data, z.ctypes.data, e.ctypes.data, Ji.ctypes.data, Jj.ctypes.data)
    return e, Ji, Jj


def se2_se3_round_trip(pose):
    pose = np.ascontiguousarray(pose, np.float64)
    out = np.zeros(3); d = np.zeros((3, 3))
    f = lib().ref_se2_se3_round_trip
    f.restype = None
    f.argtypes = [C.c_void_p] * 3
    f(pose.ctypes.data, out.ctypes.data, d.ctypes.data)
    return out, d


# ---- src/optimizer.cpp + src/converter.cpp (whole files) against the recording graph of oracle/_shim/g2o_shim.hpp
def _pose12(T):
    T = np.asarray(T, np.float64)
    return np.ascontiguousarray(np.concatenate([T[:3, :3].reshape(-1), T[:3, 3]]))


def _mat44(p12):
    T = np.eye(4)
    T[:3, :3] = p12[:9].reshape(3, 3); T[:3, 3] = p12[9:]
    return T


def _plane(fname, T, Tbc, xrot, yrot, z):
    f = getattr(lib(), fname)
    f.restype = C.c_int
    f.argtypes = [C.c_void_p, C.c_void_p, C.c_double, C.c_double, C.c_double, C.c_void_p, C.c_void_p]
    a, b = _pose12(T), _pose12(Tbc)
    meas = np.zeros(12); info = np.zeros(36)
    rc = f(a.ctypes.data, b.ctypes.data, xrot, yrot, z, meas.ctypes.data, info.ctypes.data)
    return _mat44(meas), info.reshape(6, 6), rc


def plane_motion_prior(Tcw, Tbc, xrot_info=1e6, yrot_info=1e6, z_info=1.0):
    """addVertexSE3Expmap + addPlaneMotionSE3Expmap (optimizer.cpp:226-314) -> (measurement 4x4, information 6x6 in
    (rotation, translation) order, number of edges the graph then holds).  Tbc is rounded to float32 like Config::bTc."""
    return _plane("ref_plane_motion_expmap", Tcw, Tbc, xrot_info, yrot_info, z_info)


def pg_plane_motion_prior(Twc, Tbc, xrot_info=1e6, yrot_info=1e6, z_info=1.0):
    """addVertexSE3PlaneMotion (optimizer.cpp:336-470) -> (measurement 4x4, information 6x6 in (translation, rotation)
    order, the SE3-offset parameter id handed to the prior edge)"""
    return _plane("ref_plane_motion_iso3", Twc, Tbc, xrot_info, yrot_info, z_info)


def prior_expmap_edge(meas, est):
    """EdgeSE3ExpmapPrior::computeError / linearizeOplus -> (err (6,), J (6,6))"""
    a, b = _pose12(meas), _pose12(est)
    e = np.zeros(6); J = np.zeros(36)
    f = lib().ref_prior_expmap_edge
    f.restype = None
    f.argtypes = [C.c_void_p] * 4
    f(a.ctypes.data, b.ctypes.data, e.ctypes.data, J.ctypes.data)
    return e, J.reshape(6, 6)


def edge_se3expmap_info(info_tr):
    """addEdgeSE3Expmap's information as the edge holds it ((rotation, translation) order); None when verifyInfo rejects it"""
    a = np.ascontiguousarray(info_tr, np.float64); out = np.zeros(36)
    f = lib().ref_edge_se3expmap_info
    f.restype = C.c_int
    f.argtypes = [C.c_void_p] * 2
    return out.reshape(6, 6) if f(a.ctypes.data, out.ctypes.data) == 0 else None


def so3_jacobians(v):
    v = np.ascontiguousarray(v, np.float64); a = np.zeros(9); b = np.zeros(9)
    f = lib().ref_so3_jacobians
    f.restype = None
    f.argtypes = [C.c_void_p] * 3
    f(v.ctypes.data, a.ctypes.data, b.ctypes.data)
    return a.reshape(3, 3), b.reshape(3, 3)


def inv_jjl(v6):
    v = np.ascontiguousarray(v6, np.float64); out = np.zeros(36)
    f = lib().ref_inv_jjl
    f.restype = None
    f.argtypes = [C.c_void_p] * 2
    f(v.ctypes.data, out.ctypes.data)
    return out.reshape(6, 6)


def converter_round_trip(T):
    a = _pose12(T); q = np.zeros(16, np.float32); i = np.zeros(16, np.float32)
    f = lib().ref_converter_round_trip
    f.restype = None
    f.argtypes = [C.c_void_p] * 3
    f(a.ctypes.data, q.ctypes.data, i.ctypes.data)
    return q.reshape(4, 4), i.reshape(4, 4)


def window_chi2(g, K=None):
    """synth.BAGraph built with addCamPara / addVertexSE2 / addEdgeSE2 / addVertexSBAXYZ / addEdgeSE2XYZ, every edge's own
    computeError() -> (sum rho(chi2), chi2 per observation (E,), chi2 per odometry edge (O,), (edges, fixed, marginalised))"""
    def arr(a, ty):
        return np.ascontiguousarray(a, ty)
    poses = arr(g.poses, np.float64); fixed = arr(g.fixed, np.uint8); lms = arr(g.lms, np.float64)
    e_kf = arr(g.e_kf, np.int32); e_lm = arr(g.e_lm, np.int32); e_uv = arr(g.e_uv, np.float64); e_info = arr(g.e_info, np.float64)
    o_i = arr(g.o_i, np.int32); o_j = arr(g.o_j, np.int32); o_meas = arr(g.o_meas, np.float64); o_info = arr(g.o_info, np.float64)
    K = np.ascontiguousarray([g.fx, 0, g.cx, 0, g.fx, g.cy, 0, 0, 1] if K is None else np.asarray(K).reshape(-1), np.float32)
    Tbc = np.eye(4); Tbc[:3, :3] = g.Rbc; Tbc[:3, 3] = g.tbc
    tb = _pose12(Tbc)
    chi_e = np.zeros(max(g.E, 1)); chi_o = np.zeros(max(g.O, 1)); counts = np.zeros(3, np.int32)
    f = lib().ref_window_chi2
    f.restype = C.c_double
    VP = C.c_void_p
    f.argtypes = [C.c_int, VP, VP, C.c_int, VP, C.c_int, VP, VP, VP, VP, C.c_int, VP, VP, VP, VP, VP, VP, C.c_double, VP, VP, VP]
    total = f(g.P, poses.ctypes.data, fixed.ctypes.data, g.L, lms.ctypes.data, g.E, e_kf.ctypes.data, e_lm.ctypes.data,
              e_uv.ctypes.data, e_info.ctypes.data, g.O, o_i.ctypes.data, o_j.ctypes.data, o_meas.ctypes.data, o_info.ctypes.data,
              K.ctypes.data, tb.ctypes.data, float(g.huber), chi_e.ctypes.data, chi_o.ctypes.data, counts.ctypes.data)
    return float(total), chi_e[:g.E], chi_o[:g.O], tuple(int(c) for c in counts)


# ---- src/sparsifier.cpp (whole file)
def sparsify(kf, mp, m_kf, m_mp, m_info):
    """Sparsifier::DoMarginalizeSE3XYZ -> (z_out 4x4 = KF0^-1 KF1, info_out 6x6); same arguments as oracle.sparsify"""
    k = np.ascontiguousarray(np.concatenate([_pose12(kf[0]), _pose12(kf[1])]))
    mp = np.ascontiguousarray(mp, np.float64); mk = np.ascontiguousarray(m_kf, np.int32); mm = np.ascontiguousarray(m_mp, np.int32)
    mi = np.ascontiguousarray(m_info, np.float64).reshape(-1, 9)
    z = np.zeros(12); info = np.zeros(36)
    f = lib().ref_sparsify
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    f(k.ctypes.data, len(mp), mp.ctypes.data, len(mk), mk.ctypes.data, mm.ctypes.data, mi.ctypes.data, z.ctypes.data, info.ctypes.data)
    return _mat44(z), info.reshape(6, 6)


def sparsify_hessian(kf, mp, info):
    """Sparsifier::JacobianSE3XYZ / HessianSE3XYZ of one measurement -> (J 3x9, H 9x9)"""
    k = _pose12(kf); p = np.ascontiguousarray(mp, np.float64); w = np.ascontiguousarray(info, np.float64).reshape(-1)
    J = np.zeros(27); H = np.zeros(81)
    f = lib().ref_sparsify_hessian
    f.restype = None
    f.argtypes = [C.c_void_p] * 5
    f(k.ctypes.data, p.ctypes.data, w.ctypes.data, J.ctypes.data, H.ctypes.data)
    return J.reshape(3, 9), H.reshape(9, 9)


def sparsify_info_se3(kf, H):
    """Sparsifier::InfoSE3 on a given 12x12 marginal Hessian -> 6x6"""
    k = np.ascontiguousarray(np.concatenate([_pose12(kf[0]), _pose12(kf[1])]))
    Hm = np.ascontiguousarray(H, np.float64).reshape(-1); out = np.zeros(36)
    f = lib().ref_sparsify_info_se3
    f.restype = None
    f.argtypes = [C.c_void_p] * 3
    f(k.ctypes.data, Hm.ctypes.data, out.ctypes.data)
    return out.reshape(6, 6)


# ---- the MAP build (libse2lam_ref_map.so): src/Map.cpp, src/KeyFrame.cpp, src/MapPoint.cpp with the reference's own headers
LIB_MAP = os.path.join(HERE, "_ref", "libse2lam_ref_map.so")
_map_lib = None


def lib_map() -> C.CDLL:
    global _map_lib
    if _map_lib is None:
        build()
        _map_lib = C.CDLL(LIB_MAP)
    return _map_lib


class RefMap:
    """A se2lam::Map filled through the reference's own KeyFrame / MapPoint / Map calls (oracle/ref_map_driver.cpp)."""

    def __init__(self, K, bTc, huber, xrot_info=1e6, yrot_info=1e6, z_info=1.0, max_level=8, scale_factor=1.2):
        l = lib_map()
        l.ref_map_create.restype = C.c_void_p
        l.ref_map_create.argtypes = [C.c_void_p, C.c_void_p] + [C.c_float] * 4 + [C.c_int, C.c_float]
        k = np.ascontiguousarray(K, np.float32).reshape(-1); t = np.ascontiguousarray(bTc, np.float32).reshape(-1)
        self._h = C.c_void_p(l.ref_map_create(k.ctypes.data, t.ctypes.data, huber, xrot_info, yrot_info, z_info, max_level, scale_factor))
        self._l = l

    def __del__(self):
        if getattr(self, "_h", None):
            self._l.ref_map_destroy.argtypes = [C.c_void_p]
            self._l.ref_map_destroy(self._h)
            self._h = None

    # ---- MapStorage.cpp (oracle/ref_storage_driver.cpp): files travel as node lines, see oracle/_shim/cv_shim.hpp
    def _storage_check(self, r):
        if r < 0:
            self._l.ref_storage_error.restype = C.c_char_p
            raise RuntimeError("reference MapStorage: " + self._l.ref_storage_error().decode())
        return r

    def storage_load(self, events: str) -> int:
        """MapStorage::loadMap() of the file given as node lines, into this map; returns the number of key frames."""
        f = self._l.ref_storage_load
        f.restype = C.c_long
        f.argtypes = [C.c_void_p, C.c_char_p]
        return self._storage_check(f(self._h, events.encode()))

    def storage_save(self) -> str:
        """MapStorage::saveMap() of this map; the file it wrote as node lines."""
        f = self._l.ref_storage_save
        f.restype = C.c_long
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_long]
        n = self._storage_check(f(self._h, None, 0))
        buf = C.create_string_buffer(n + 1)
        self._storage_check(f(self._h, buf, n + 1))
        return buf.value.decode()

    def save_trajectory(self, bTc, directory: str, frame_ids=None) -> str:
        """OdoSLAM::saveMap()'s key-frame trajectory (src/OdoSLAM.cpp:198-212, compiled) of this map; returns the text it wrote.
        frame_ids: KeyFrame::id per key frame (the number each line starts with; the map file does not store it)."""
        if frame_ids is not None:
            f = self._l.ref_system_kf_set_id
            f.restype = None
            f.argtypes = [C.c_void_p, C.c_int, C.c_int]
            for k, i in enumerate(frame_ids):
                f(self._h, k, int(i))
        t = np.ascontiguousarray(bTc, np.float32).reshape(-1)
        g = self._l.ref_system_save_trajectory
        g.restype = C.c_int
        g.argtypes = [C.c_void_p, C.c_void_p, C.c_char_p]
        if g(self._h, t.ctypes.data, directory.encode()) != 0:
            raise RuntimeError("reference OdoSLAM::saveMap raised")
        return open(os.path.join(directory, "se2lam_kf_trajectory.txt")).read()

    def storage_counts(self):
        out = np.zeros(2, np.int32)
        self._l.ref_storage_counts.argtypes = [C.c_void_p, C.c_void_p]
        self._l.ref_storage_counts(self._h, out.ctypes.data)
        return int(out[0]), int(out[1])

    def storage_kf_state(self, kf):
        out = np.zeros(12, np.int32)
        self._l.ref_storage_kf_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._l.ref_storage_kf_state(self._h, int(kf), out.ctypes.data)
        names = ("kps", "kps_un", "desc_rows", "view_mps", "view_infos", "obs", "covisible", "odo_from", "odo_to", "ftr_from", "ftr_to", "img_rows")
        return dict(zip(names, (int(v) for v in out)))

    def storage_mp_state(self, mp):
        out = np.zeros(4, np.int32)
        self._l.ref_storage_mp_state.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self._l.ref_storage_mp_state(self._h, int(mp), out.ctypes.data)
        return dict(obs=int(out[0]), good_prl=bool(out[1]), null=bool(out[2]), id=int(out[3]))

    def storage_mp_ftr_idx(self, mp, kf) -> int:
        f = self._l.ref_storage_mp_ftr_idx
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_int]
        return f(self._h, int(mp), int(kf))

    def storage_mp_set_good_prl(self, mp, good: bool):
        f = self._l.ref_storage_mp_set_good_prl
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int, C.c_int]
        f(self._h, int(mp), 1 if good else 0)

    def add_kf(self, id_kf, frame_id, twb, kp_xy=None, kp_octave=None, view_lc=None) -> int:
        xy = np.ascontiguousarray(np.zeros((0, 2)) if kp_xy is None else kp_xy, np.float32).reshape(-1, 2)
        n = len(xy)
        octv = np.ascontiguousarray(np.zeros(n) if kp_octave is None else kp_octave, np.int32)
        lc = np.ascontiguousarray(np.tile([0.0, 0.0, 1000.0], (n, 1)) if view_lc is None else view_lc, np.float32).reshape(-1, 3)
        t = np.ascontiguousarray(twb, np.float32)
        f = self._l.ref_map_add_kf
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p]
        return f(self._h, int(id_kf), int(frame_id), t.ctypes.data, n, xy.ctypes.data, octv.ctypes.data, lc.ctypes.data)

    def add_mp(self, id_mp, pos) -> int:
        p = np.ascontiguousarray(pos, np.float32)
        f = self._l.ref_map_add_mp
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        return f(self._h, int(id_mp), p.ctypes.data)

    def observe(self, kf, mp, ftr):
        f = self._l.ref_map_observe
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_int]
        f(self._h, int(kf), int(mp), int(ftr))

    def covisible(self, a, b):
        f = self._l.ref_map_covisible
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int, C.c_int]
        f(self._h, int(a), int(b))

    def set_odo(self, kf_from, kf_to, meas, cov):
        m = np.ascontiguousarray(meas, np.float64); c = np.ascontiguousarray(cov, np.float64).reshape(-1)
        f = self._l.ref_map_set_odo
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        f(self._h, int(kf_from), int(kf_to), m.ctypes.data, c.ctypes.data)

    def kf_pose(self, kf) -> np.ndarray:
        out = np.zeros(16, np.float32)
        f = self._l.ref_map_kf_pose
        f.restype = None
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        f(self._h, int(kf), out.ctypes.data)
        return out.reshape(4, 4)

    def update_local_graph(self, current_kf, cap_kf=4096, cap_mp=1 << 18):
        """Map::setCurrentKF + Map::updateLocalGraph -> (mIdKF of the local key frames, of the reference key frames, mId of the local map points)"""
        lk = np.zeros(cap_kf, np.int32); rk = np.zeros(cap_kf, np.int32); lm = np.zeros(cap_mp, np.int32); cnt = np.zeros(3, np.int32)
        f = self._l.ref_map_update_local_graph
        f.restype = C.c_int
        f.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p]
        rc = f(self._h, int(current_kf), lk.ctypes.data, rk.ctypes.data, lm.ctypes.data, cap_kf, cap_mp, cnt.ctypes.data)
        assert rc == 0, cnt
        return lk[:cnt[0]].copy(), rk[:cnt[1]].copy(), lm[:cnt[2]].copy()

    def load_local_graph(self, cap_v=1 << 18, cap_o=4096, cap_e=1 << 20):
        """Map::loadLocalGraph(SlamOptimizer&) on the last local graph -> what the recording optimizer holds"""
        v_id = np.zeros(cap_v, np.int32); v_kind = np.zeros(cap_v, np.int32); v_est = np.zeros((cap_v, 3)); v_flags = np.zeros(cap_v, np.uint8)
        o_ids = np.zeros((cap_o, 2), np.int32); o_meas = np.zeros((cap_o, 3)); o_info = np.zeros((cap_o, 9)); o_chi2 = np.zeros(cap_o)
        e_ids = np.zeros((cap_e, 2), np.int32); e_uv = np.zeros((cap_e, 2)); e_info = np.zeros((cap_e, 4)); e_delta = np.zeros(cap_e); e_chi2 = np.zeros(cap_e)
        cnt = np.zeros(3, np.int32)
        f = self._l.ref_map_load_local_graph
        f.restype = C.c_double
        VP = C.c_void_p
        f.argtypes = [VP, C.c_int, VP, VP, VP, VP, C.c_int, VP, VP, VP, VP, C.c_int, VP, VP, VP, VP, VP, VP]
        total = f(self._h, cap_v, v_id.ctypes.data, v_kind.ctypes.data, v_est.ctypes.data, v_flags.ctypes.data, cap_o, o_ids.ctypes.data,
                  o_meas.ctypes.data, o_info.ctypes.data, o_chi2.ctypes.data, cap_e, e_ids.ctypes.data, e_uv.ctypes.data, e_info.ctypes.data,
                  e_delta.ctypes.data, e_chi2.ctypes.data, cnt.ctypes.data)
        assert total >= 0, cnt
        nv, no, ne = (int(c) for c in cnt)
        return dict(chi2=float(total), v_id=v_id[:nv], v_kind=v_kind[:nv], v_est=v_est[:nv], v_fixed=(v_flags[:nv] & 1).astype(bool),
                    v_marginalized=(v_flags[:nv] & 2).astype(bool), o_ids=o_ids[:no], o_meas=o_meas[:no], o_info=o_info[:no].reshape(-1, 3, 3),
                    o_chi2=o_chi2[:no], e_ids=e_ids[:ne], e_uv=e_uv[:ne], e_info=e_info[:ne].reshape(-1, 2, 2), e_delta=e_delta[:ne],
                    e_chi2=e_chi2[:ne])


def _refmap_set_odo_se3(self, kf, to, measure, info):
    """KeyFrame::setOdoMeasureFrom(to, measure 4x4, info 6x6 in (translation, rotation) order) - both CV_32F in the reference"""
    m = np.ascontiguousarray(measure, np.float32).reshape(-1); w = np.ascontiguousarray(info, np.float32).reshape(-1)
    f = self._l.ref_map_set_odo_se3
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    f(self._h, int(kf), int(to), m.ctypes.data, w.ctypes.data)


def _refmap_load_local_graph_se3(self, cap_v=1 << 18, cap_p=4096, cap_o=4096, cap_e=1 << 20):
    """Map::loadLocalGraph(optimizer, vpEdgesAll, vnAllIdx) (the SE3-expmap local graph) -> what the recording optimizer holds"""
    v_id = np.zeros(cap_v, np.int32); v_kind = np.zeros(cap_v, np.int32); v_est = np.zeros((cap_v, 12)); v_flags = np.zeros(cap_v, np.uint8)
    p_id = np.zeros(cap_p, np.int32); p_meas = np.zeros((cap_p, 12)); p_info = np.zeros((cap_p, 36)); p_chi2 = np.zeros(cap_p)
    o_ids = np.zeros((cap_o, 2), np.int32); o_meas = np.zeros((cap_o, 12)); o_info = np.zeros((cap_o, 36)); o_chi2 = np.zeros(cap_o)
    e_ids = np.zeros((cap_e, 2), np.int32); e_uv = np.zeros((cap_e, 2)); e_info = np.zeros((cap_e, 4)); e_delta = np.zeros(cap_e)
    e_level = np.zeros(cap_e, np.int32); e_chi2 = np.zeros(cap_e)
    cnt = np.zeros(5, np.int32)
    f = self._l.ref_map_load_local_graph_se3
    f.restype = C.c_double
    VP = C.c_void_p
    f.argtypes = [VP, C.c_int, VP, VP, VP, VP, C.c_int, VP, VP, VP, VP, C.c_int, VP, VP, VP, VP, C.c_int, VP, VP, VP, VP, VP, VP, VP]
    total = f(self._h, cap_v, v_id.ctypes.data, v_kind.ctypes.data, v_est.ctypes.data, v_flags.ctypes.data, cap_p, p_id.ctypes.data,
              p_meas.ctypes.data, p_info.ctypes.data, p_chi2.ctypes.data, cap_o, o_ids.ctypes.data, o_meas.ctypes.data, o_info.ctypes.data,
              o_chi2.ctypes.data, cap_e, e_ids.ctypes.data, e_uv.ctypes.data, e_info.ctypes.data, e_delta.ctypes.data, e_level.ctypes.data,
              e_chi2.ctypes.data, cnt.ctypes.data)
    assert total >= 0, cnt
    nv, npri, no, ne, nall = (int(c) for c in cnt)

    def T(a):
        out = np.tile(np.eye(4), (len(a), 1, 1))
        out[:, :3, :3] = a[:, :9].reshape(-1, 3, 3); out[:, :3, 3] = a[:, 9:]
        return out
    return dict(chi2=float(total), v_id=v_id[:nv], v_kind=v_kind[:nv], v_est=v_est[:nv], v_fixed=(v_flags[:nv] & 1).astype(bool),
                v_marginalized=(v_flags[:nv] & 2).astype(bool), p_id=p_id[:npri], p_meas=T(p_meas[:npri]), p_info=p_info[:npri].reshape(-1, 6, 6),
                p_chi2=p_chi2[:npri], o_ids=o_ids[:no], o_meas=T(o_meas[:no]), o_info=o_info[:no].reshape(-1, 6, 6), o_chi2=o_chi2[:no],
                e_ids=e_ids[:ne], e_uv=e_uv[:ne], e_info=e_info[:ne].reshape(-1, 2, 2), e_delta=e_delta[:ne], e_level=e_level[:ne],
                e_chi2=e_chi2[:ne], edges_returned=nall)


RefMap.set_odo_se3 = _refmap_set_odo_se3
RefMap.load_local_graph_se3 = _refmap_load_local_graph_se3


# ---- the threads (src/Track.cpp, src/Localizer.cpp ...; oracle/ref_threads_driver.cpp)
def track_triangulate(K, kps_ref, kps_cur, match_idx, has_obs, view_mp, Tcr, lower, upper, frame_gap=10):
    """Track::doTriangulate -> (pos (n,3) f32 = mLocalMPs (untouched entries stay (-1,-1,-1)), good (n,) u8, match_idx updated, n_good, n_tracked_old)"""
    l = lib_map()
    k = np.ascontiguousarray(K, np.float32).reshape(-1)
    k1 = np.ascontiguousarray(kps_ref); k2 = np.ascontiguousarray(kps_cur)
    n = len(k1)
    m = np.ascontiguousarray(match_idx, np.int32).copy()
    ho = np.ascontiguousarray(np.zeros(n) if has_obs is None else has_obs, np.uint8)
    vm = np.ascontiguousarray(np.zeros((n, 3)) if view_mp is None else view_mp, np.float32)
    T = np.ascontiguousarray(Tcr, np.float32).reshape(-1)
    pos = np.zeros((max(n, 1), 3), np.float32); good = np.zeros(max(n, 1), np.uint8); ng = np.zeros(1, np.int32)
    f = l.ref_track_triangulate
    f.restype = C.c_int
    VP = C.c_void_p
    f.argtypes = [VP, C.c_float, C.c_float, C.c_int, VP, VP, VP, C.c_int, VP, VP, VP, C.c_int, VP, VP, VP]
    nold = f(k.ctypes.data, lower, upper, n, k1.ctypes.data, ho.ctypes.data, vm.ctypes.data, len(k2), k2.ctypes.data, m.ctypes.data, T.ctypes.data,
             int(frame_gap), pos.ctypes.data, good.ctypes.data, ng.ctypes.data)
    return pos[:n], good[:n], m, int(ng[0]), int(nold)


def track_update_frame_pose(bTc, noise, kf_odom, kf_twb, last_odom, odom, meas, cov):
    """Track::updateFramePose -> dict(meas (3,), cov (9,) as the array holds it, Trb, Twb (3,) f32, Tcr, Tcw (4,4) f32)"""
    l = lib_map()
    b = np.ascontiguousarray(bTc, np.float32).reshape(-1); nz = np.ascontiguousarray(noise, np.float32)
    a = [np.ascontiguousarray(x, np.float32) for x in (kf_odom, kf_twb, last_odom, odom)]
    m = np.ascontiguousarray(meas, np.float64).copy(); c = np.ascontiguousarray(cov, np.float64).reshape(-1).copy()
    trb = np.zeros(3, np.float32); twb = np.zeros(3, np.float32); Tcr = np.zeros(16, np.float32); Tcw = np.zeros(16, np.float32)
    f = l.ref_track_update_frame_pose
    f.restype = None
    f.argtypes = [C.c_void_p] * 12
    f(b.ctypes.data, nz.ctypes.data, a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, a[3].ctypes.data, m.ctypes.data, c.ctypes.data,
      trb.ctypes.data, twb.ctypes.data, Tcr.ctypes.data, Tcw.ctypes.data)
    return dict(meas=m, cov=c, Trb=trb, Twb=twb, Tcr=Tcr.reshape(4, 4), Tcw=Tcw.reshape(4, 4))


def localizer_do_local_ba(K, bTc, huber, Tcw, kps, mp_pos, mp_good, id_kf=7, xrot_info=1e6, yrot_info=1e6, z_info=1.0):
    """Localizer::DoLocalBA up to optimizer.optimize(30): the graph the reference built and its cost at the start"""
    l = lib_map()
    k = np.ascontiguousarray(K, np.float32).reshape(-1); b = np.ascontiguousarray(bTc, np.float32).reshape(-1)
    T = np.ascontiguousarray(Tcw, np.float32).reshape(-1)
    kp = np.ascontiguousarray(kps); n = len(kp)
    pos = np.ascontiguousarray(mp_pos, np.float32).reshape(-1, 3); gd = np.ascontiguousarray(mp_good, np.uint8)
    out5 = np.zeros(5, np.int32); pm = np.zeros(12); pi = np.zeros(36)
    e_point = np.zeros(max(n, 1), np.int32); e_uv = np.zeros((max(n, 1), 2)); e_w = np.zeros(max(n, 1)); e_chi2 = np.zeros(max(n, 1)); e_delta = np.zeros(max(n, 1))
    f = l.ref_localizer_do_local_ba
    f.restype = C.c_double
    VP = C.c_void_p
    f.argtypes = [VP, VP] + [C.c_float] * 4 + [C.c_int, VP, C.c_int, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP, VP]
    total = f(k.ctypes.data, b.ctypes.data, huber, xrot_info, yrot_info, z_info, int(id_kf), T.ctypes.data, n, kp.ctypes.data, pos.ctypes.data,
              gd.ctypes.data, out5.ctypes.data, pm.ctypes.data, pi.ctypes.data, e_point.ctypes.data, e_uv.ctypes.data, e_w.ctypes.data,
              e_chi2.ctypes.data, e_delta.ctypes.data)
    ne = int(out5[2])
    return dict(chi2=float(total), n_vertices=int(out5[0]), n_fixed=int(out5[1]), n_edges=ne, n_priors=int(out5[3]), iterations=int(out5[4]),
                prior_meas=_mat44(pm), prior_info=pi.reshape(6, 6), e_point=e_point[:ne], e_uv=e_uv[:ne], e_w=e_w[:ne], e_chi2=e_chi2[:ne],
                e_delta=e_delta[:ne])


def _refmap_add_ftr_measure(self, kf_from, kf_to, measure, info):
    """KeyFrame::addFtrMeasureFrom / To: the feature constraint (4x4 T, 6x6 information in (translation, rotation) order, CV_32F)"""
    m = np.ascontiguousarray(measure, np.float32).reshape(-1); w = np.ascontiguousarray(info, np.float32).reshape(-1)
    f = self._l.ref_map_add_ftr_measure
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
    f(self._h, int(kf_from), int(kf_to), m.ctypes.data, w.ctypes.data)


def _refmap_mp_pos(self, mp):
    out = np.zeros(3, np.float32)
    f = self._l.ref_map_mp_pos
    f.restype = None
    f.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
    f(self._h, int(mp), out.ctypes.data)
    return out


def _refmap_global_ba(self, global_iter=15, cap_v=4096, cap_e=1 << 16):
    """GlobalMapper::GlobalBA (GlobalMapper.cpp:328-535) up to optimize(): the pose graph the reference built and its cost at the start"""
    v_id = np.zeros(cap_v, np.int32); v_est = np.zeros((cap_v, 12)); v_fixed = np.zeros(cap_v, np.uint8)
    p_id = np.zeros(cap_v, np.int32); p_meas = np.zeros((cap_v, 12)); p_info = np.zeros((cap_v, 36)); p_chi2 = np.zeros(cap_v)
    e_ids = np.zeros((cap_e, 2), np.int32); e_meas = np.zeros((cap_e, 12)); e_info = np.zeros((cap_e, 36)); e_chi2 = np.zeros(cap_e)
    cnt = np.zeros(4, np.int32)
    f = self._l.ref_global_ba
    f.restype = C.c_double
    VP = C.c_void_p
    f.argtypes = [VP, C.c_int, C.c_int, VP, VP, VP, C.c_int, VP, VP, VP, VP, C.c_int, VP, VP, VP, VP, VP]
    total = f(self._h, int(global_iter), cap_v, v_id.ctypes.data, v_est.ctypes.data, v_fixed.ctypes.data, cap_v, p_id.ctypes.data, p_meas.ctypes.data,
              p_info.ctypes.data, p_chi2.ctypes.data, cap_e, e_ids.ctypes.data, e_meas.ctypes.data, e_info.ctypes.data, e_chi2.ctypes.data,
              cnt.ctypes.data)
    assert total >= 0, cnt
    nv, npri, ne, it = (int(c) for c in cnt)

    def T(a):
        out = np.tile(np.eye(4), (len(a), 1, 1))
        out[:, :3, :3] = a[:, :9].reshape(-1, 3, 3); out[:, :3, 3] = a[:, 9:]
        return out
    return dict(chi2=float(total), iterations=it, v_id=v_id[:nv], v_est=T(v_est[:nv]), v_fixed=v_fixed[:nv].astype(bool), p_id=p_id[:npri],
                p_meas=T(p_meas[:npri]), p_info=p_info[:npri].reshape(-1, 6, 6), p_chi2=p_chi2[:npri], e_ids=e_ids[:ne], e_meas=T(e_meas[:ne]),
                e_info=e_info[:ne].reshape(-1, 6, 6), e_chi2=e_chi2[:ne])


RefMap.add_ftr_measure = _refmap_add_ftr_measure
RefMap.mp_pos = _refmap_mp_pos
RefMap.global_ba = _refmap_global_ba


# ---- the vendored DBoW2 vocabulary (Thirdparty/DBoW2, compiled from the reference's tree; oracle/ref_voc_driver.cpp)
class RefVocabulary:
    """se2lam::ORBVocabulary = DBoW2::TemplatedVocabulary<FORB::TDescriptor, FORB>, loaded with loadFromBinaryFile"""

    def __init__(self, path):
        self._l = lib_map()
        info = np.zeros(6, np.int32)
        f = self._l.ref_voc_load
        f.restype = C.c_void_p
        f.argtypes = [C.c_char_p, C.c_void_p]
        self._h = C.c_void_p(f(str(path).encode(), info.ctypes.data))
        self.k, self.L, self.words, self.scoring, self.weighting = (int(x) for x in info[:5])
        self.loaded = bool(info[5])

    def __del__(self):
        if getattr(self, "_h", None):
            self._l.ref_voc_free.argtypes = [C.c_void_p]
            self._l.ref_voc_free(self._h)
            self._h = None

    def transform(self, desc, levelsup):
        """-> (word ids, word values, {node id: [feature indices]})"""
        d = np.ascontiguousarray(desc, np.uint8).reshape(-1, 32)
        n = len(d)
        bid = np.zeros(max(n, 1), np.uint32); bval = np.zeros(max(n, 1)); nb = np.zeros(1, np.int32)
        node = np.zeros(max(n, 1), np.int32); ptr = np.zeros(max(n, 1) + 1, np.int32); idx = np.zeros(max(n, 1), np.int32); nn = np.zeros(1, np.int32)
        f = self._l.ref_voc_transform
        f.restype = C.c_int
        VP = C.c_void_p
        f.argtypes = [VP, VP, C.c_int, C.c_int, C.c_int, VP, VP, VP, C.c_int, VP, VP, C.c_int, VP, VP]
        rc = f(self._h, d.ctypes.data, n, int(levelsup), len(bid), bid.ctypes.data, bval.ctypes.data, nb.ctypes.data, len(node), node.ctypes.data,
               ptr.ctypes.data, len(idx), idx.ctypes.data, nn.ctypes.data)
        assert rc == 0
        k, m = int(nb[0]), int(nn[0])
        fv = {int(node[i]): idx[ptr[i]:ptr[i + 1]].tolist() for i in range(m)}
        return bid[:k].tolist(), bval[:k].copy(), fv

    def score(self, a, b):
        ia = np.ascontiguousarray(a[0], np.uint32); va = np.ascontiguousarray(a[1], np.float64)
        ib = np.ascontiguousarray(b[0], np.uint32); vb = np.ascontiguousarray(b[1], np.float64)
        f = self._l.ref_voc_score
        f.restype = C.c_double
        VP = C.c_void_p
        f.argtypes = [VP, C.c_int, VP, VP, C.c_int, VP, VP]
        return float(f(self._h, len(ia), ia.ctypes.data, va.ctypes.data, len(ib), ib.ctypes.data, vb.ctypes.data))
