"""Python mirror of se2lam::ORBmatcher over the C ABI (harness view; C++ twin: include/se2lam_amd/ORBmatcher.h).

Reference interface: /root/reference/include/se2lam/ORBmatcher.h:46-77
    ORBmatcher(nnratio=0.6, checkOri=true)
    static DescriptorDistance(a, b)
    MatchByWindow(frame1, frame2, vbPrevMatched, winSize, vnMatches12, levelOffset=1, minLevel=0, maxLevel=8)
    MatchByProjection(pNewKF, localMPs, winSize, levelOffset, vMatchesIdxMP)
A "frame" here is the POD content the matchers read: key points (cv::KeyPoint layout), descriptors and the
image bounds that define the 64x48 grid (Frame.cpp:37-44).  All compute happens in libse2gpu.so (HIP).
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi

TH_HIGH, TH_LOW, HISTO_LENGTH = 100, 75, 30


def bounds(cols=640, rows=480):
    return capi.FrameBounds(0.0, 0.0, float(cols), float(rows))


class ORBmatcher:
    def __init__(self, nnratio=0.6, checkOri=True, max_features=4096, max_batch=1):
        assert checkOri, "the reference always checks orientation in MatchByWindow (ORBmatcher.cpp:350-372)"
        self.mfNNratio = float(nnratio)
        self._h = C.c_void_p()
        capi.check(capi.lib().se2gpu_matcher_create(max_features, max_batch, C.byref(self._h)))

    def spill_calls(self) -> int:
        """se2gpu_matcher_spill_calls: calls of this matcher in which a query took the exact spill scan (> 128 candidates)"""
        v = C.c_longlong(0)
        capi.check(capi.lib().se2gpu_matcher_spill_calls(self._h, C.byref(v)))
        return int(v.value)

    @staticmethod
    def DescriptorDistance(a, b) -> int:
        a = np.ascontiguousarray(a, np.uint8); b = np.ascontiguousarray(b, np.uint8)
        return int(capi.lib().se2gpu_hamming(a.ctypes.data, b.ctypes.data))

    def MatchByWindow(self, kps1, desc1, kps2, desc2, vbPrevMatched, winSize, levelOffset=1, minLevel=0, maxLevel=8,
                      frame_bounds=None):
        """-> (nmatches, vnMatches12); vbPrevMatched (n1,2) float32 is updated in place."""
        fb = frame_bounds or bounds()
        kps1 = np.ascontiguousarray(kps1); kps2 = np.ascontiguousarray(kps2)
        desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
        assert vbPrevMatched.dtype == np.float32 and vbPrevMatched.flags.c_contiguous
        n1, n2 = len(kps1), len(kps2)
        m12 = np.full(max(n1, 1), -1, np.int32)
        nm = C.c_int(0)
        capi.check(capi.lib().se2gpu_match_window(self._h, C.byref(fb), kps1.ctypes.data, desc1.ctypes.data, n1,
                                                  kps2.ctypes.data, desc2.ctypes.data, n2, vbPrevMatched.ctypes.data,
                                                  winSize, levelOffset, minLevel, maxLevel, self.mfNNratio,
                                                  m12.ctypes.data, C.byref(nm)))
        return nm.value, m12[:n1].copy()

    def MatchByProjection(self, mp_pos, mp_desc, mp_octave, mp_skip, Tcw, K4, kps, desc, kf_observed, winSize,
                          levelOffset, frame_bounds=None):
        """-> (nmatches, vMatchesIdxMP).  mp_skip[i]=1: the reference skips the map point (null / bad parallax /
        already observed, ORBmatcher.cpp:392-395); kf_observed[idx]=1: hasObservation(idx) (:417)."""
        fb = frame_bounds or bounds()
        mp_pos = np.ascontiguousarray(mp_pos, np.float32); mp_desc = np.ascontiguousarray(mp_desc, np.uint8)
        mp_octave = np.ascontiguousarray(mp_octave, np.int32); mp_skip = np.ascontiguousarray(mp_skip, np.uint8)
        Tcw = np.ascontiguousarray(Tcw, np.float32).reshape(-1)[:12].copy()
        kps = np.ascontiguousarray(kps); desc = np.ascontiguousarray(desc, np.uint8)
        kf_observed = np.ascontiguousarray(kf_observed, np.uint8)
        n, m = len(kps), len(mp_octave)
        out = np.full(max(n, 1), -1, np.int32)
        nm = C.c_int(0)
        fx, fy, cx, cy = [float(v) for v in K4]
        capi.check(capi.lib().se2gpu_match_projection(self._h, C.byref(fb), mp_pos.ctypes.data, mp_desc.ctypes.data,
                                                      mp_octave.ctypes.data, mp_skip.ctypes.data, m, Tcw.ctypes.data,
                                                      fx, fy, cx, cy, kps.ctypes.data, desc.ctypes.data,
                                                      kf_observed.ctypes.data, n, winSize, levelOffset, self.mfNNratio,
                                                      out.ctypes.data, C.byref(nm)))
        return nm.value, out[:n].copy()

    def SearchByBoW(self, kps1, desc1, fv1, has_mp1, kps2, desc2, fv2, has_mp2, bIfMPOnly=True, checkOri=True):
        """ORBmatcher.h:55 SearchByBoW(pKF1, pKF2, mapMatches12, bIfMPOnly).  fv = (nodes, ptr, idx) int32 CSR of the
        key frame's DBoW2::FeatureVector (host vocabulary).  -> (nmatches, matches12 (n1,), -1 = no match)"""
        kps1 = np.ascontiguousarray(kps1); kps2 = np.ascontiguousarray(kps2)
        desc1 = np.ascontiguousarray(desc1, np.uint8); desc2 = np.ascontiguousarray(desc2, np.uint8)
        a = [np.ascontiguousarray(x, np.int32) for x in fv1]; b = [np.ascontiguousarray(x, np.int32) for x in fv2]
        h1 = np.ascontiguousarray(has_mp1, np.uint8); h2 = np.ascontiguousarray(has_mp2, np.uint8)
        out = np.full(max(len(kps1), 1), -1, np.int32)
        nm = C.c_int(0)
        capi.check(capi.lib().se2gpu_search_by_bow(self._h, kps1.ctypes.data, desc1.ctypes.data, len(kps1),
                                                   a[0].ctypes.data, a[1].ctypes.data, a[2].ctypes.data, len(a[0]),
                                                   h1.ctypes.data, kps2.ctypes.data, desc2.ctypes.data, len(kps2),
                                                   b[0].ctypes.data, b[1].ctypes.data, b[2].ctypes.data, len(b[0]),
                                                   h2.ctypes.data, int(bIfMPOnly), self.mfNNratio, int(checkOri),
                                                   out.ctypes.data, C.byref(nm)))
        return nm.value, out[:len(kps1)].copy()

    # -- batched, device resident -----------------------------------------------------------
    def match_window_batch_device(self, d_kps, d_desc, d_counts, cap, d_pair_a, d_pair_b, npairs, winSize,
                                  d_matches12, d_nmatches, levelOffset=1, minLevel=0, maxLevel=8, frame_bounds=None):
        fb = frame_bounds or bounds()
        capi.check(capi.lib().se2gpu_match_window_batch_device(self._h, C.byref(fb), d_kps, d_desc, d_counts, cap,
                                                               d_pair_a, d_pair_b, npairs, winSize, levelOffset,
                                                               minLevel, maxLevel, self.mfNNratio, d_matches12,
                                                               d_nmatches))

    def sync(self):
        capi.check(capi.lib().se2gpu_matcher_sync(self._h))

    def stream(self):
        return capi.lib().se2gpu_matcher_stream(self._h)

    def __del__(self):
        try:
            if self._h:
                capi.lib().se2gpu_matcher_destroy(self._h)
                self._h = None
        except Exception:
            pass


def doTriangulate(kps_ref, kps_cur, match_idx, has_observation, P_ref, P_cur, Ocam, lower_depth, upper_depth,
                  min_degree=2):
    """Track::doTriangulate (Track.cpp:378-419) for all matches of a frame pair on the device.
    -> (pos (n,3) float32, good_parallax (n,) uint8, match_idx updated (n,) int32, n_good, n_tracked_old)"""
    kps_ref = np.ascontiguousarray(kps_ref); kps_cur = np.ascontiguousarray(kps_cur)
    n = len(kps_ref)
    m = np.ascontiguousarray(match_idx, np.int32).copy()
    ho = None if has_observation is None else np.ascontiguousarray(has_observation, np.uint8)
    P1 = np.ascontiguousarray(P_ref, np.float32).reshape(-1); P2 = np.ascontiguousarray(P_cur, np.float32).reshape(-1)
    oc = np.ascontiguousarray(Ocam, np.float32)
    pos = np.zeros((max(n, 1), 3), np.float32)
    good = np.zeros(max(n, 1), np.uint8)
    ng, nold = C.c_int(0), C.c_int(0)
    capi.check(capi.lib().se2gpu_triangulate(n, kps_ref.ctypes.data, kps_cur.ctypes.data, len(kps_cur), m.ctypes.data,
                                             None if ho is None else ho.ctypes.data, P1.ctypes.data, P2.ctypes.data,
                                             oc.ctypes.data, float(lower_depth), float(upper_depth), int(min_degree),
                                             pos.ctypes.data, good.ctypes.data, C.byref(ng), C.byref(nold)))
    return pos[:n], good[:n], m, int(ng.value), int(nold.value)
