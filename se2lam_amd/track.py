"""Host-side mirror of the Track members that sit either side of MatchByWindow (/root/reference/src/Track.cpp).

    Track.removeOutliers(kp1, kp2, matches)   Track.cpp:308-344   cv::findFundamentalMat RANSAC mask on the device
    Track.doTriangulate(...)                  Track.cpp:378-419   see matcher.doTriangulate

Everything is computed by libse2gpu (csrc/ransac.hip, csrc/triangulate.hip); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


class Track:
    """Owns the device workspace of one tracking thread (the reference's Track is single-threaded)."""

    def __init__(self):
        h = C.c_void_p()
        capi.check(capi.lib().se2gpu_track_create(C.byref(h)))
        self._h = h

    def findFundamentalMat(self, pt1, pt2):
        """cv::findFundamentalMat(pt1, pt2, mask) with FM_RANSAC / 3 px / 0.99 -> (mask (n,) uint8, n_inliers)"""
        p1 = np.ascontiguousarray(pt1, np.float32).reshape(-1, 2)
        p2 = np.ascontiguousarray(pt2, np.float32).reshape(-1, 2)
        if len(p1) != len(p2):
            raise ValueError("pt1 and pt2 differ in length")
        n = len(p1)
        mask = np.zeros(max(n, 1), np.uint8)
        ni = C.c_int(0)
        capi.check(capi.lib().se2gpu_track_fundamental_mask(self._h, p1.ctypes.data, p2.ctypes.data, n, mask.ctypes.data,
                                                            C.byref(ni)))
        return mask[:n], int(ni.value)

    def removeOutliers(self, kp1, kp2, matches):
        """Track::removeOutliers: `matches` (int32, len(kp1)) is updated IN PLACE like the reference's vector<int>&;
        returns the number of inliers (0 when fewer than 10 survive, with every match dropped)."""
        k1 = np.ascontiguousarray(kp1); k2 = np.ascontiguousarray(kp2)
        if not (isinstance(matches, np.ndarray) and matches.dtype == np.int32 and matches.flags.c_contiguous):
            raise TypeError("matches must be a contiguous int32 array (it is modified in place)")
        if len(matches) != len(k1):
            raise ValueError("matches must have one entry per key point of kp1")
        ni = C.c_int(0)
        capi.check(capi.lib().se2gpu_track_remove_outliers(self._h, k1.ctypes.data, len(k1), k2.ctypes.data, len(k2),
                                                           matches.ctypes.data, C.byref(ni)))
        return int(ni.value)

    def last_ransac(self):
        """{inliers, sample, model, iterations} of the last findFundamentalMat"""
        info = np.zeros(4, np.int32)
        capi.check(capi.lib().se2gpu_track_last_ransac(self._h, info.ctypes.data))
        return dict(inliers=int(info[0]), sample=int(info[1]), model=int(info[2]), iterations=int(info[3]))

    def doTriangulate(self, kps_ref, kps_cur, match_idx, has_observation, P_ref, P_cur, Ocam, lower_depth, upper_depth,
                      min_degree=2):
        """Track::doTriangulate on this thread's workspace; same results as matcher.doTriangulate.
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
        capi.check(capi.lib().se2gpu_track_triangulate(
            self._h, n, kps_ref.ctypes.data, kps_cur.ctypes.data, len(kps_cur), m.ctypes.data,
            None if ho is None else ho.ctypes.data, P1.ctypes.data, P2.ctypes.data, oc.ctypes.data, float(lower_depth),
            float(upper_depth), int(min_degree), pos.ctypes.data, good.ctypes.data, C.byref(ng), C.byref(nold)))
        return pos[:n], good[:n], m, int(ng.value), int(nold.value)

    def __del__(self):
        try:
            if self._h:
                capi.lib().se2gpu_track_destroy(self._h)
                self._h = None
        except Exception:
            pass
