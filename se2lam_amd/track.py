"""Host-side mirror of the Track members that sit either side of MatchByWindow (/root/reference/src/Track.cpp).

    Track.removeOutliers(kp1, kp2, matches)   Track.cpp:308-344   cv::findFundamentalMat RANSAC mask on the device
    Track.doTriangulate(...)                  Track.cpp:378-419   see matcher.doTriangulate

Everything is computed by libse2gpu (csrc/ransac.hip, csrc/triangulate.hip); there is no CPU fallback.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi
from .matcher import doTriangulate as _do_triangulate


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

    doTriangulate = staticmethod(_do_triangulate)

    def __del__(self):
        try:
            if self._h:
                capi.lib().se2gpu_track_destroy(self._h)
                self._h = None
        except Exception:
            pass
