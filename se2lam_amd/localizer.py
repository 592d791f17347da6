"""Host-side mirror of Localizer::DoLocalBA (/root/reference/src/Localizer.cpp:233-302): pose-only bundle adjustment of
the current key frame against the fixed map points it observes, computed by libse2gpu (csrc/pose_ba.hip) in one launch.
There is no CPU fallback."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


def _pose12(T) -> np.ndarray:
    T = np.asarray(T, np.float64)
    if T.shape == (12,):
        return np.ascontiguousarray(T)
    return np.ascontiguousarray(np.concatenate([T[:3, :3].reshape(-1), T[:3, 3]]))


def _matrix(p) -> np.ndarray:
    T = np.eye(4)
    T[:3, :3] = np.asarray(p[:9]).reshape(3, 3)
    T[:3, 3] = p[9:12]
    return T


def addPlaneMotionSE3Expmap(Tcw, bTc, xrot_info=1e6, yrot_info=1e6, z_info=1.0):
    """optimizer.cpp:236-314 -> (measurement 4x4, information 6x6); the defaults are Config::PLANEMOTION_*_INFO"""
    meas = np.zeros(12); info = np.zeros(36)
    a, b = _pose12(Tcw), _pose12(bTc)
    capi.check(capi.lib().se2gpu_plane_motion_prior(a.ctypes.data, b.ctypes.data, float(xrot_info), float(yrot_info),
                                                    float(z_info), meas.ctypes.data, info.ctypes.data))
    return _matrix(meas), info.reshape(6, 6)


class Localizer:
    """Device workspace of the localisation thread."""

    def __init__(self):
        h = C.c_void_p()
        capi.check(capi.lib().se2gpu_track_create(C.byref(h)))
        self._h = h
        self.stats = None

    def DoLocalBA(self, Tcw, bTc, map_points, keypoints_uv, inv_sigma2, fx, cx, cy, th_huber, iterations=30,
                  plane_info=(1e6, 1e6, 1.0)):
        """Tcw: 4x4 pose of mpKFCurr; map_points (n,3) world positions of its good-parallax observations; keypoints_uv
        (n,2) = keyPointsUn[ftrIdx].pt; inv_sigma2 (n,) = mvInvLevelSigma2[octave]; th_huber = Config::TH_HUBER.
        -> optimised Tcw (4x4); self.stats holds the LM history."""
        meas, info = addPlaneMotionSE3Expmap(Tcw, bTc, *plane_info)
        return self.pose_ba(Tcw, meas, info, map_points, keypoints_uv, inv_sigma2, fx, cx, cy, th_huber, iterations)

    def pose_ba(self, Tcw, prior_meas, prior_info, xyz, uv, inv_sigma2, f, cx, cy, delta, iterations=30):
        xyz = np.ascontiguousarray(xyz, np.float64).reshape(-1, 3)
        uv = np.ascontiguousarray(uv, np.float64).reshape(-1, 2)
        w = np.ascontiguousarray(inv_sigma2, np.float64).reshape(-1)
        if not (len(xyz) == len(uv) == len(w)):
            raise ValueError("map points, key points and inv_sigma2 differ in length")
        a, m = _pose12(Tcw), _pose12(prior_meas)
        pi = np.ascontiguousarray(prior_info, np.float64).reshape(-1)
        out = np.zeros(12)
        st = capi.BaStats()
        capi.check(capi.lib().se2gpu_track_pose_ba(self._h, a.ctypes.data, m.ctypes.data, pi.ctypes.data, len(xyz),
                                                   xyz.ctypes.data, uv.ctypes.data, w.ctypes.data, float(f), float(cx),
                                                   float(cy), float(delta), int(iterations), out.ctypes.data, C.byref(st)))
        n = min(st.iterations, 64)
        self.stats = dict(iterations=st.iterations, trials=st.trials, terminated=bool(st.terminated),
                          chi2_init=st.chi2_init, chi2_final=st.chi2_final, lambda_final=st.lambda_final,
                          chi2_hist=list(st.chi2_hist[:n]), lambda_hist=list(st.lambda_hist[:n]),
                          trials_hist=list(st.trials_hist[:n]))
        return _matrix(out)

    def __del__(self):
        try:
            if self._h:
                capi.lib().se2gpu_track_destroy(self._h)
                self._h = None
        except Exception:
            pass
