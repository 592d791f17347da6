"""Python mirror of Sparsifier::DoMarginalizeSE3XYZ (/root/reference/src/sparsifier.cpp:105-275) over se2gpu_sparsify_se3xyz -
harness for the tests; the computation is the HIP kernel k_sparsify."""
from __future__ import annotations

import numpy as np

from . import capi


def DoMarginalizeSE3XYZ_batch(pairs):
    """pairs: list of (kf (2,4,4) T_w_c, mp (N,3), m_kf (M,), m_mp (M,), m_info (M,3,3)) -> list of (z_out 4x4, info_out 6x6)"""
    n = len(pairs)
    kf = np.zeros((n, 2, 12))
    mp_ptr = np.zeros(n + 1, np.int32)
    m_ptr = np.zeros(n + 1, np.int32)
    for p, (k, mp, mk, mm, mi) in enumerate(pairs):
        k = np.asarray(k, np.float64)
        kf[p] = np.concatenate([k[:, :3, :3].reshape(2, 9), k[:, :3, 3]], axis=1)
        mp_ptr[p + 1] = mp_ptr[p] + len(mp)
        m_ptr[p + 1] = m_ptr[p] + len(mk)
    mp = np.ascontiguousarray(np.concatenate([np.asarray(q[1], np.float64).reshape(-1, 3) for q in pairs]))
    mk = np.ascontiguousarray(np.concatenate([np.asarray(q[2], np.int32) for q in pairs]))
    mm = np.ascontiguousarray(np.concatenate([np.asarray(q[3], np.int32) for q in pairs]))
    mi = np.ascontiguousarray(np.concatenate([np.asarray(q[4], np.float64).reshape(-1, 9) for q in pairs]))
    z = np.zeros((n, 12))
    info = np.zeros((n, 36))
    capi.check(capi.lib().se2gpu_sparsify_se3xyz(n, kf.ctypes.data, mp_ptr.ctypes.data, mp.ctypes.data, m_ptr.ctypes.data,
                                                 mk.ctypes.data, mm.ctypes.data, mi.ctypes.data, z.ctypes.data, info.ctypes.data))
    out = []
    for p in range(n):
        T = np.eye(4)
        T[:3, :3] = z[p, :9].reshape(3, 3)
        T[:3, 3] = z[p, 9:]
        out.append((T, info[p].reshape(6, 6).copy()))
    return out


def DoMarginalizeSE3XYZ(kf, mp, m_kf, m_mp, m_info):
    return DoMarginalizeSE3XYZ_batch([(kf, mp, m_kf, m_mp, m_info)])[0]
