"""Python mirror of se2gpu_map_update_local_graph (Map::updateLocalGraph, /root/reference/src/Map.cpp:285-331) - harness for
the tests; the computation is host code inside libse2gpu.so."""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


def _csr(lists, n):
    ptr = np.zeros(n + 1, np.int32)
    for i, l in enumerate(lists):
        ptr[i + 1] = ptr[i] + len(l)
    idx = np.fromiter((x for l in lists for x in l), np.int32, count=int(ptr[-1])) if ptr[-1] else np.zeros(0, np.int32)
    return ptr, np.ascontiguousarray(idx)


def updateLocalGraph(kf_id, covisible, kf_obs, mp_id, mp_obs, current_kf, search_level=3):
    """kf_id[i], covisible[i] = positions of the key frames covisible with i, kf_obs[i] = positions of the map points i observes,
    mp_id[j], mp_obs[j] = positions of the key frames that observe j -> (mLocalGraphKFs, mRefKFs, mLocalGraphMPs) as positions."""
    K, M = len(kf_id), len(mp_id)
    keep = [np.ascontiguousarray(kf_id, np.int32), *_csr(covisible, K), *_csr(kf_obs, K), np.ascontiguousarray(mp_id, np.int32),
            *_csr(mp_obs, M)]
    v = capi.MapView()
    v.n_kf, v.n_mp = K, M
    (v.kf_id, v.covis_ptr, v.covis_idx, v.kf_mp_ptr, v.kf_mp_idx, v.mp_id, v.mp_kf_ptr, v.mp_kf_idx) = [a.ctypes.data for a in keep]
    lk, rk, lm = np.zeros(K, np.int32), np.zeros(K, np.int32), np.zeros(max(M, 1), np.int32)
    nl, nr, nm = C.c_int(), C.c_int(), C.c_int()
    capi.check(capi.lib().se2gpu_map_update_local_graph(C.byref(v), int(current_kf), int(search_level), lk.ctypes.data,
                                                        C.byref(nl), rk.ctypes.data, C.byref(nr), lm.ctypes.data, C.byref(nm)))
    return lk[:nl.value].copy(), rk[:nr.value].copy(), lm[:nm.value].copy()
