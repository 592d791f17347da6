"""ctypes binding of include/se2gpu.h (libse2gpu.so).

This is harness plumbing for tests/ and bench.py: the product is the C-ABI library and the C++
adapters in include/se2lam_amd/.  There is NO fallback: if the shared library is missing this
module raises, and if no GPU is visible every compute call raises `Se2GpuError`.
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "lib", "libse2gpu.so")

OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_CAPACITY, ERR_STATE = -1, -2, -3, -4, -5


class Se2GpuError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__(f"se2gpu error {code}: {msg}")
        self.code = code


class Keypoint(C.Structure):
    _fields_ = [("x", C.c_float), ("y", C.c_float), ("size", C.c_float), ("angle", C.c_float),
                ("response", C.c_float), ("octave", C.c_int32), ("class_id", C.c_int32)]


KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("size", "<f4"), ("angle", "<f4"), ("response", "<f4"),
                     ("octave", "<i4"), ("class_id", "<i4")])
assert KP_DTYPE.itemsize == 28 == C.sizeof(Keypoint)


class OrbParams(C.Structure):
    _fields_ = [("nfeatures", C.c_int32), ("scale_factor", C.c_float), ("nlevels", C.c_int32),
                ("score_type", C.c_int32), ("fast_th", C.c_int32), ("max_rows", C.c_int32),
                ("max_cols", C.c_int32), ("max_batch", C.c_int32)]


class FrameBounds(C.Structure):
    _fields_ = [("min_x", C.c_float), ("min_y", C.c_float), ("max_x", C.c_float), ("max_y", C.c_float)]


class LocalGraph(C.Structure):
    _fields_ = [("n_local_kf", C.c_int32), ("n_ref_kf", C.c_int32), ("n_mp", C.c_int32), ("n_obs", C.c_int32),
                ("kf_id", C.c_void_p), ("kf_Twb", C.c_void_p), ("kf_Rcw", C.c_void_p),
                ("odo_to", C.c_void_p), ("odo_meas", C.c_void_p), ("odo_cov", C.c_void_p),
                ("mp_pos", C.c_void_p), ("obs_mp", C.c_void_p), ("obs_kf", C.c_void_p), ("obs_uv", C.c_void_p),
                ("obs_lc", C.c_void_p), ("obs_sigma2", C.c_void_p),
                ("fx", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
                ("Rbc", C.c_double * 9), ("tbc", C.c_double * 3),
                ("huber_delta", C.c_float), ("xrot_info", C.c_float), ("z_info", C.c_float)]


class MapView(C.Structure):
    _fields_ = [("n_kf", C.c_int32), ("n_mp", C.c_int32), ("kf_id", C.c_void_p), ("covis_ptr", C.c_void_p),
                ("covis_idx", C.c_void_p), ("kf_mp_ptr", C.c_void_p), ("kf_mp_idx", C.c_void_p), ("mp_id", C.c_void_p),
                ("mp_kf_ptr", C.c_void_p), ("mp_kf_idx", C.c_void_p)]


class BaStats(C.Structure):
    _fields_ = [("iterations", C.c_int32), ("trials", C.c_int32), ("terminated", C.c_int32),
                ("stopped", C.c_int32), ("chi2_init", C.c_double), ("chi2_final", C.c_double),
                ("lambda_final", C.c_double), ("chi2_hist", C.c_double * 64), ("lambda_hist", C.c_double * 64),
                ("trials_hist", C.c_int32 * 64)]


ALLREDUCE_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p)

# every symbol declared in include/se2gpu.h: name -> (restype, argtypes)
_VP, _I, _D, _F, _SZ = C.c_void_p, C.c_int, C.c_double, C.c_float, C.c_size_t
_PD, _PI32, _PU8, _PF = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_uint8), C.POINTER(C.c_float)
_PKP = C.POINTER(Keypoint)
SYMBOLS = {
    "se2gpu_last_error": (C.c_char_p, []),
    "se2gpu_device_count": (_I, []),
    "se2gpu_version": (C.c_char_p, []),
    # ORB extractor
    "se2gpu_orb_create": (_I, [C.POINTER(OrbParams), C.POINTER(_VP)]),
    "se2gpu_orb_destroy": (None, [_VP]),
    "se2gpu_orb_levels": (_I, [_VP]),
    "se2gpu_orb_scale_factor": (_F, [_VP]),
    "se2gpu_orb_extract": (_I, [_VP, _VP, _I, _I, _SZ, _VP, _VP, _VP, _I, C.POINTER(_I)]),
    "se2gpu_orb_extract_batch_device": (_I, [_VP, _VP, _I, _I, _I, _VP, _VP, _VP, _I]),
    "se2gpu_orb_sync": (_I, [_VP]),
    "se2gpu_orb_set_stream": (_I, [_VP, _VP]),
    "se2gpu_orb_debug_level": (_I, [_VP, _I, _I, _I, _VP, _SZ, C.POINTER(_I), C.POINTER(_I)]),
    "se2gpu_orb_debug_score": (_I, [_VP, _I, _I, _VP, _SZ, C.POINTER(_I), C.POINTER(_I)]),
    "se2gpu_orb_debug_nth_element": (_I, [_VP, _I, _I, _I]),
    "se2gpu_orb_stream": (_VP, [_VP]),
    "se2gpu_orb_score_kernel": (_I, [_VP, _VP]),
    "se2gpu_orb_profile": (_I, [_VP, _I]),
    "se2gpu_orb_profile_get": (_I, [_VP, _I, C.POINTER(C.c_char_p), _PD, C.POINTER(C.c_int64)]),
    # matcher
    "se2gpu_hamming": (_I, [_VP, _VP]),
    "se2gpu_three_maxima": (_I, [_VP, _I, _VP, _VP, _VP]),
    "se2gpu_matcher_create": (_I, [_I, _I, C.POINTER(_VP)]),
    "se2gpu_matcher_destroy": (None, [_VP]),
    "se2gpu_matcher_set_stream": (_I, [_VP, _VP]),
    "se2gpu_matcher_sync": (_I, [_VP]),
    "se2gpu_matcher_spill_calls": (_I, [_VP, _VP]),
    "se2gpu_matcher_stream": (_VP, [_VP]),
    "se2gpu_match_window": (_I, [_VP, C.POINTER(FrameBounds), _VP, _VP, _I, _VP, _VP, _I, _VP, _I, _I, _I, _I, _F,
                                 _VP, C.POINTER(_I)]),
    "se2gpu_match_window_batch_device": (_I, [_VP, C.POINTER(FrameBounds), _VP, _VP, _VP, _I, _VP, _VP, _I, _I, _I,
                                              _I, _I, _F, _VP, _VP]),
    "se2gpu_search_by_bow": (_I, [_VP, _VP, _VP, _I, _VP, _VP, _VP, _I, _VP, _VP, _VP, _I, _VP, _VP, _VP, _I, _VP, _I, _F, _I,
                                  _VP, C.POINTER(_I)]),
    "se2gpu_match_projection": (_I, [_VP, C.POINTER(FrameBounds), _VP, _VP, _VP, _VP, _I, _VP, _F, _F, _F, _F,
                                     _VP, _VP, _VP, _I, _I, _I, _F, _VP, C.POINTER(_I)]),
    # BA
    "se2gpu_ba_create": (_I, [C.POINTER(_VP)]),
    "se2gpu_ba_reserve": (_I, [_I, _I, _I]),
    "se2gpu_ba_destroy": (None, [_VP]),
    "se2gpu_ba_clear": (_I, [_VP]),
    "se2gpu_ba_set_stream": (_I, [_VP, _VP]),
    "se2gpu_ba_stream": (_VP, [_VP]),
    "se2gpu_ba_add_cam": (_I, [_VP, _D, _D, _D]),
    "se2gpu_ba_set_Tbc": (_I, [_VP, _PD, _PD]),
    "se2gpu_ba_add_vertex_se2": (_I, [_VP, _I, _D, _D, _D, _I]),
    "se2gpu_ba_add_vertex_xyz": (_I, [_VP, _I, _PD, _I, _I]),
    "se2gpu_ba_add_edge_se2xyz": (_I, [_VP, _I, _I, _PD, _PD, _D]),
    "se2gpu_ba_add_edge_se2": (_I, [_VP, _I, _I, _PD, _PD]),
    "se2gpu_ba_load": (_I, [_VP, _I, _I, _I, _I, _PD, _PU8, _PD, _PI32, _PI32, _PD, _PD, _PI32, _PI32, _PD, _PD, _D]),
    "se2gpu_ba_load_local_graph": (_I, [_VP, C.POINTER(LocalGraph)]),
    "se2gpu_ba_add_vertex_se3": (_I, [_VP, _I, _PD, _I]),
    "se2gpu_ba_add_vertex_iso3": (_I, [_VP, _I, _PD, _I]),
    "se2gpu_plane_motion_prior_iso3": (_I, [_VP, _VP, _D, _D, _D, _VP, _VP]),
    "se2gpu_ba_add_prior_se3": (_I, [_VP, _I, _PD, _PD]),
    "se2gpu_ba_add_edge_se3": (_I, [_VP, _I, _I, _PD, _PD]),
    "se2gpu_ba_add_edge_xyz2uv": (_I, [_VP, _I, _I, _PD, _D, _D]),
    "se2gpu_ba_get_se3": (_I, [_VP, _I, _PD]),
    "se2gpu_ba_edge_chi2": (_I, [_VP, _PD, _I]),
    "se2gpu_map_update_local_graph": (_I, [C.POINTER(MapView), _I, _I, _VP, C.POINTER(_I), _VP, C.POINTER(_I), _VP, C.POINTER(_I)]),
    "se2gpu_sparsify_se3xyz": (_I, [_I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP]),
    "se2gpu_ba_initialize": (_I, [_VP]),
    "se2gpu_ba_set_edge_level": (_I, [_VP, _I, _I]),
    "se2gpu_ba_reset_estimates": (_I, [_VP]),
    "se2gpu_ba_reset_estimates_batch": (_I, [_VP, _I]),
    "se2gpu_ba_optimize": (_I, [_VP, _I, _I, _PU8, _I, C.POINTER(BaStats)]),
    "se2gpu_ba_optimize_batch": (_I, [C.POINTER(_VP), _I, _I, _I, _PU8, C.POINTER(BaStats)]),
    "se2gpu_ba_last_batch_path": (_I, []),
    "se2gpu_ba_get_se2": (_I, [_VP, _I, _PD]),
    "se2gpu_ba_get_xyz": (_I, [_VP, _I, _PD]),
    "se2gpu_ba_get_all": (_I, [_VP, _PD, _PD]),
    "se2gpu_ba_chi2": (_D, [_VP]),
    "se2gpu_ba_debug_reduced_system": (_I, [_VP, _D, _PD, _PD]),
    "se2gpu_ba_debug_solve": (_I, [_VP, _D, _PD, C.POINTER(C.c_int)]),
    "se2gpu_ba_debug_solver_path": (_I, [_VP]),
    "se2gpu_ba_debug_chol_verify": (_I, [_VP, _VP, _VP, _I]),
    "se2gpu_ba_debug_pool_sizes": (_I, [_VP]),
    "se2gpu_ba_debug_solve_plan": (_I, [_I, _I, _VP, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _I]),
    "se2gpu_ba_debug_solve_plan_tile": (_I, [_I, _I, _VP, _I, _I, _VP, _VP, _VP, _VP, _VP, _VP, _VP, _I, _VP, _I]),
    "se2gpu_triangulate": (_I, [_I, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, C.c_float, C.c_float, _I, _VP, _VP,
                           C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "se2gpu_track_create": (_I, [C.POINTER(_VP)]),
    "se2gpu_track_destroy": (None, [_VP]),
    "se2gpu_track_fundamental_mask": (_I, [_VP, _VP, _VP, _I, _VP, C.POINTER(C.c_int)]),
    "se2gpu_track_remove_outliers": (_I, [_VP, _VP, _I, _VP, _I, _VP, C.POINTER(C.c_int)]),
    "se2gpu_track_last_ransac": (_I, [_VP, _VP]),
    "se2gpu_plane_motion_prior": (_I, [_VP, _VP, _D, _D, _D, _VP, _VP]),
    "se2gpu_track_pose_ba": (_I, [_VP, _VP, _VP, _VP, _I, _VP, _VP, _VP, _D, _D, _D, _D, _I, _VP, C.POINTER(BaStats)]),
    "se2gpu_track_triangulate": (_I, [_VP, _I, _VP, _VP, _I, _VP, _VP, _VP, _VP, _VP, C.c_float, C.c_float, _I, _VP, _VP,
                                 C.POINTER(C.c_int), C.POINTER(C.c_int)]),
    "se2gpu_ba_reduce_buffer_doubles": (_SZ, [_VP, _I]),
    "se2gpu_ba_exchange_doubles": (_SZ, [_I]),
    "se2gpu_ba_exchange_doubles_h": (_SZ, [_VP]),
    "se2gpu_ba_exchange_row": (_I, [_I, C.POINTER(C.c_size_t), C.POINTER(C.c_int)]),
    "se2gpu_ba_set_allreduce": (_I, [_VP, ALLREDUCE_FN, _VP, _VP]),
    "se2gpu_ba_set_shard": (_I, [_VP, _I, _I]),
    "se2gpu_comm_unique_id": (_I, [_VP]),
    "se2gpu_comm_create": (_I, [_VP, _I, _I, C.POINTER(_VP)]),
    "se2gpu_comm_destroy": (None, [_VP]),
    "se2gpu_comm_count": (_I, [_VP, C.POINTER(_I)]),
    "se2gpu_comm_allreduce_sum_f64": (_I, [_VP, _VP, _SZ, _VP]),
    "se2gpu_ba_set_comm": (_I, [_VP, _VP]),
    "se2gpu_ba_shard_landmarks": (_I, [_I, _I, _PI32, _PI32, _I, _PI32]),
    "se2gpu_ba_edge_information": (_I, [_I, _VP, _VP, _VP, _VP, _I, _VP, _VP, _F, _F, _F, _VP]),
    "se2gpu_ba_profile": (_I, [_VP, _I]),
    "se2gpu_ba_profile_get": (_I, [_VP, _I, C.POINTER(C.c_char_p), _PD, C.POINTER(C.c_int64)]),
    # timers / memory
    "se2gpu_timer_create": (_I, [C.POINTER(_VP)]),
    "se2gpu_timer_destroy": (None, [_VP]),
    "se2gpu_timer_start": (_I, [_VP, _VP]),
    "se2gpu_timer_stop": (_I, [_VP, _VP]),
    "se2gpu_timer_elapsed_ms": (_I, [_VP, C.POINTER(C.c_float)]),
    "se2gpu_malloc": (_I, [C.POINTER(_VP), _SZ]),
    "se2gpu_free": (_I, [_VP]),
    "se2gpu_memcpy_h2d": (_I, [_VP, _VP, _SZ]),
    "se2gpu_memcpy_d2h": (_I, [_VP, _VP, _SZ]),
    "se2gpu_host_alloc": (_I, [C.POINTER(_VP), _SZ]),
    "se2gpu_host_free": (_I, [_VP]),
    "se2gpu_memcpy_h2d_async": (_I, [_VP, _VP, _SZ, _VP]),
    "se2gpu_memcpy_d2h_async": (_I, [_VP, _VP, _SZ, _VP]),
    "se2gpu_stream_create": (_I, [C.POINTER(_VP)]),
    "se2gpu_stream_destroy": (_I, [_VP]),
    "se2gpu_stream_synchronize": (_I, [_VP]),
    "se2gpu_timer_stream_wait": (_I, [_VP, _VP]),
    "se2gpu_device_synchronize": (_I, []),
    "se2gpu_set_device": (_I, [_I]),
}

_lib = None


def lib() -> C.CDLL:
    """Load libse2gpu.so (fails loudly when it has not been built: there is no fallback)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                f"{LIB_PATH} is missing - run `python -m se2lam_amd.build` (or __graft_entry__.build()). "
                "se2lam_amd has no CPU / eager fallback.")
        l = C.CDLL(LIB_PATH)
        for name, (res, args) in SYMBOLS.items():
            fn = getattr(l, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = l
    return _lib


def check(rc: int):
    if rc != OK:
        raise Se2GpuError(rc, lib().se2gpu_last_error().decode(errors="replace"))


def device_count() -> int:
    return int(lib().se2gpu_device_count())


def vp(a: np.ndarray) -> C.c_void_p:
    return C.c_void_p(a.ctypes.data)


def pd(a: np.ndarray):
    return a.ctypes.data_as(_PD)


class DeviceArray:
    """A raw device allocation (se2gpu_malloc) with numpy round-trips; for tests and bench."""

    def __init__(self, nbytes: int):
        self.ptr = C.c_void_p()
        self.nbytes = int(nbytes)
        check(lib().se2gpu_malloc(C.byref(self.ptr), self.nbytes))

    @classmethod
    def from_numpy(cls, a: np.ndarray) -> "DeviceArray":
        a = np.ascontiguousarray(a)
        d = cls(a.nbytes)
        if a.nbytes:
            check(lib().se2gpu_memcpy_h2d(d.ptr, vp(a), a.nbytes))
        return d

    def to_numpy(self, dtype, shape) -> np.ndarray:
        out = np.empty(shape, dtype=dtype)
        assert out.nbytes <= self.nbytes
        if out.nbytes:
            check(lib().se2gpu_memcpy_d2h(vp(out), self.ptr, out.nbytes))
        return out

    def free(self):
        if self.ptr:
            lib().se2gpu_free(self.ptr)
            self.ptr = C.c_void_p()

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class PinnedArray:
    """Pinned host allocation (se2gpu_host_alloc) viewed as a numpy array; for the streaming bench."""

    def __init__(self, shape, dtype):
        self.ptr = C.c_void_p()
        n = int(np.prod(shape)) * np.dtype(dtype).itemsize
        check(lib().se2gpu_host_alloc(C.byref(self.ptr), n))
        buf = (C.c_uint8 * n).from_address(self.ptr.value)
        self.array = np.frombuffer(buf, dtype=dtype).reshape(shape)

    def __del__(self):
        try:
            if self.ptr:
                self.array = None
                lib().se2gpu_host_free(self.ptr)
                self.ptr = C.c_void_p()
        except Exception:
            pass


class Stream:
    def __init__(self):
        self.h = C.c_void_p()
        check(lib().se2gpu_stream_create(C.byref(self.h)))

    def sync(self):
        check(lib().se2gpu_stream_synchronize(self.h))

    def __del__(self):
        try:
            if self.h:
                lib().se2gpu_stream_destroy(self.h)
        except Exception:
            pass


class Timer:
    def __init__(self):
        self.h = C.c_void_p()
        check(lib().se2gpu_timer_create(C.byref(self.h)))

    def start(self, stream):
        check(lib().se2gpu_timer_start(self.h, stream))

    def stop(self, stream):
        check(lib().se2gpu_timer_stop(self.h, stream))

    def make_wait(self, stream):
        """`stream` waits for this timer's stop event (cross-stream ordering without a host sync)"""
        check(lib().se2gpu_timer_stream_wait(self.h, stream))

    def elapsed_ms(self) -> float:
        ms = C.c_float()
        check(lib().se2gpu_timer_elapsed_ms(self.h, C.byref(ms)))
        return float(ms.value)

    def __del__(self):
        try:
            if self.h:
                lib().se2gpu_timer_destroy(self.h)
        except Exception:
            pass
