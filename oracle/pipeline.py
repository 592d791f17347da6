"""ORACLE - TEST INFRASTRUCTURE ONLY.  ctypes loader of the two pipeline builds of oracle/_ref (`make -C oracle pipeline`):

  libse2lam_pipeline_cpu.so     the reference's Track -> LocalMapper -> optimizer, every source file its own (ORBextractor.cpp and
                                ORBmatcher.cpp included), compiled where it lies; g2o's optimize() and cv::findFundamentalMat from the
                                oracle's restatements.  This is BASELINE.json configs[0], the CPU reference run.
  libse2lam_pipeline_dropin.so  the same reference sources with ORBextractor.cpp / ORBmatcher.cpp replaced by tests/dropin/*.cpp (the
                                bindings of INTEGRATION.md over libse2gpu) and optimize() / findFundamentalMat forwarded to libse2gpu.
                                Needs a GPU at run time; it links no oracle code.

Both are driven by oracle/ref_pipeline_driver.cpp: one call per frame = the body of Track::run's loop, then the body of
LocalMapper::run's loop.  The libraries can only be BUILT where /root/reference exists (this container); they are git-ignored but
travel to the GPU box with the snapshot.  Only tests/ and bench.py's cpu_baseline leg import this module."""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
REFERENCE = os.environ.get("SE2LAM_REFERENCE", "/root/reference")
LIBS = {"cpu": os.path.join(HERE, "_ref", "libse2lam_pipeline_cpu.so"),
        "dropin": os.path.join(HERE, "_ref", "libse2lam_pipeline_dropin.so")}
_libs: dict = {}

FX, CX, CY = 400.0, 320.0, 240.0
Z0 = 3000.0                       # distance of the textured plane from the camera (mm): the frames of synth.frame(t) are crops of one
SHIFT = (3.0, 1.0)                # texture moving by (3, 1) px per frame = a camera looking along the body's z axis at a plane Z0 away
TBC = (100.0, 0.0, 300.0)         # while the body translates in its plane (the ceiling-camera rig of tests/test_pipeline.py)


class Config(C.Structure):
    _fields_ = [("K", C.c_float * 9), ("bTc", C.c_float * 16), ("upper_depth", C.c_float), ("lower_depth", C.c_float),
                ("scale_factor", C.c_float), ("max_level", C.c_int32), ("max_features", C.c_int32),
                ("odo_noise", C.c_float * 3), ("odo_uncertain", C.c_float * 3),
                ("planemotion_z_info", C.c_float), ("planemotion_xrot_info", C.c_float), ("planemotion_yrot_info", C.c_float),
                ("th_huber2", C.c_float), ("local_iter", C.c_int32), ("fps", C.c_int32)]


class FrameRecord(C.Structure):
    _fields_ = [(n, C.c_int32) for n in ("frame_id", "n_keypoints", "n_raw_matches", "n_matches", "new_kf", "local_ba", "n_kfs", "n_mps",
                                         "n_good_prl", "n_local_kfs", "n_local_mps", "n_ref_kfs", "n_match_entries", "n_raw_entries")] + \
               [("kp_hash", C.c_uint64), ("desc_hash", C.c_uint64), ("ms_track", C.c_double), ("ms_mapper", C.c_double),
                ("ba", C.c_double * 10), ("Twb", C.c_float * 3), ("Tcw", C.c_float * 16)]


def default_config() -> Config:
    """SURVEY.md section 8(d)'s camera and settings; extrinsic = camera axes parallel to the body's, lever arm TBC"""
    c = Config()
    c.K[:] = [FX, 0, CX, 0, FX, CY, 0, 0, 1]
    c.bTc[:] = [1, 0, 0, TBC[0], 0, 1, 0, TBC[1], 0, 0, 1, TBC[2], 0, 0, 0, 1]
    c.upper_depth, c.lower_depth = 12000.0, 300.0
    c.scale_factor, c.max_level, c.max_features = 1.2, 8, 1000
    c.odo_noise[:] = [2.0, 2.0, 0.002]
    c.odo_uncertain[:] = [0.01, 0.01, 0.01]
    c.planemotion_z_info, c.planemotion_xrot_info, c.planemotion_yrot_info = 1.0, 1e6, 1e6    # src/Config.cpp:46-48
    c.th_huber2, c.local_iter, c.fps = 5.991, 10, 30
    return c


def true_pose(t: float) -> np.ndarray:
    return np.array([SHIFT[0] * t * Z0 / FX, SHIFT[1] * t * Z0 / FX, 0.0])


def odometry(n: int, seed: int = 20190520, sigma_xy: float = 2.0) -> np.ndarray:
    """odometry readings of the n frames: the true body pose + N(0, sigma_xy^2) mm on x, y (frame 0 exact), float32 like Se2"""
    rng = np.random.default_rng(seed)
    odo = np.stack([true_pose(t) for t in range(n)])
    odo[1:, :2] += rng.normal(0.0, sigma_xy, (n - 1, 2))
    return odo.astype(np.float32)


def can_build() -> bool:
    return os.path.isdir(os.path.join(REFERENCE, "src"))


def build() -> None:
    """both libraries, where /root/reference is (the drop-in one needs se2lam_amd/lib/libse2gpu.so built first)"""
    if can_build():
        subprocess.check_call(["make", "-C", HERE, "pipeline", "REF=" + REFERENCE], stdout=subprocess.DEVNULL)


def available(kind: str) -> bool:
    return os.path.exists(LIBS[kind])


def lib(kind: str) -> C.CDLL:
    if kind not in _libs:
        if not os.path.exists(LIBS[kind]):
            build()
        if not os.path.exists(LIBS[kind]):
            raise FileNotFoundError(LIBS[kind] + " is absent and /root/reference is not here to build it from")
        L = C.CDLL(LIBS[kind])
        L.ref_pipe_create.restype = C.c_void_p
        L.ref_pipe_create.argtypes = [C.POINTER(Config)]
        L.ref_pipe_destroy.argtypes = [C.c_void_p]
        L.ref_pipe_last_error.restype = C.c_char_p
        L.ref_pipe_kind.restype = C.c_char_p
        L.ref_pipe_feed.restype = C.c_int
        L.ref_pipe_feed.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.POINTER(FrameRecord), C.c_void_p, C.c_void_p, C.c_int]
        L.ref_pipe_keyframes.restype = C.c_int
        L.ref_pipe_keyframes.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 5
        L.ref_pipe_mappoints.restype = C.c_int
        L.ref_pipe_mappoints.argtypes = [C.c_void_p, C.c_int] + [C.c_void_p] * 4
        L.ref_pipe_shim_calls.argtypes = [C.c_void_p]
        L.ref_pipe_keyframe_addresses.restype = C.c_int
        L.ref_pipe_keyframe_addresses.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.ref_pipe_observations.restype = C.c_int
        L.ref_pipe_observations.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_void_p]
        _libs[kind] = L
    return _libs[kind]


class Pipeline:
    """the reference's Track + LocalMapper + Map of one run; kind = 'cpu' (reference CPU build) or 'dropin' (over libse2gpu)"""

    def __init__(self, kind: str, config: Config | None = None, raw_matches: bool = True):
        self.kind, self.L, self.raw = kind, lib(kind), raw_matches
        self.config = config or default_config()
        self.h = self.L.ref_pipe_create(C.byref(self.config))
        if not self.h:
            raise RuntimeError("ref_pipe_create (%s): %s" % (kind, self.L.ref_pipe_last_error().decode()))

    def close(self):
        if self.h:
            self.L.ref_pipe_destroy(self.h)
            self.h = None

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    def name(self) -> str:
        return self.L.ref_pipe_kind().decode()

    def feed(self, img: np.ndarray, odo) -> dict:
        img = np.ascontiguousarray(img, np.uint8)
        odo = np.ascontiguousarray(odo, np.float32)
        cap = 4 * self.config.max_features
        rec = FrameRecord()
        mi, raw = np.full(cap, -2, np.int32), np.full(cap, -2, np.int32)
        rc = self.L.ref_pipe_feed(self.h, img.ctypes.data, img.shape[0], img.shape[1], odo.ctypes.data, C.byref(rec), mi.ctypes.data,
                                  raw.ctypes.data if self.raw else None, cap)
        if rc != 0:
            raise RuntimeError("ref_pipe_feed (%s): %s" % (self.kind, self.L.ref_pipe_last_error().decode()))
        out = {n: getattr(rec, n) for n, _ in FrameRecord._fields_ if n not in ("ba", "Twb", "Tcw")}
        out["ba"] = np.array(rec.ba[:])
        out["Twb"] = np.array(rec.Twb[:], np.float32)
        out["Tcw"] = np.array(rec.Tcw[:], np.float32).reshape(4, 4)
        out["match_idx"] = mi[:rec.n_match_entries].copy()
        out["raw_matches"] = raw[:rec.n_raw_entries].copy()
        return out

    def keyframes(self, cap: int = 4096) -> dict:
        ids, idkf, nobs = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        twb, tcw = np.zeros((cap, 3), np.float32), np.zeros((cap, 4, 4), np.float32)
        n = self.L.ref_pipe_keyframes(self.h, cap, ids.ctypes.data, idkf.ctypes.data, twb.ctypes.data, tcw.ctypes.data, nobs.ctypes.data)
        assert n <= cap
        return dict(id=ids[:n], id_kf=idkf[:n], Twb=twb[:n], Tcw=tcw[:n], n_obs=nobs[:n])

    def mappoints(self, cap: int = 1 << 18) -> dict:
        ids, nobs, good = np.zeros(cap, np.int32), np.zeros(cap, np.int32), np.zeros(cap, np.uint8)
        pos = np.zeros((cap, 3), np.float32)
        n = self.L.ref_pipe_mappoints(self.h, cap, ids.ctypes.data, pos.ctypes.data, nobs.ctypes.data, good.ctypes.data)
        assert n <= cap
        return dict(id=ids[:n], pos=pos[:n], n_obs=nobs[:n], good_prl=good[:n])

    def keyframe_address_rank(self, cap: int = 4096) -> np.ndarray:
        """rank of every key frame's heap address among the key frames (what std::map<PtrKeyFrame, int> orders a map point's
        observations by - see ref_pipe_keyframe_addresses)"""
        a = np.zeros(cap, np.uint64)
        n = self.L.ref_pipe_keyframe_addresses(self.h, cap, a.ctypes.data)
        assert 0 <= n <= cap
        return np.argsort(np.argsort(a[:n])).astype(np.int32)

    def observations(self, kf_pos: int, cap: int = 4096):
        """(feature indices, map-point ids) of the key frame at position kf_pos of keyframes()"""
        ftr, mp = np.zeros(cap, np.int32), np.zeros(cap, np.int32)
        n = self.L.ref_pipe_observations(self.h, kf_pos, cap, ftr.ctypes.data, mp.ctypes.data)
        assert 0 <= n <= cap
        return ftr[:n].copy(), mp[:n].copy()

    def shim_calls(self) -> dict:
        c = (C.c_longlong * 4)()
        self.L.ref_pipe_shim_calls(c)
        return dict(FAST=c[0], resize=c[1], GaussianBlur=c[2], findFundamentalMat=c[3])


def run(kind: str, frames, odo, config: Config | None = None, raw_matches: bool = True) -> dict:
    """feeds the frames; returns the per-frame records, the map after every local BA and at the end, and the wall times"""
    with Pipeline(kind, config, raw_matches) as p:
        recs, after_ba = [], []
        for t in range(len(frames)):
            r = p.feed(frames[t], odo[t])
            recs.append(r)
            if r["local_ba"]:
                after_ba.append(dict(frame=t, kfs=p.keyframes(), mps=p.mappoints()))
        return dict(kind=p.name(), frames=recs, after_ba=after_ba, kfs=p.keyframes(), mps=p.mappoints(), shim_calls=p.shim_calls(),
                    ms_track=float(sum(r["ms_track"] for r in recs)), ms_mapper=float(sum(r["ms_mapper"] for r in recs)))
