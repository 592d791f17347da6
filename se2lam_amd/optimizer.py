"""Python mirror of the reference's optimiser call surface over the C ABI.

Names and argument meaning follow /root/reference/include/se2lam/optimizer.h:78-141 and the
g2o::SparseOptimizer methods LocalMapper::localBA uses (/root/reference/src/LocalMapper.cpp:239-260)
so the parity tests read like the reference's own call sites (Map.cpp:891-1053).  The C++ twin of
this file is include/se2lam_amd/optimizer.h.  All compute happens in libse2gpu.so (HIP).
"""
from __future__ import annotations

import ctypes as C
import os

import numpy as np

from . import capi

LM, GN = 0, 1


def _stats_dict(st):
    n = min(st.iterations, 64)
    return dict(iterations=st.iterations, trials=st.trials, terminated=bool(st.terminated),
                stopped=bool(st.stopped), chi2_init=st.chi2_init, chi2_final=st.chi2_final,
                lambda_final=st.lambda_final, chi2_hist=list(st.chi2_hist[:n]),
                lambda_hist=list(st.lambda_hist[:n]), trials_hist=list(st.trials_hist[:n]))


def _no_fallback(h):
    """This harness (tests, bench) treats the library's solver fallback as a failure: results would still be right, but a
    dataflow time-out outside the fault-injection test is a bug that the fallback must not hide."""
    if capi.lib().se2gpu_ba_debug_solver_path(h) == 2 and os.environ.get("SE2GPU_BA_CHOL_FAULT") != "1":
        raise RuntimeError("se2gpu_ba: k_chol_tiles timed out and the handle fell back to k_chol_step")


def reset_estimates_batch(optimizers):
    """se2gpu_ba_reset_estimates_batch: every optimizer back to the estimates it was initialised with, one launch"""
    n = len(optimizers)
    hs = (C.c_void_p * n)(*[o._h for o in optimizers])
    capi.check(capi.lib().se2gpu_ba_reset_estimates_batch(hs, n))


def optimize_batch(optimizers, iterations: int, mode: int = 0, stop=None):
    """se2gpu_ba_optimize_batch: optimize(iterations) of several initialised SlamOptimizers at once (independent
    windows, concurrently on the device).  Fills every optimizer's .stats; returns the iteration counts."""
    n = len(optimizers)
    hs = (C.c_void_p * n)(*[o._h for o in optimizers])
    st = (capi.BaStats * n)()
    sp = stop.ctypes.data_as(C.POINTER(C.c_uint8)) if stop is not None else None
    capi.check(capi.lib().se2gpu_ba_optimize_batch(hs, n, int(iterations), int(mode), sp, st))
    # (the records become dictionaries when somebody reads them: 256 of them cost this harness 0.4 ms per batch, inside the
    # bench's timed region; the resident path - last_batch_path 2 - has no dataflow solve whose fallback could be hidden)
    resident = capi.lib().se2gpu_ba_last_batch_path() == 2
    for o, s in zip(optimizers, st):
        o._stats, o._stats_raw = None, s
        if not resident:
            _no_fallback(o._h)
    return [s.iterations for s in st]


class SlamOptimizer:
    """g2o::SparseOptimizer with SlamAlgorithm = Levenberg, BlockSolverX, dense pose solve."""

    def __init__(self):
        self._h = C.c_void_p()
        capi.check(capi.lib().se2gpu_ba_create(C.byref(self._h)))
        self._stop = None
        self._verbose = False
        self._cb = None
        self._keep = []
        self._stats = None
        self._stats_raw = None

    @property
    def stats(self):
        """the last run's se2gpu_ba_stats as a dictionary (made on first use)"""
        if self._stats is None and self._stats_raw is not None:
            self._stats = _stats_dict(self._stats_raw)
        return self._stats

    @stats.setter
    def stats(self, value):
        self._stats, self._stats_raw = value, None

    def stat(self, name: str):
        """one scalar field of the last run's record (iterations, trials, chi2_final ...) without building the dictionary"""
        return getattr(self._stats_raw, name) if self._stats_raw is not None else self.stats[name]

    # -- SparseOptimizer methods -----------------------------------------------------------
    def setVerbose(self, v: bool):
        self._verbose = bool(v)

    def setForceStopFlag(self, flag: np.ndarray | None):
        """flag: a 1-element uint8 numpy array polled between LM trials (bool* in the reference)."""
        self._stop = flag

    def clear(self):
        capi.check(capi.lib().se2gpu_ba_clear(self._h))

    def clearParameters(self):
        """g2o drops the camera parameters here, not in clear(); this harness re-adds the camera with every load()"""

    def initializeOptimization(self, level: int = 0):
        assert level == 0
        capi.check(capi.lib().se2gpu_ba_initialize(self._h))

    def optimize(self, iterations: int, mode: int = LM) -> int:
        st = capi.BaStats()
        stop = self._stop.ctypes.data_as(C.POINTER(C.c_uint8)) if self._stop is not None else None
        capi.check(capi.lib().se2gpu_ba_optimize(self._h, int(iterations), int(mode), stop, int(self._verbose),
                                                 C.byref(st)))
        self.stats = _stats_dict(st)
        _no_fallback(self._h)
        return st.iterations

    def exchange_doubles(self) -> int:
        """se2gpu_ba_exchange_doubles_h: size of the packed system exchange of this (initialised) handle"""
        return int(capi.lib().se2gpu_ba_exchange_doubles_h(self._h))

    def chol_verify(self):
        """se2gpu_ba_debug_chol_verify (SE2GPU_BA_CHOL_VERIFY=1): (mismatches, half-slabs checked, records (k, 8) uint64)"""
        counts = np.zeros(2, np.uint64)
        rec = np.zeros((64, 8), np.uint64)
        capi.check(capi.lib().se2gpu_ba_debug_chol_verify(self._h, counts.ctypes.data, rec.ctypes.data, 64))
        return int(counts[0]), int(counts[1]), rec[:min(int(counts[0]), 64)]

    def solver_path(self) -> int:
        """se2gpu_ba_debug_solver_path: 0 dataflow, 1 column launches (configured), 2 column launches (fallback), 3 host"""
        return int(capi.lib().se2gpu_ba_debug_solver_path(self._h))

    def activeRobustChi2(self) -> float:
        v = capi.lib().se2gpu_ba_chi2(self._h)
        if v < 0:
            capi.check(capi.ERR_STATE)
        return float(v)

    # -- harness extras ----------------------------------------------------------------------
    def load(self, g):
        """Bulk se2gpu_ba_load of a synth.BAGraph (ids: poses 0..P-1, landmarks P..P+L-1)."""
        l = capi.lib()
        capi.check(l.se2gpu_ba_add_cam(self._h, g.fx, g.cx, g.cy))
        R = np.ascontiguousarray(g.Rbc, np.float64).reshape(-1)
        t = np.ascontiguousarray(g.tbc, np.float64).reshape(-1)
        capi.check(l.se2gpu_ba_set_Tbc(self._h, capi.pd(R), capi.pd(t)))
        a = [np.ascontiguousarray(g.poses, np.float64), np.ascontiguousarray(g.fixed, np.uint8),
             np.ascontiguousarray(g.lms, np.float64), np.ascontiguousarray(g.e_kf, np.int32),
             np.ascontiguousarray(g.e_lm, np.int32), np.ascontiguousarray(g.e_uv, np.float64),
             np.ascontiguousarray(g.e_info, np.float64), np.ascontiguousarray(g.o_i, np.int32),
             np.ascontiguousarray(g.o_j, np.int32), np.ascontiguousarray(g.o_meas, np.float64),
             np.ascontiguousarray(g.o_info, np.float64)]
        P32, PU8 = C.POINTER(C.c_int32), C.POINTER(C.c_uint8)
        capi.check(l.se2gpu_ba_load(self._h, g.P, g.L, g.E, g.O, capi.pd(a[0]), a[1].ctypes.data_as(PU8),
                                    capi.pd(a[2]), a[3].ctypes.data_as(P32), a[4].ctypes.data_as(P32),
                                    capi.pd(a[5]), capi.pd(a[6]), a[7].ctypes.data_as(P32),
                                    a[8].ctypes.data_as(P32), capi.pd(a[9]), capi.pd(a[10]), float(g.huber)))
        self._shape = (g.P, g.L)
        self._keep = a          # the edge arrays are borrowed by the library until initializeOptimization

    def estimates(self):
        P, L = self._shape
        poses = np.zeros((P, getattr(self, "_pose_dim", 3)))
        lms = np.zeros((max(L, 1), 3))
        capi.check(capi.lib().se2gpu_ba_get_all(self._h, capi.pd(poses), capi.pd(lms)))
        return poses, lms[:L]

    def reset_estimates(self):
        capi.check(capi.lib().se2gpu_ba_reset_estimates(self._h))

    def reduced_system(self, lam: float):
        P, _ = self._shape
        n = (6 if getattr(self, "_pose_dim", 3) == 12 else 3) * P
        S = np.zeros((n, n))
        bs = np.zeros(n)
        capi.check(capi.lib().se2gpu_ba_debug_reduced_system(self._h, float(lam), capi.pd(S), capi.pd(bs)))
        return S, bs

    def solve(self, lam: float):
        """x with S(lam) x = bs by the device factorisation; returns (x, factor_ok)."""
        P, _ = self._shape
        x = np.zeros(3 * P)
        ok = C.c_int(0)
        capi.check(capi.lib().se2gpu_ba_debug_solve(self._h, float(lam), capi.pd(x), C.byref(ok)))
        return x, bool(ok.value)

    def set_shard(self, rank: int, world: int):
        capi.check(capi.lib().se2gpu_ba_set_shard(self._h, rank, world))

    def set_allreduce(self, fn, buffer_ptr=None):
        """fn(dev_ptr:int, count:int, stream:int) -> None must sum `count` doubles in place over ranks."""
        def _tramp(ptr, count, stream, user):
            try:
                fn(ptr, count, stream)
                return 0
            except Exception as exc:  # never unwind through the C ABI
                print(f"se2lam_amd: all-reduce callback failed: {exc!r}")
                return 1
        self._cb = capi.ALLREDUCE_FN(_tramp)
        capi.check(capi.lib().se2gpu_ba_set_allreduce(self._h, self._cb, None, buffer_ptr))

    def set_comm(self, comm):
        """comm: capi-level se2gpu_comm* (c_void_p) from se2gpu_comm_create - native RCCL all-reduce"""
        capi.check(capi.lib().se2gpu_ba_set_comm(self._h, comm))

    def reduce_buffer_doubles(self, P: int) -> int:
        return int(capi.lib().se2gpu_ba_reduce_buffer_doubles(self._h, P))

    def set_stream(self, stream_ptr):
        capi.check(capi.lib().se2gpu_ba_set_stream(self._h, stream_ptr))

    def stream(self):
        return capi.lib().se2gpu_ba_stream(self._h)

    def profile(self, enable: bool):
        capi.check(capi.lib().se2gpu_ba_profile(self._h, int(enable)))

    def profile_report(self):
        out = {}
        i = 0
        while True:
            name = C.c_char_p()
            ms = C.c_double()
            n = C.c_int64()
            if capi.lib().se2gpu_ba_profile_get(self._h, i, C.byref(name), C.byref(ms), C.byref(n)) != 0:
                break
            out[name.value.decode()] = (ms.value, n.value)
            i += 1
        return out

    def __del__(self):
        try:
            if self._h:
                capi.lib().se2gpu_ba_destroy(self._h)
                self._h = None
        except Exception:
            pass


# -- free functions of optimizer.h ------------------------------------------------------------
def addCamPara(opt: SlamOptimizer, K: np.ndarray, id: int = 0):
    """optimizer.h:85 / optimizer.cpp:207-215: single focal length K(0,0), principal point K(0,2),K(1,2)."""
    K = np.asarray(K, dtype=np.float32)
    capi.check(capi.lib().se2gpu_ba_add_cam(opt._h, float(K[0, 0]), float(K[0, 2]), float(K[1, 2])))


def setExtParameter(opt: SlamOptimizer, Rbc, tbc):
    """EdgeSE2XYZ::setExtParameter(Tbc) (EdgeSE2XYZ.h:53) - one extrinsic for the whole graph."""
    R = np.ascontiguousarray(Rbc, np.float64).reshape(-1)
    t = np.ascontiguousarray(tbc, np.float64).reshape(-1)
    capi.check(capi.lib().se2gpu_ba_set_Tbc(opt._h, capi.pd(R), capi.pd(t)))


def addVertexSE2(opt: SlamOptimizer, pose, id: int, fixed: bool = False):
    """optimizer.h:104"""
    capi.check(capi.lib().se2gpu_ba_add_vertex_se2(opt._h, int(id), float(pose[0]), float(pose[1]), float(pose[2]),
                                                   int(bool(fixed))))


def addVertexSBAXYZ(opt: SlamOptimizer, xyz, id: int, marginal: bool = True, fixed: bool = False):
    """optimizer.h:91"""
    a = np.ascontiguousarray(xyz, np.float64)
    capi.check(capi.lib().se2gpu_ba_add_vertex_xyz(opt._h, int(id), capi.pd(a), int(marginal), int(fixed)))


def addEdgeSE2XYZ(opt: SlamOptimizer, meas, id0: int, id1: int, info, thHuber: float):
    """optimizer.h:100 (campara / Tbc are graph-wide: addCamPara, setExtParameter)."""
    m = np.ascontiguousarray(meas, np.float64)
    w = np.ascontiguousarray(info, np.float64).reshape(-1)
    assert w.size == 4
    capi.check(capi.lib().se2gpu_ba_add_edge_se2xyz(opt._h, int(id0), int(id1), capi.pd(m), capi.pd(w), float(thHuber)))


def addEdgeSE2(opt: SlamOptimizer, meas, id0: int, id1: int, info):
    """optimizer.h:109"""
    m = np.ascontiguousarray(meas, np.float64)
    w = np.ascontiguousarray(info, np.float64).reshape(-1)
    assert w.size == 9
    capi.check(capi.lib().se2gpu_ba_add_edge_se2(opt._h, int(id0), int(id1), capi.pd(m), capi.pd(w)))


def estimateVertexSE2(opt: SlamOptimizer, id: int) -> np.ndarray:
    """optimizer.h:107"""
    out = np.zeros(3)
    capi.check(capi.lib().se2gpu_ba_get_se2(opt._h, int(id), capi.pd(out)))
    return out


def estimateVertexSBAXYZ(opt: SlamOptimizer, id: int) -> np.ndarray:
    """optimizer.h:141"""
    out = np.zeros(3)
    capi.check(capi.lib().se2gpu_ba_get_xyz(opt._h, int(id), capi.pd(out)))
    return out


def loadLocalGraph(opt: SlamOptimizer, *, kf_id, kf_Twb, kf_Rcw, n_local, odo_to, odo_meas, odo_cov, mp_pos, obs_mp, obs_kf,
                   obs_uv, obs_lc, obs_sigma2, K, Rbc, tbc, huber, xrot_info=1e6, z_info=1.0):
    """Map::loadLocalGraph(optimizer) (Map.cpp:891-1022) through se2gpu_ba_load_local_graph: the POD view of the local
    window (key frames: local first, then reference; observations grouped by map point)."""
    keep = [np.ascontiguousarray(kf_id, np.int32), np.ascontiguousarray(kf_Twb, np.float32),
            np.ascontiguousarray(kf_Rcw, np.float32), np.ascontiguousarray(odo_to, np.int32),
            np.ascontiguousarray(odo_meas, np.float64), np.ascontiguousarray(odo_cov, np.float64),
            np.ascontiguousarray(mp_pos, np.float32), np.ascontiguousarray(obs_mp, np.int32),
            np.ascontiguousarray(obs_kf, np.int32), np.ascontiguousarray(obs_uv, np.float32),
            np.ascontiguousarray(obs_lc, np.float32), np.ascontiguousarray(obs_sigma2, np.float32)]
    g = capi.LocalGraph()
    g.n_local_kf, g.n_ref_kf = int(n_local), int(len(keep[0]) - n_local)
    g.n_mp, g.n_obs = int(len(keep[6])), int(len(keep[7]))
    (g.kf_id, g.kf_Twb, g.kf_Rcw, g.odo_to, g.odo_meas, g.odo_cov, g.mp_pos, g.obs_mp, g.obs_kf, g.obs_uv, g.obs_lc,
     g.obs_sigma2) = [a.ctypes.data for a in keep]
    K = np.asarray(K, np.float32)
    g.fx, g.cx, g.cy = float(K[0, 0]), float(K[0, 2]), float(K[1, 2])
    g.Rbc = (C.c_double * 9)(*np.asarray(Rbc, np.float64).reshape(-1))
    g.tbc = (C.c_double * 3)(*np.asarray(tbc, np.float64).reshape(-1))
    g.huber_delta, g.xrot_info, g.z_info = float(huber), float(xrot_info), float(z_info)
    capi.check(capi.lib().se2gpu_ba_load_local_graph(opt._h, C.byref(g)))
    opt._shape = (len(keep[0]), len(keep[6]))


# -- SE3-expmap graphs (optimizer.h:82-98, 138): Map::loadLocalGraph(optimizer, vpEdgesAll, vnAllIdx) ------------------------
def _pose12(T):
    T = np.asarray(T, np.float64)
    return np.ascontiguousarray(np.concatenate([T[:3, :3].reshape(-1), T[:3, 3]]))


def _pose44(p):
    T = np.eye(4)
    T[:3, :3] = np.asarray(p[:9]).reshape(3, 3)
    T[:3, 3] = p[9:12]
    return T


def addVertexSE3Expmap(opt: SlamOptimizer, Tcw, id: int, fixed: bool = False):
    """optimizer.h:88: Tcw 4x4"""
    capi.check(capi.lib().se2gpu_ba_add_vertex_se3(opt._h, int(id), capi.pd(_pose12(Tcw)), int(bool(fixed))))


def addPlaneMotionSE3Expmap(opt: SlamOptimizer, Tcw, vId: int, extPara, xrot_info=1e6, yrot_info=1e6, z_info=1.0):
    """optimizer.h:82 / optimizer.cpp:236-314: the EdgeSE3ExpmapPrior that keeps key frame vId on the plane (extPara = Config::bTc)."""
    meas = np.zeros(12)
    info = np.zeros(36)
    a, b = _pose12(Tcw), _pose12(extPara)      # named: a temporary's buffer is gone before the call
    capi.check(capi.lib().se2gpu_plane_motion_prior(a.ctypes.data, b.ctypes.data, float(xrot_info),
                                                    float(yrot_info), float(z_info), meas.ctypes.data, info.ctypes.data))
    capi.check(capi.lib().se2gpu_ba_add_prior_se3(opt._h, int(vId), capi.pd(meas), capi.pd(info)))


def addPriorSE3Expmap(opt: SlamOptimizer, vId: int, meas, info):
    capi.check(capi.lib().se2gpu_ba_add_prior_se3(opt._h, int(vId), capi.pd(_pose12(meas)),
                                                  capi.pd(np.ascontiguousarray(info, np.float64).reshape(-1))))


def _add_edge_se3(opt: SlamOptimizer, measure, id0: int, id1: int, info):
    capi.check(capi.lib().se2gpu_ba_add_edge_se3(opt._h, int(id0), int(id1), capi.pd(_pose12(measure)),
                                                 capi.pd(np.ascontiguousarray(info, np.float64).reshape(-1))))


def swapInfoBlocks(info) -> np.ndarray:
    """(translation, rotation) <-> (rotation, translation) order of a 6x6 information matrix (optimizer.cpp:492-497)"""
    w = np.asarray(info, np.float64).reshape(6, 6)
    n = np.empty((6, 6))
    n[:3, :3] = w[3:, 3:]; n[3:, :3] = w[:3, 3:]; n[:3, 3:] = w[3:, :3]; n[3:, 3:] = w[:3, :3]
    return n


def addEdgeSE3Expmap(opt: SlamOptimizer, measure, id0: int, id1: int, info):
    """optimizer.h:94 / optimizer.cpp:482-500: "The input info is [trans rot] order, but EdgeSE3Expmap requires [rot trans]" -
    the blocks are swapped here as the reference swaps them; the C ABI takes g2o's order."""
    _add_edge_se3(opt, measure, id0, id1, swapInfoBlocks(info))


def addEdgeXYZ2UV(opt: SlamOptimizer, measure, idMP: int, idKF: int, paraId: int, info, thHuber: float):
    """optimizer.h:97: info = invSigma2 * I (Map.cpp:527)"""
    w = np.asarray(info, np.float64)
    inv_sigma2 = float(w.reshape(-1)[0])
    capi.check(capi.lib().se2gpu_ba_add_edge_xyz2uv(opt._h, int(idMP), int(idKF), capi.pd(np.ascontiguousarray(measure, np.float64)),
                                                    inv_sigma2, float(thHuber)))


def estimateVertexSE3Expmap(opt: SlamOptimizer, id: int) -> np.ndarray:
    """optimizer.h:138 -> Tcw 4x4"""
    out = np.zeros(12)
    capi.check(capi.lib().se2gpu_ba_get_se3(opt._h, int(id), capi.pd(out)))
    return _pose44(out)


def edgeChi2(opt: SlamOptimizer, n_edges: int) -> np.ndarray:
    """chi2() of every EdgeProjectXYZ2UV at the current estimate, in the order the edges were added"""
    out = np.zeros(max(n_edges, 1))
    capi.check(capi.lib().se2gpu_ba_edge_chi2(opt._h, capi.pd(out), int(n_edges)))
    return out[:n_edges]


def load_se3_graph(opt: SlamOptimizer, g, K=None):
    """A synth.BA3Graph through the reference's call sequence (Map.cpp:414-566): ids as the reference numbers them."""
    K = np.array([[g.fx, 0, g.cx], [0, g.fx, g.cy], [0, 0, 1]], np.float32) if K is None else K
    addCamPara(opt, K, 0)
    for a in range(g.P):
        addVertexSE3Expmap(opt, g.poses[a], a, bool(g.fixed[a]))
        if g.has_prior[a]:
            addPriorSE3Expmap(opt, a, g.prior_meas[a], g.prior_info[a])
    for k in range(g.O):
        addEdgeSE3Expmap(opt, g.o_meas[k], int(g.o_i[k]), int(g.o_j[k]), swapInfoBlocks(g.o_info[k]))   # the graph holds g2o's order
    maxKFid = g.P + 1
    for l in range(g.L):
        addVertexSBAXYZ(opt, g.lms[l], maxKFid + l)
    for k in range(g.E):
        addEdgeXYZ2UV(opt, g.e_uv[k], maxKFid + int(g.e_lm[k]), int(g.e_kf[k]), 0, np.eye(2) * g.e_w[k], g.huber)
    opt._shape = (g.P, g.L)
    opt._pose_dim = 12
    return maxKFid


# -- pose graphs (optimizer.h:117-135): GlobalMapper::GlobalBA -------------------------------------------------------------
def addVertexSE3(opt: SlamOptimizer, Twc, id: int, fixed: bool = False):
    """optimizer.h:120: g2o::VertexSE3 with the Isometry3D T_w_c"""
    capi.check(capi.lib().se2gpu_ba_add_vertex_iso3(opt._h, int(id), capi.pd(_pose12(Twc)), int(bool(fixed))))


def addVertexSE3PlaneMotion(opt: SlamOptimizer, Twc, id: int, extPara, paraSE3OffsetId: int = 0, fixed: bool = False,
                            xrot_info=1e6, yrot_info=1e6, z_info=1.0):
    """optimizer.h:123 / optimizer.cpp:336-470: the vertex plus its EdgeSE3Prior (extPara = Config::bTc)"""
    addVertexSE3(opt, Twc, id, fixed)
    meas = np.zeros(12)
    info = np.zeros(36)
    a, b = _pose12(Twc), _pose12(extPara)      # named: a temporary's buffer is gone before the call
    capi.check(capi.lib().se2gpu_plane_motion_prior_iso3(a.ctypes.data, b.ctypes.data,
                                                         float(xrot_info), float(yrot_info), float(z_info), meas.ctypes.data,
                                                         info.ctypes.data))
    capi.check(capi.lib().se2gpu_ba_add_prior_se3(opt._h, int(id), capi.pd(meas), capi.pd(info)))
    return _pose44(meas), info.reshape(6, 6)


def addEdgeSE3(opt: SlamOptimizer, measure, id0: int, id1: int, info):
    """optimizer.h:129: EdgeSE3 takes (translation, rotation) as it is"""
    _add_edge_se3(opt, measure, id0, id1, info)


estimateVertexSE3 = estimateVertexSE3Expmap     # optimizer.h:135 (the pose type follows the graph)


def load_pose_graph(opt: SlamOptimizer, g):
    """A synth.PoseGraph through GlobalMapper::GlobalBA's call sequence (GlobalMapper.cpp:352-412)."""
    for a in range(g.P):
        addVertexSE3(opt, g.poses[a], a, bool(g.fixed[a]))
        if g.has_prior[a]:
            addPriorSE3Expmap(opt, a, g.prior_meas[a], g.prior_info[a])
    for k in range(g.O):
        addEdgeSE3(opt, g.o_meas[k], int(g.o_i[k]), int(g.o_j[k]), g.o_info[k])
    opt._shape = (g.P, 0)
    opt._pose_dim = 12


def shard_landmarks(L: int, e_kf: np.ndarray, e_lm: np.ndarray, world: int) -> np.ndarray:
    """Host-side landmark partition of the library (no device needed)."""
    e_kf = np.ascontiguousarray(e_kf, np.int32)
    e_lm = np.ascontiguousarray(e_lm, np.int32)
    owner = np.zeros(L, np.int32)
    P32 = C.POINTER(C.c_int32)
    capi.check(capi.lib().se2gpu_ba_shard_landmarks(int(L), int(e_kf.size), e_kf.ctypes.data_as(P32),
                                                     e_lm.ctypes.data_as(P32), int(world), owner.ctypes.data_as(P32)))
    return owner


def edge_information(lc, lw, e_kf, sigma2, Rcw, twb_xy, fx, xrot_info=1e6, z_info=1.0) -> np.ndarray:
    """Map::loadLocalGraph's per-observation information (Map.cpp:1024-1049) on the device -> (E, 2, 2) float64."""
    lc = np.ascontiguousarray(lc, np.float32); lw = np.ascontiguousarray(lw, np.float32)
    e_kf = np.ascontiguousarray(e_kf, np.int32); sigma2 = np.ascontiguousarray(sigma2, np.float32)
    Rcw = np.ascontiguousarray(Rcw, np.float32).reshape(-1, 9); twb_xy = np.ascontiguousarray(twb_xy, np.float32)
    E, P = len(e_kf), len(Rcw)
    out = np.zeros((max(E, 1), 2, 2))
    capi.check(capi.lib().se2gpu_ba_edge_information(E, lc.ctypes.data, lw.ctypes.data, e_kf.ctypes.data,
                                                     sigma2.ctypes.data, P, Rcw.ctypes.data, twb_xy.ctypes.data,
                                                     float(fx), float(xrot_info), float(z_info), out.ctypes.data))
    return out[:E]
