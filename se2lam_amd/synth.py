"""Seeded synthetic inputs for the se2lam hot path.

Nothing here is on the product path: these generators only build inputs that
the tests, `bench.py` and `__graft_entry__.smoke()` feed to BOTH the oracle
and the HIP library.

* `texture()/frame()`      - 640x480 u8 crops of a random-rectangle texture
                             (seed 20190520) for the ORB extractor / matcher.
* `ba_graph(P, L, ...)`    - circular-trajectory room graph (seed 424242) in the
                             formulation of `Map::loadLocalGraph`
                             (/root/reference/src/Map.cpp:891-1053): SE(2) poses,
                             XYZ landmarks, EdgeSE2XYZ observations with the
                             plane-motion information of Map.cpp:1024-1049 and
                             PreEdgeSE2 odometry edges whose measurement and
                             covariance are pre-integrated exactly as
                             Track::updateFramePose (src/Track.cpp:162-188).
"""
from __future__ import annotations

import dataclasses
import functools

import numpy as np

ORB_SEED = 20190520
BA_SEED = 424242

IMG_W, IMG_H = 640, 480
FX = 400.0
CX, CY = 320.0, 240.0

# Camera (z fwd, x right, y down) mounted on body (x fwd, y left, z up); mm.
RBC = np.array([[0.0, 0.0, 1.0], [-1.0, 0.0, 0.0], [0.0, -1.0, 0.0]])
TBC = np.array([100.0, 0.0, 300.0])

TH_HUBER = float(np.sqrt(5.991))
PLANEMOTION_XROT_INFO = 1e6  # src/Config.cpp:46
PLANEMOTION_Z_INFO = 1.0  # src/Config.cpp:48
SCALE_FACTOR = np.float32(1.2)


# --------------------------------------------------------------------------
# ORB frames
# --------------------------------------------------------------------------
@functools.lru_cache(maxsize=2)
def texture(seed: int = ORB_SEED) -> np.ndarray:
    """1280x960 u8: 4000 random axis-aligned rectangles over mid-grey + N(0,3^2)."""
    rng = np.random.default_rng(seed)
    W, H = 1280, 960
    tex = np.full((H, W), 128.0, dtype=np.float64)
    n = 4000
    xs = rng.integers(0, W, n)
    ys = rng.integers(0, H, n)
    ws = rng.integers(4, 41, n)
    hs = rng.integers(4, 41, n)
    gs = rng.integers(0, 256, n)
    for x, y, w, h, g in zip(xs, ys, ws, hs, gs):
        tex[y:y + h, x:x + w] = g
    tex += rng.normal(0.0, 3.0, size=tex.shape)
    return np.clip(np.rint(tex), 0, 255).astype(np.uint8)


def frame(t: int, seed: int = ORB_SEED) -> np.ndarray:
    """Frame t = 640x480 crop at offset (100+3t, 80+t) (wrapped to stay inside)."""
    tex = texture(seed)
    ox = 100 + (3 * t) % 540
    oy = 80 + t % 400
    return np.ascontiguousarray(tex[oy:oy + IMG_H, ox:ox + IMG_W])


def frames(n: int, start: int = 0, seed: int = ORB_SEED) -> np.ndarray:
    return np.stack([frame(start + i, seed) for i in range(n)], axis=0)


# --------------------------------------------------------------------------
# BA graphs
# --------------------------------------------------------------------------
@dataclasses.dataclass
class BAGraph:
    """Flat (SoA) description of one local-BA window.

    poses      (P,3) f64  initial (x, y, theta) of each keyframe  [VertexSE2]
    fixed      (P,)  u8   1 = fixed vertex (Map.cpp:927,969)
    lms        (L,3) f64  initial landmark positions              [VertexSBAPointXYZ]
    e_kf,e_lm  (E,)  i32  vertices of each EdgeSE2XYZ
    e_uv       (E,2) f64  measurement
    e_info     (E,3) f64  information (xx, xy, yy) = Sigma_all^-1 (Map.cpp:1045-1049)
    o_i,o_j    (O,)  i32  PreEdgeSE2 vertices (this KF, next KF) (Map.cpp:942-953)
    o_meas     (O,3) f64
    o_info     (O,9) f64  row-major 3x3 = cov^-1
    """
    poses: np.ndarray
    fixed: np.ndarray
    lms: np.ndarray
    e_kf: np.ndarray
    e_lm: np.ndarray
    e_uv: np.ndarray
    e_info: np.ndarray
    o_i: np.ndarray
    o_j: np.ndarray
    o_meas: np.ndarray
    o_info: np.ndarray
    fx: float = FX
    cx: float = CX
    cy: float = CY
    Rbc: np.ndarray = dataclasses.field(default_factory=lambda: RBC.copy())
    tbc: np.ndarray = dataclasses.field(default_factory=lambda: TBC.copy())
    huber: float = TH_HUBER
    poses_true: np.ndarray | None = None
    lms_true: np.ndarray | None = None
    e_level: np.ndarray | None = None      # pyramid level of every observation (octave)

    @property
    def P(self) -> int:
        return int(self.poses.shape[0])

    @property
    def L(self) -> int:
        return int(self.lms.shape[0])

    @property
    def E(self) -> int:
        return int(self.e_kf.shape[0])

    @property
    def O(self) -> int:
        return int(self.o_i.shape[0])

    def algorithmic_bytes_per_iter(self) -> int:
        """B_ba of SURVEY.md §8(d): 48E + 48L + 48P + 4*3P*(3P+1) + 24P."""
        P, L, E = self.P, self.L, self.E
        return 48 * E + 48 * L + 48 * P + 4 * 3 * P * (3 * P + 1) + 24 * P

    def shard(self, rank: int, world: int) -> "BAGraph":
        """Landmark shard `rank` of `world`: landmarks are ordered by the
        lowest-id observing keyframe and split into `world` contiguous chunks of
        (nearly) equal EDGE count; poses are replicated; odometry edges go to rank 0."""
        if world == 1:
            return self
        owner = shard_landmarks(self.e_kf, self.e_lm, self.L, world)
        keep_lm = np.nonzero(owner == rank)[0]
        remap = -np.ones(self.L, dtype=np.int64)
        remap[keep_lm] = np.arange(keep_lm.size)
        keep_e = np.nonzero(owner[self.e_lm] == rank)[0]
        odo = slice(None) if rank == 0 else slice(0, 0)
        return dataclasses.replace(
            self,
            lms=self.lms[keep_lm].copy(),
            e_kf=self.e_kf[keep_e].copy(),
            e_lm=remap[self.e_lm[keep_e]].astype(np.int32),
            e_uv=self.e_uv[keep_e].copy(),
            e_info=self.e_info[keep_e].copy(),
            o_i=self.o_i[odo].copy(), o_j=self.o_j[odo].copy(),
            o_meas=self.o_meas[odo].copy(), o_info=self.o_info[odo].copy(),
            lms_true=None if self.lms_true is None else self.lms_true[keep_lm].copy(),
            e_level=None if self.e_level is None else self.e_level[keep_e].copy(),
        )


def _bagraph_shard_landmarks(self, rank: int, world: int) -> np.ndarray:
    """indices (into this graph's landmarks) of the landmarks shard `rank` holds, in the shard's order"""
    if world == 1:
        return np.arange(self.L)
    return np.nonzero(shard_landmarks(self.e_kf, self.e_lm, self.L, world) == rank)[0]


BAGraph.shard_landmarks = _bagraph_shard_landmarks


def shard_landmarks(e_kf: np.ndarray, e_lm: np.ndarray, L: int, world: int) -> np.ndarray:
    """owner[l] in [0, world): contiguous chunks, balanced by edge count, of the landmarks
    sorted by (lowest observing keyframe id, landmark id) - "sharded by keyframe window"."""
    first_kf = np.full(L, np.iinfo(np.int64).max, dtype=np.int64)
    np.minimum.at(first_kf, e_lm, e_kf.astype(np.int64))
    deg = np.bincount(e_lm, minlength=L)
    order = np.lexsort((np.arange(L), first_kf))
    csum = np.cumsum(deg[order])
    total = int(csum[-1]) if L else 0
    owner = np.zeros(L, dtype=np.int32)
    # landmark at sorted position k goes to the chunk its cumulative edge count falls in
    bounds = [(total * (r + 1)) // world for r in range(world)]
    chunk = np.searchsorted(np.asarray(bounds), csum, side="left")
    owner[order] = np.minimum(chunk, world - 1).astype(np.int32)
    return owner


def _rot2(th):
    c, s = np.cos(th), np.sin(th)
    return np.array([[c, -s], [s, c]])


def _norm_angle(a):
    return (a + np.pi) % (2 * np.pi) - np.pi


def _preintegrate(rng, pose_i, pose_j, nsub=10, sx=2.0, sy=2.0, st=0.002):
    """PreSE2 (meas, cov) between two keyframes, accumulated over `nsub` odometry frames
    exactly as Track::updateFramePose (src/Track.cpp:170-187), in double."""
    # true relative motion, split into nsub equal body-frame increments + noise
    dth = _norm_angle(pose_j[2] - pose_i[2])
    dxy = _rot2(pose_i[2]).T @ (pose_j[:2] - pose_i[:2])
    # constant-twist interpolation: recover the per-step increment by sampling the arc
    ths = pose_i[2] + dth * np.arange(nsub + 1) / nsub
    # positions along a straight chord in the i frame rotated progressively (good enough
    # as a *generator*: only consistency between meas and cov matters)
    pts = np.stack([dxy * k / nsub for k in range(nsub + 1)], axis=0)
    meas = np.zeros(3)
    cov = np.zeros((3, 3))
    Sv = np.diag([sx * sx, sy * sy, st * st])
    for k in range(nsub):
        # odometry increment expressed in the frame at step k (relative heading ths[k]-ths[0])
        Rk = _rot2(ths[k] - ths[0])
        odo_xy = Rk.T @ (pts[k + 1] - pts[k]) + rng.normal(0.0, [sx, sy])
        odo_th = (ths[k + 1] - ths[k]) + rng.normal(0.0, st)
        Phi = _rot2(meas[2])
        A = np.eye(3)
        B = np.eye(3)
        A[:2, 2] = Phi @ np.array([-odo_xy[1], odo_xy[0]])
        B[:2, :2] = Phi
        meas[:2] += Phi @ odo_xy
        meas[2] += odo_th
        cov = A @ cov @ A.T + B @ Sv @ B.T
    return meas, cov


def _project(poses, lms, kf, lm, Rcb, tcb, fx, cx, cy):
    """u,v,depth of landmark lm seen from keyframe kf (EdgeSE2XYZ.cpp:61-72 closed form)."""
    th = poses[kf, 2]
    c, s = np.cos(th), np.sin(th)
    d = lms[lm] - np.stack([poses[kf, 0], poses[kf, 1], np.zeros_like(th)], axis=-1)
    # Rz(-th) * d
    bx = c * d[..., 0] + s * d[..., 1]
    by = -s * d[..., 0] + c * d[..., 1]
    bz = d[..., 2]
    b = np.stack([bx, by, bz], axis=-1)
    lc = b @ Rcb.T + tcb
    z = lc[..., 2]
    with np.errstate(divide="ignore", invalid="ignore"):
        u = fx * lc[..., 0] / z + cx
        v = fx * lc[..., 1] / z + cy
    return u, v, z, lc


def edge_information(poses, lms, e_kf, e_lm, level, Rbc, tbc, fx):
    """Per-observation information of Map::loadLocalGraph (src/Map.cpp:1024-1049):
    Sigma = s_rot*J_r*J_r^T + s_z*J_z*J_z^T + sigma2_level*I, Omega = Sigma^-1.
    lc / Rcw go through float32 as in the reference (mViewMPs, Tcw are CV_32F)."""
    Rcb = Rbc.T
    tcb = -Rcb @ tbc
    _, _, _, lc = _project(poses, lms, e_kf, e_lm, Rcb, tcb, fx, 0.0, 0.0)
    lc = lc.astype(np.float32).astype(np.float64)
    th = poses[e_kf, 2]
    c, s = np.cos(th), np.sin(th)
    E = e_kf.shape[0]
    Rbw = np.zeros((E, 3, 3))
    Rbw[:, 0, 0] = c
    Rbw[:, 0, 1] = s
    Rbw[:, 1, 0] = -s
    Rbw[:, 1, 1] = c
    Rbw[:, 2, 2] = 1.0
    Rcw = (Rcb[None] @ Rbw).astype(np.float32).astype(np.float64)
    zi = 1.0 / lc[:, 2]
    zi2 = zi * zi
    fxf = float(np.float32(fx))
    Jpi = np.zeros((E, 2, 3))
    Jpi[:, 0, 0] = fxf * zi
    Jpi[:, 0, 2] = -fxf * lc[:, 0] * zi2
    Jpi[:, 1, 1] = fxf * zi
    Jpi[:, 1, 2] = -fxf * lc[:, 1] * zi2
    A = Jpi @ Rcw  # (E,2,3)
    # lw from float MP position, pi from float Twb
    lw = lms[e_lm].astype(np.float32).astype(np.float64)
    pi = np.zeros((E, 3))
    pi[:, 0] = poses[e_kf, 0].astype(np.float32)
    pi[:, 1] = poses[e_kf, 1].astype(np.float32)
    d = lw - pi
    sk = np.zeros((E, 3, 3))
    sk[:, 0, 1] = -d[:, 2]
    sk[:, 0, 2] = d[:, 1]
    sk[:, 1, 0] = d[:, 2]
    sk[:, 1, 2] = -d[:, 0]
    sk[:, 2, 0] = -d[:, 1]
    sk[:, 2, 1] = d[:, 0]
    Jr = (A @ sk)[:, :, :2]
    Jz = -A[:, :, 2:3]
    s_rot = float(np.float32(1.0 / PLANEMOTION_XROT_INFO))
    s_z = float(np.float32(1.0 / PLANEMOTION_Z_INFO))
    # mvLevelSigma2[octave] (float chain, src/Frame.cpp:50-58)
    sf = np.ones(8, dtype=np.float32)
    for i in range(1, 8):
        sf[i] = sf[i - 1] * SCALE_FACTOR
    sig2 = (sf * sf).astype(np.float32)
    sig2[0] = 1.0
    Sigma = s_rot * (Jr @ Jr.transpose(0, 2, 1)) + s_z * (Jz @ Jz.transpose(0, 2, 1))
    Sigma[:, 0, 0] += sig2[level]
    Sigma[:, 1, 1] += sig2[level]
    info = np.linalg.inv(Sigma)
    return np.stack([info[:, 0, 0], 0.5 * (info[:, 0, 1] + info[:, 1, 0]), info[:, 1, 1]], axis=1)


@functools.lru_cache(maxsize=4)
def ba_graph(P: int = 50, L: int = 5000, obs_per_lm: float = 6.0, seed: int = BA_SEED) -> BAGraph:
    """Room 20 m x 20 m x 3 m, (almost) closed circle radius 6 m, P keyframes, L landmarks on the
    walls / ceiling, ~obs_per_lm observations per landmark (config 3: P=50, L=5000;
    config 4: P=200, L=20000).  Units: mm."""
    rng = np.random.default_rng(seed + 1000003 * P + L)
    Rcb = RBC.T
    tcb = -Rcb @ TBC
    # --- truth
    # Headings sweep (-pi+0.1, pi-0.1): an almost-closed circle (348.5 deg).  The gap keeps every
    # heading away from +-pi, where the reference's PreEdgeSE2 error `aj - ai - z` has no angle
    # wrap (EdgeSE2XYZ.h:80) and VertexSE2::oplus re-normalises theta - a crossing would inject a
    # spurious 2*pi odometry residual into the benchmark graph.
    head = -np.pi + 0.1 + (2 * np.pi - 0.2) * np.arange(P) / max(P - 1, 1)
    ang = head - np.pi / 2
    poses_t = np.stack([6000.0 * np.cos(ang), 6000.0 * np.sin(ang), head], axis=1)

    def sample_lms(n):
        face = rng.integers(0, 5, n)  # 4 walls + ceiling
        a = rng.uniform(-10000.0, 10000.0, n)
        b = rng.uniform(-10000.0, 10000.0, n)
        z = rng.uniform(0.0, 3000.0, n)
        # faces 0/1: (+-10000, a, z) ; faces 2/3: (a, +-10000, z) ; face 4: (a, b, 3000)
        x = np.select([face == 0, face == 1], [10000.0, -10000.0], default=a)
        y = np.select([face == 2, face == 3, face == 4], [10000.0, -10000.0, b], default=a)
        zz = np.where(face == 4, 3000.0, z)
        return np.stack([x, y, zz], axis=1)

    lms_list, ekf_list, elm_list = [], [], []
    nlm = 0
    guard = 0
    while nlm < L:
        guard += 1
        assert guard < 200, "landmark sampling does not converge"
        cand = sample_lms(max(2 * (L - nlm), 64))
        kf = np.repeat(np.arange(P), cand.shape[0])
        lm = np.tile(np.arange(cand.shape[0]), P)
        u, v, z, _ = _project(poses_t, cand, kf, lm, Rcb, tcb, FX, CX, CY)
        vis = (z >= 300.0) & (z <= 12000.0) & (u >= 0) & (u < IMG_W) & (v >= 0) & (v < IMG_H)
        vis = vis.reshape(P, -1)
        for j in range(cand.shape[0]):
            if nlm >= L:
                break
            who = np.nonzero(vis[:, j])[0]
            if who.size < 2:
                continue
            k = int(np.clip(rng.poisson(obs_per_lm - 2.0) + 2, 2, who.size))
            sel = np.sort(rng.choice(who, size=k, replace=False))
            lms_list.append(cand[j])
            ekf_list.append(sel)
            elm_list.append(np.full(k, nlm))
            nlm += 1
    lms_t = np.stack(lms_list, axis=0)
    e_kf = np.concatenate(ekf_list).astype(np.int32)
    e_lm = np.concatenate(elm_list).astype(np.int32)
    E = e_kf.shape[0]
    # --- measurements
    level = rng.integers(0, 8, E)
    u, v, _, _ = _project(poses_t, lms_t, e_kf, e_lm, Rcb, tcb, FX, CX, CY)
    sig = 1.2 ** level
    e_uv = np.stack([u + rng.normal(0.0, 1.0, E) * sig, v + rng.normal(0.0, 1.0, E) * sig], axis=1)
    # a few gross outliers so the Huber branch is exercised
    nout = max(1, E // 200)
    oidx = rng.choice(E, size=nout, replace=False)
    e_uv[oidx] += rng.normal(0.0, 25.0, size=(nout, 2))
    # --- initial estimates
    poses0 = poses_t + rng.normal(0.0, 1.0, size=(P, 3)) * np.array([20.0, 20.0, 0.01])
    poses0[0] = poses_t[0]
    poses0[:, 2] = _norm_angle(poses0[:, 2])
    lms0 = lms_t + rng.normal(0.0, 50.0, size=(L, 3))
    fixed = np.zeros(P, dtype=np.uint8)
    fixed[0] = 1
    e_info = edge_information(poses0, lms0, e_kf, e_lm, level, RBC, TBC, FX)
    # --- odometry (consecutive KFs; the loop is NOT closed by odometry)
    o_i = np.arange(P - 1, dtype=np.int32)
    o_j = o_i + 1
    o_meas = np.zeros((P - 1, 3))
    o_info = np.zeros((P - 1, 9))
    for i in range(P - 1):
        m, c = _preintegrate(rng, poses_t[i], poses_t[i + 1])
        o_meas[i] = m
        o_info[i] = np.linalg.inv(c).reshape(-1)
    return BAGraph(poses=poses0, fixed=fixed, lms=lms0, e_kf=e_kf, e_lm=e_lm, e_uv=e_uv,
                   e_info=e_info, o_i=o_i, o_j=o_j, o_meas=o_meas, o_info=o_info,
                   poses_true=poses_t, lms_true=lms_t, e_level=level.astype(np.int32))


# --------------------------------------------------------------------------
# SE3-expmap graphs: the same room / trajectory in the formulation of
# Map::loadLocalGraph(optimizer, vpEdgesAll, vnAllIdx) (/root/reference/src/Map.cpp:414-566)
# --------------------------------------------------------------------------
def se2_to_Tcw(pose, Rbc=RBC, tbc=TBC):
    """Tcw (4x4) of a body at SE(2) pose (x, y, theta) on the plane: Tcw = (Twb Tbc)^-1."""
    c, s_ = np.cos(pose[2]), np.sin(pose[2])
    Twb = np.array([[c, -s_, 0, pose[0]], [s_, c, 0, pose[1]], [0, 0, 1, 0], [0, 0, 0, 1.0]])
    Tbc = np.eye(4)
    Tbc[:3, :3] = Rbc
    Tbc[:3, 3] = tbc
    return np.linalg.inv(Twb @ Tbc)


def _so3_exp(w):
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])
    if th < 1e-12:
        return np.eye(3) + K
    return np.eye(3) + np.sin(th) / th * K + (1 - np.cos(th)) / th ** 2 * K @ K


def se3_exp_np(u):
    """SE3Quat::exp of (omega, upsilon) as a 4x4 matrix (numpy restatement used by the generator and the independent model)."""
    w, v = np.asarray(u[:3], float), np.asarray(u[3:], float)
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])
    if th < 1e-9:
        V = np.eye(3) + 0.5 * K
    else:
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * K + (th - np.sin(th)) / th ** 3 * K @ K
    T = np.eye(4)
    T[:3, :3] = _so3_exp(w)
    T[:3, 3] = V @ v
    return T


def se3_log_np(T):
    """SE3Quat::log -> (omega, upsilon)."""
    from scipy.spatial.transform import Rotation
    w = Rotation.from_matrix(T[:3, :3]).as_rotvec()
    th = np.linalg.norm(w)
    K = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0.0]])
    if th < 1e-9:
        Vi = np.eye(3) - 0.5 * K
    else:
        Vi = np.eye(3) - 0.5 * K + (1 - th / (2 * np.tan(th / 2))) / th ** 2 * K @ K
    return np.concatenate([w, Vi @ T[:3, 3]])


def se3_adj_np(T):
    R, t = T[:3, :3], T[:3, 3]
    K = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0.0]])
    A = np.zeros((6, 6))
    A[:3, :3] = R
    A[3:, 3:] = R
    A[3:, :3] = K @ R
    return A


def plane_motion_prior_np(Tcw, Rbc=RBC, tbc=TBC, xrot=PLANEMOTION_XROT_INFO, yrot=PLANEMOTION_XROT_INFO, zinfo=PLANEMOTION_Z_INFO):
    """addPlaneMotionSE3Expmap (src/optimizer.cpp:236-314, non-Euler branch): (measurement 4x4, information 6x6)."""
    from scipy.spatial.transform import Rotation
    Tbc = np.eye(4)
    Tbc[:3, :3] = Rbc
    Tbc[:3, 3] = tbc
    Tbw = Tbc @ Tcw
    yaw = Rotation.from_matrix(Tbw[:3, :3]).as_rotvec()[2]
    Tbw2 = np.eye(4)
    Tbw2[:3, :3] = _so3_exp(np.array([0, 0, yaw]))
    Tbw2[:3, 3] = [Tbw[0, 3], Tbw[1, 3], 0.0]
    A = se3_adj_np(Tbc)
    info = A.T @ np.diag([xrot, yrot, 1e-4, 1e-4, 1e-4, zinfo]) @ A
    iu = np.triu_indices(6, 1)
    info[(iu[1], iu[0])] = info[iu]          # the reference copies the upper triangle down (:296-298)
    return np.linalg.inv(Tbc) @ Tbw2, info


@dataclasses.dataclass
class BA3Graph:
    """poses (P,4,4) Tcw, fixed (P,), lms (L,3), e_kf/e_lm (E,), e_uv (E,2), e_w (E,) = invSigma2,
    has_prior (P,), prior_meas (P,4,4), prior_info (P,6,6), o_i/o_j (O,) (vertex 0 / 1 of EdgeSE3Expmap), o_meas (O,4,4),
    o_info (O,6,6)."""
    poses: np.ndarray
    fixed: np.ndarray
    lms: np.ndarray
    e_kf: np.ndarray
    e_lm: np.ndarray
    e_uv: np.ndarray
    e_w: np.ndarray
    has_prior: np.ndarray
    prior_meas: np.ndarray
    prior_info: np.ndarray
    o_i: np.ndarray
    o_j: np.ndarray
    o_meas: np.ndarray
    o_info: np.ndarray
    fx: float = FX
    cx: float = CX
    cy: float = CY
    huber: float = TH_HUBER
    P = property(lambda self: int(self.poses.shape[0]))
    L = property(lambda self: int(self.lms.shape[0]))
    E = property(lambda self: int(self.e_kf.shape[0]))
    O = property(lambda self: int(self.o_i.shape[0]))


@functools.lru_cache(maxsize=4)
def ba3_graph(P: int = 50, L: int = 5000, n_ref: int = 0, seed: int = BA_SEED) -> BA3Graph:
    """The window of ba_graph(P, L) as an SE3-expmap graph: Tcw vertices (with small out-of-plane errors for the
    plane-motion priors to pull back), invSigma2-weighted projection edges, one prior per local key frame, SE3 odometry
    edges between consecutive key frames; the last n_ref key frames are reference key frames (fixed, no prior)."""
    g = ba_graph(P, L, seed=seed)
    rng = np.random.default_rng(seed + 77 * P + L + n_ref)
    poses = np.stack([se2_to_Tcw(p) for p in g.poses])
    for a in range(1, P):   # roll / pitch / height errors of a few mrad / mm
        poses[a] = se3_exp_np(rng.normal(0, 1.0, 6) * np.array([2e-3, 2e-3, 2e-3, 3.0, 3.0, 3.0])) @ poses[a]
    true = np.stack([se2_to_Tcw(p) for p in g.poses_true])
    fixed = g.fixed.copy()
    has_prior = np.ones(P, np.uint8)
    if n_ref:
        fixed[P - n_ref:] = 1
        has_prior[P - n_ref:] = 0
        fixed[0] = 0              # with reference key frames no local one is fixed (Map.cpp:428-438)
    sf = np.ones(8, np.float32)
    for i in range(1, 8):
        sf[i] = sf[i - 1] * SCALE_FACTOR
    inv_sig2 = (1.0 / (sf * sf)).astype(np.float32).astype(np.float64)
    e_w = inv_sig2[g.e_level]
    pm = [plane_motion_prior_np(poses[a]) for a in range(P)]
    nL = P - n_ref
    o_i = np.arange(nL - 1, dtype=np.int32)
    o_j = o_i + 1
    o_meas = np.zeros((len(o_i), 4, 4))
    o_info = np.zeros((len(o_i), 6, 6))
    for k in range(len(o_i)):
        noise = se3_exp_np(rng.normal(0, 1.0, 6) * np.array([1e-3, 1e-3, 2e-3, 2.0, 2.0, 2.0]))
        o_meas[k] = noise @ true[o_j[k]] @ np.linalg.inv(true[o_i[k]])
        A = np.diag([5e5, 5e5, 2e5, 0.2, 0.2, 0.2]) + 0.05 * np.diag([700, 700, 450, 0.45, 0.45, 0.45]) @ rng.normal(0, 1, (6, 6))
        o_info[k] = 0.5 * (A + A.T) + np.diag([1e4, 1e4, 1e4, 0.02, 0.02, 0.02])
        assert np.linalg.eigvalsh(o_info[k]).min() > 0
    return BA3Graph(poses=poses, fixed=fixed, lms=g.lms.copy(), e_kf=g.e_kf, e_lm=g.e_lm, e_uv=g.e_uv, e_w=e_w,
                    has_prior=has_prior, prior_meas=np.stack([m for m, _ in pm]), prior_info=np.stack([i for _, i in pm]),
                    o_i=o_i, o_j=o_j, o_meas=o_meas, o_info=o_info)


# --------------------------------------------------------------------------
# pose graphs of GlobalMapper::GlobalBA: g2o::VertexSE3 (T_w_c) + EdgeSE3Prior + EdgeSE3
# --------------------------------------------------------------------------
def mqt_np(T):
    """g2o::internal::toVectorMQT: (translation, compact quaternion q_xyz of the rotation, w >= 0)."""
    from scipy.spatial.transform import Rotation
    q = Rotation.from_matrix(T[:3, :3]).as_quat()      # (x, y, z, w)
    if q[3] < 0:
        q = -q
    return np.concatenate([T[:3, 3], q[:3]])


def from_mqt_np(v):
    from scipy.spatial.transform import Rotation
    w2 = 1 - v[3:] @ v[3:]
    T = np.eye(4)
    if w2 >= 0:
        T[:3, :3] = Rotation.from_quat([v[3], v[4], v[5], np.sqrt(w2)]).as_matrix()
    T[:3, 3] = v[:3]
    return T


def plane_motion_prior_se3_np(Twc, Rbc=RBC, tbc=TBC, xrot=PLANEMOTION_XROT_INFO, yrot=PLANEMOTION_XROT_INFO, zinfo=PLANEMOTION_Z_INFO):
    """addVertexSE3PlaneMotion (src/optimizer.cpp:336-470, #else branch): (measurement T_w_c 4x4, information 6x6 in the
    order (translation, rotation))."""
    from scipy.spatial.transform import Rotation
    Tbc = np.eye(4)
    Tbc[:3, :3] = Rbc
    Tbc[:3, 3] = tbc
    Twb = Twc @ np.linalg.inv(Tbc)
    yaw = Rotation.from_matrix(Twb[:3, :3]).as_rotvec()[2]
    Twb2 = np.eye(4)
    Twb2[:3, :3] = _so3_exp(np.array([0, 0, yaw]))
    Twb2[:3, 3] = [Twb[0, 3], Twb[1, 3], 0.0]
    R, t = Tbc[:3, :3], Tbc[:3, 3]
    K = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0.0]])
    A = np.zeros((6, 6))            # AdjTR (optimizer.cpp:95-104): [R skew(t) R; 0 R]
    A[:3, :3] = R
    A[3:, 3:] = R
    A[:3, 3:] = K @ R
    return Twb2 @ Tbc, A.T @ np.diag([1e-4, 1e-4, zinfo, xrot, yrot, 1e-4]) @ A


@dataclasses.dataclass
class PoseGraph:
    """poses (P,4,4) T_w_c, fixed (P,), has_prior (P,), prior_meas (P,4,4), prior_info (P,6,6),
    o_i / o_j (O,) = vertex 0 / 1 of every EdgeSE3, o_meas (O,4,4), o_info (O,6,6); information order (translation, rotation)."""
    poses: np.ndarray
    fixed: np.ndarray
    has_prior: np.ndarray
    prior_meas: np.ndarray
    prior_info: np.ndarray
    o_i: np.ndarray
    o_j: np.ndarray
    o_meas: np.ndarray
    o_info: np.ndarray
    poses_true: np.ndarray | None = None
    P = property(lambda self: int(self.poses.shape[0]))
    O = property(lambda self: int(self.o_i.shape[0]))


@functools.lru_cache(maxsize=4)
def pose_graph(P: int = 200, seed: int = BA_SEED) -> PoseGraph:
    """All key frames of a map on the circular trajectory: odometry edges between consecutive key frames, feature edges
    to the key frames 2 and 3 ahead (both an odometry and a feature edge for some consecutive pairs, as the reference
    produces), loop-closing feature edges between the two ends; drifted start, key frame 0 fixed."""
    g = ba_graph(8, 60, seed=seed)           # only for the camera constants
    del g
    rng = np.random.default_rng(seed + 5 * P)
    head = -np.pi + 0.1 + (2 * np.pi - 0.2) * np.arange(P) / max(P - 1, 1)
    ang = head - np.pi / 2
    se2 = np.stack([6000.0 * np.cos(ang), 6000.0 * np.sin(ang), head], axis=1)
    true = np.stack([np.linalg.inv(se2_to_Tcw(p)) for p in se2])            # T_w_c
    poses = true.copy()
    drift = np.eye(4)
    for a in range(1, P):   # accumulated odometry drift + out-of-plane noise
        drift = drift @ from_mqt_np(rng.normal(0, 1.0, 6) * np.array([4.0, 4.0, 0.5, 2e-4, 2e-4, 8e-4]))
        poses[a] = true[a] @ drift
    fixed = np.zeros(P, np.uint8)
    fixed[0] = 1
    pm = [plane_motion_prior_se3_np(poses[a]) for a in range(P)]
    oi, oj, om, ow = [], [], [], []

    def add(i, j, sig_t, sig_r):
        Z = np.linalg.inv(true[i]) @ true[j] @ from_mqt_np(rng.normal(0, 1.0, 6) * np.array([sig_t] * 3 + [sig_r] * 3))
        A = np.diag([1 / sig_t ** 2] * 3 + [1 / sig_r ** 2] * 3)
        B = rng.normal(0, 1, (6, 6)) * 0.05
        S = np.diag(np.sqrt(np.diag(A)))
        W = A + S @ (B + B.T) @ S
        assert np.linalg.eigvalsh(W).min() > 0
        oi.append(i); oj.append(j); om.append(Z); ow.append(W)

    for a in range(P - 1):
        add(a + 1, a, 3.0, 1e-3)                 # mOdoMeasureFrom: (this KF, the previous one)  GlobalMapper.cpp:386-388
    for a in range(P):
        for d in (1, 2, 3):
            if a + d < P and (d > 1 or a % 3 == 0):
                add(a, a + d, 8.0, 2e-3)         # mFtrMeasureFrom
    for a in range(min(4, P // 8)):
        add(P - 1 - a, a, 10.0, 3e-3)            # loop closure
    return PoseGraph(poses=poses, fixed=fixed, has_prior=np.ones(P, np.uint8), prior_meas=np.stack([m for m, _ in pm]),
                     prior_info=np.stack([w for _, w in pm]), o_i=np.array(oi, np.int32), o_j=np.array(oj, np.int32),
                     o_meas=np.stack(om), o_info=np.stack(ow), poses_true=true)


def kf_pair(N: int = 80, seed: int = 0, baseline: float = 400.0):
    """Two key frames (T_w_c) of the circular trajectory `baseline` mm apart and the N map points both see, with the
    per-observation 3x3 information of KeyFrame::mViewMPsInfo (camera frame: 1/sigma^2 with a depth-dependent z term,
    randomly rotated a little) - the inputs of GlobalMapper::CreateFeatEdge / Sparsifier::DoMarginalizeSE3XYZ.
    -> (kf (2,4,4), mp (N,3), m_kf (2N,), m_mp (2N,), m_info (2N,3,3))"""
    rng = np.random.default_rng(1000 + seed)
    th0 = rng.uniform(-2.5, 2.5)
    dth = baseline / 6000.0
    se2 = [np.array([6000 * np.cos(a - np.pi / 2), 6000 * np.sin(a - np.pi / 2), a]) for a in (th0, th0 + dth)]
    kf = np.stack([np.linalg.inv(se2_to_Tcw(p)) for p in se2])
    Xc = np.stack([rng.uniform(-1500, 1500, N), rng.uniform(-900, 900, N), rng.uniform(1500, 7000, N)], 1)   # in camera 0
    mp = (kf[0][:3, :3] @ Xc.T).T + kf[0][:3, 3]
    m_kf, m_mp, m_info = [], [], []
    for j in range(N):
        for k in (0, 1):
            z = np.linalg.inv(kf[k]) @ np.r_[mp[j], 1.0]
            sz = 0.02 * z[2] ** 2 / 400.0 + 5.0
            Rn = _so3_exp(rng.normal(0, 0.05, 3))
            m_kf.append(k); m_mp.append(j)
            m_info.append(Rn @ np.diag([1 / 4.0, 1 / 4.0, 1 / sz ** 2]) @ Rn.T)
    return kf, mp, np.array(m_kf, np.int32), np.array(m_mp, np.int32), np.stack(m_info)


def kidnapped(g: "BAGraph", dxy: float, dth: float, nbad: int, seed: int) -> "BAGraph":
    """A copy of g with `nbad` key frames displaced by N(0, dxy) mm / N(0, dth) rad (landmarks untouched): the first steps
    overshoot and Levenberg-Marquardt has to reject trials.  Headings stay in [-pi, pi) (the reference's Se2 normalises on
    construction, and PreEdgeSE2 has no angle wrap)."""
    import copy
    k = copy.copy(g)          # the generator caches its graphs: never modify the shared instance
    rng = np.random.default_rng(seed)
    k.poses = g.poses.copy()
    idx = rng.choice(np.arange(1, g.P), min(nbad, g.P - 1), replace=False)
    k.poses[idx, :2] += rng.normal(0, dxy, (len(idx), 2))
    k.poses[idx, 2] += rng.normal(0, dth, len(idx))
    k.poses[:, 2] = (k.poses[:, 2] + np.pi) % (2 * np.pi) - np.pi
    return k


def mixed_windows(count: int = 64, p_range=(30, 60), l_range=(3000, 6000), kidnapped_every: int = 11, seed: int = 4242):
    """`count` DISTINCT local windows: key-frame counts, landmark counts and seeds all differ, and every
    `kidnapped_every`-th window starts from displaced key frames so that its LM run rejects trials.  What a mapper's batch of
    windows looks like - as opposed to `count` copies of one graph, where every window has the same sizes, the same solve
    plan and the same accept / reject pattern."""
    out = []
    for k in range(count):
        P = p_range[0] + (7 * k + 3) % (p_range[1] - p_range[0] + 1)
        L = l_range[0] + (997 * k + 131) % (l_range[1] - l_range[0] + 1)
        g = ba_graph(P, L, seed=seed + k)
        if kidnapped_every and k % kidnapped_every == kidnapped_every // 2:
            # Huber-weighted windows of this size shrug most displacements off; the displacement seeds below were picked (with
            # the CPU oracle, once) so that the default batch's kidnapped windows really make g2o's policy reject trials
            g = kidnapped(g, 3000.0, 0.8, 2, _KIDNAP_SEEDS.get((count, seed, k), seed + 100 + k))
        out.append(g)
    return out


# (count, seed, window) -> displacement seed whose start rejects trials (oracle trial histories e.g. [1,1,1,1,6,1,...])
_KIDNAP_SEEDS = {(64, 4242, 5): 13, (64, 4242, 16): 4, (64, 4242, 27): 8, (64, 4242, 38): 2, (64, 4242, 49): 1, (64, 4242, 60): 3}
