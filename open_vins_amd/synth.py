"""Deterministic synthetic MSCKF-update snapshots.

One snapshot = the inputs of one ``UpdaterMSCKF::update`` call: a window of
IMU clones (estimate + FEJ), camera calibration, a prior covariance and F
feature tracks, generated with the measurement model of the reference's
simulator (ov_msckf/src/sim/Simulator.cpp:453-547: lift a random pixel at a
random depth into the global frame, project into every clone camera, keep
0.1 < z < sim_max_feature_gen_dist and pixels inside the image, add N(0, 1 px)
noise, :438-442) and stored the way the front-end stores them
(float32 uv / uv_norm, ov_core/src/feat/Feature.h:49-55).

The camera rig is config/rpng_sim/kalibr_imucam_chain.yaml of the reference
(radtan 752x480 cam0..cam3); the IMU trajectory window is the committed fixture
tests/golden/sim_traj_window.txt (tools/make_traj_fixture.py).

This module is plain numpy: it produces workloads for tests and bench.py, it is
not on the product's compute path.
"""
from __future__ import annotations

import os
from dataclasses import dataclass, field

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
TRAJ_FIXTURE = os.path.join(_HERE, "..", "tests", "golden", "sim_traj_window.txt")

# config/rpng_sim/kalibr_imucam_chain.yaml:3-54 — T_imu_cam = [R_CtoI p_CinI], intrinsics, radtan distortion
_T_IMU_CAM = [
    [[0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
     [0.999557249008, 0.0149672133247, 0.025715529948, -0.064676986768],
     [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949]],
    [[0.0125552670891, -0.999755099723, 0.0182237714554, -0.0198435579556],
     [0.999598781151, 0.0130119051815, 0.0251588363115, 0.0453689425024],
     [-0.0253898008918, 0.0179005838253, 0.999517347078, 0.00786212447038]],
    [[0.0148655429818, -0.999880929698, 0.00414029679422, -0.0216401454975],
     [0.999557249008, 0.0149672133247, 0.025715529948, 0.124676986768],
     [-0.0257744366974, 0.00375618835797, 0.999660727178, 0.00981073058949]],
    [[0.0125552670891, -0.999755099723, 0.0182237714554, -0.0198435579556],
     [0.999598781151, 0.0130119051815, 0.0251588363115, 0.2253689425024],
     [-0.0253898008918, 0.0179005838253, 0.999517347078, 0.00786212447038]],
]
_INTRINSICS = [
    [458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05],
    [457.587, 456.134, 379.999, 255.238, -0.28368365, 0.07451284, -0.00010473, -3.55590700e-05],
    [458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05],
    [457.587, 456.134, 379.999, 255.238, -0.28368365, 0.07451284, -0.00010473, -3.55590700e-05],
]
# an equidistant (fisheye) variant used by the CamEqui parity tests (Kalibr pinhole-equi, EuRoC-like numbers)
_INTRINSICS_EQUI = [
    [461.629, 460.152, 362.680, 246.049, -0.0073, 0.0158, -0.0082, 0.0019],
    [460.281, 459.149, 374.656, 250.205, -0.0049, 0.0055, 0.0032, -0.0021],
    [461.629, 460.152, 362.680, 246.049, -0.0073, 0.0158, -0.0082, 0.0019],
    [460.281, 459.149, 374.656, 250.205, -0.0049, 0.0055, 0.0032, -0.0021],
]
IMG_W, IMG_H = 752, 480

# BASELINE.json configs: cfg -> (C clones, K cams, F features)
CONFIGS = {
    1: dict(C=12, K=1, F=50),
    2: dict(C=30, K=2, F=800),
    3: dict(C=30, K=2, F=2000),
    4: dict(C=30, K=4, F=10000),
    5: dict(C=50, K=4, F=20000),
}


# ----------------------------------------------------------------------------
# JPL quaternion helpers (ov_core/src/utils/quat_ops.h)
# ----------------------------------------------------------------------------
def skew(w):
    return np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]], dtype=np.float64)


def quat_2_rot(q):
    """quat_ops.h:153-158"""
    q = np.asarray(q, dtype=np.float64)
    qv = q[:3]
    return (2 * q[3] ** 2 - 1) * np.eye(3) - 2 * q[3] * skew(qv) + 2 * np.outer(qv, qv)


def rot_2_quat(rot):
    """quat_ops.h:87-119"""
    q = np.zeros(4)
    T = np.trace(rot)
    if rot[0, 0] >= T and rot[0, 0] >= rot[1, 1] and rot[0, 0] >= rot[2, 2]:
        q[0] = np.sqrt((1 + 2 * rot[0, 0] - T) / 4)
        q[1] = (1 / (4 * q[0])) * (rot[0, 1] + rot[1, 0])
        q[2] = (1 / (4 * q[0])) * (rot[0, 2] + rot[2, 0])
        q[3] = (1 / (4 * q[0])) * (rot[1, 2] - rot[2, 1])
    elif rot[1, 1] >= T and rot[1, 1] >= rot[0, 0] and rot[1, 1] >= rot[2, 2]:
        q[1] = np.sqrt((1 + 2 * rot[1, 1] - T) / 4)
        q[0] = (1 / (4 * q[1])) * (rot[0, 1] + rot[1, 0])
        q[2] = (1 / (4 * q[1])) * (rot[1, 2] + rot[2, 1])
        q[3] = (1 / (4 * q[1])) * (rot[2, 0] - rot[0, 2])
    elif rot[2, 2] >= T and rot[2, 2] >= rot[0, 0] and rot[2, 2] >= rot[1, 1]:
        q[2] = np.sqrt((1 + 2 * rot[2, 2] - T) / 4)
        q[0] = (1 / (4 * q[2])) * (rot[0, 2] + rot[2, 0])
        q[1] = (1 / (4 * q[2])) * (rot[1, 2] + rot[2, 1])
        q[3] = (1 / (4 * q[2])) * (rot[0, 1] - rot[1, 0])
    else:
        q[3] = np.sqrt((1 + T) / 4)
        q[0] = (1 / (4 * q[3])) * (rot[1, 2] - rot[2, 1])
        q[1] = (1 / (4 * q[3])) * (rot[2, 0] - rot[0, 2])
        q[2] = (1 / (4 * q[3])) * (rot[0, 1] - rot[1, 0])
    if q[3] < 0:
        q = -q
    return q / np.linalg.norm(q)


def quat_multiply(q, p):
    """quat_ops.h:180-200"""
    Qm = np.zeros((4, 4))
    Qm[:3, :3] = q[3] * np.eye(3) - skew(q[:3])
    Qm[:3, 3] = q[:3]
    Qm[3, :3] = -q[:3]
    Qm[3, 3] = q[3]
    t = Qm @ p
    if t[3] < 0:
        t = -t
    return t / np.linalg.norm(t)


def boxplus_pose(q_p, dx6):
    """PoseJPL::update, ov_core/src/types/PoseJPL.h:74-91"""
    dq = np.array([0.5 * dx6[0], 0.5 * dx6[1], 0.5 * dx6[2], 1.0])
    dq /= np.linalg.norm(dq)
    out = np.empty(7)
    out[:4] = quat_multiply(dq, q_p[:4])
    out[4:] = q_p[4:] + dx6[3:]
    return out


# ----------------------------------------------------------------------------
# camera models (numpy restatement used only to GENERATE measurements)
# ----------------------------------------------------------------------------
def radtan_distort(cam, xn, yn):
    r2 = xn * xn + yn * yn
    r4 = r2 * r2
    rad = 1 + cam[4] * r2 + cam[5] * r4
    x1 = xn * rad + 2 * cam[6] * xn * yn + cam[7] * (r2 + 2 * xn * xn)
    y1 = yn * rad + cam[6] * (r2 + 2 * yn * yn) + 2 * cam[7] * xn * yn
    return cam[0] * x1 + cam[2], cam[1] * y1 + cam[3]


def radtan_undistort(cam, u, v, iters=5):
    """cv::undistortPoints default (5 fixed-point iterations), CamRadtan.h:99-121"""
    x0 = (u - cam[2]) / cam[0]
    y0 = (v - cam[3]) / cam[1]
    x, y = x0, y0
    for _ in range(iters):
        r2 = x * x + y * y
        icdist = 1.0 / (1 + cam[4] * r2 + cam[5] * r2 * r2)
        dx = 2 * cam[6] * x * y + cam[7] * (r2 + 2 * x * x)
        dy = cam[6] * (r2 + 2 * y * y) + 2 * cam[7] * x * y
        x = (x0 - dx) * icdist
        y = (y0 - dy) * icdist
    return x, y


def equi_distort(cam, xn, yn):
    r = np.sqrt(xn * xn + yn * yn)
    th = np.arctan(r)
    thd = th + cam[4] * th ** 3 + cam[5] * th ** 5 + cam[6] * th ** 7 + cam[7] * th ** 9
    cdist = np.where(r > 1e-8, thd / np.maximum(r, 1e-300), 1.0)
    return cam[0] * xn * cdist + cam[2], cam[1] * yn * cdist + cam[3]


def equi_undistort(cam, u, v, iters=10):
    """cv::fisheye::undistortPoints (Newton on theta), CamEqui.h:108-130"""
    x0 = (u - cam[2]) / cam[0]
    y0 = (v - cam[3]) / cam[1]
    thd = np.sqrt(x0 * x0 + y0 * y0)
    th = thd.copy() if isinstance(thd, np.ndarray) else thd
    for _ in range(iters):
        th2 = th * th
        th4, th6, th8 = th2 * th2, th2 * th2 * th2, th2 * th2 * th2 * th2
        f = th * (1 + cam[4] * th2 + cam[5] * th4 + cam[6] * th6 + cam[7] * th8) - thd
        fp = 1 + 3 * cam[4] * th2 + 5 * cam[5] * th4 + 7 * cam[6] * th6 + 9 * cam[7] * th8
        th = th - f / fp
    scale = np.where(thd > 1e-8, np.tan(th) / np.maximum(thd, 1e-300), 1.0)
    return x0 * scale, y0 * scale


# ----------------------------------------------------------------------------
@dataclass
class Problem:
    """One update snapshot in the flat layout of include/ovgpu.h."""
    cfg: int
    seed: int
    N: int
    C: int
    K: int
    P: np.ndarray
    clone_q_p: np.ndarray
    clone_q_p_fej: np.ndarray
    clone_q_p_true: np.ndarray
    clone_cov_id: np.ndarray
    calib_q_p: np.ndarray
    calib_q_p_true: np.ndarray
    intrinsics: np.ndarray
    cam_is_fisheye: np.ndarray
    calib_cov_id: np.ndarray
    intr_cov_id: np.ndarray
    meas_offsets: np.ndarray
    uv: np.ndarray
    uvn: np.ndarray
    clone_idx: np.ndarray
    cam_idx: np.ndarray
    p_FinG_true: np.ndarray
    meta: dict = field(default_factory=dict)
    # SLAM snapshots (make_slam_problem): landmarks that live in the state, feature f observes landmark lm_index[f]
    lm_value: np.ndarray | None = None
    lm_fej: np.ndarray | None = None
    lm_cov_id: np.ndarray | None = None
    lm_index: np.ndarray | None = None
    lm_rep: int = 0
    lm_anchor_cam: np.ndarray | None = None
    lm_anchor_clone: np.ndarray | None = None

    @property
    def F(self):
        return len(self.meas_offsets) - 1

    @property
    def M(self):
        return int(self.meas_offsets[-1])

    @property
    def Dmax(self):
        return 6 * self.C + 14 * self.K

    def subset(self, feat_ids):
        """A snapshot with only the listed features (used to shard across ranks)."""
        feat_ids = np.asarray(feat_ids, dtype=np.int64)
        offs = [0]
        sel = []
        for f in feat_ids:
            a, b = int(self.meas_offsets[f]), int(self.meas_offsets[f + 1])
            sel.append(np.arange(a, b))
            offs.append(offs[-1] + (b - a))
        sel = np.concatenate(sel) if sel else np.zeros(0, dtype=np.int64)
        import copy
        q = copy.copy(self)
        q.meas_offsets = np.asarray(offs, dtype=np.int32)
        q.uv = np.ascontiguousarray(self.uv.reshape(-1, 2)[sel].reshape(-1))
        q.uvn = np.ascontiguousarray(self.uvn.reshape(-1, 2)[sel].reshape(-1))
        q.clone_idx = np.ascontiguousarray(self.clone_idx[sel])
        q.cam_idx = np.ascontiguousarray(self.cam_idx[sel])
        q.p_FinG_true = np.ascontiguousarray(self.p_FinG_true[feat_ids])
        return q


def load_traj_window():
    return np.loadtxt(TRAJ_FIXTURE, comments="#")


def state_sigmas(C, K, imu_intrinsics=False):
    """Per-dof prior sigma in covariance order: IMU 15, [IMU intrinsics 24: dw 6, da 6, tg 9, R_GYROtoIMU 3 -- State.cpp:65-88,
    priors :137-150], dt 1, K x (extrinsic 6, intrinsic 8), C x clone 6.

    IMU block: ov_msckf/src/core/VioManagerHelper.cpp:49-52; calibration: ov_msckf/src/state/State.cpp:150-164;
    clones: SURVEY.md §8d (0.01 rad, 0.05 m)."""
    s = [0.017] * 3 + [0.05] * 3 + [0.01] * 3 + [0.02] * 3 + [0.02] * 3  # q p v bg ba
    if imu_intrinsics:
        s += [0.005] * 6 + [0.008] * 6 + [0.005] * 9 + [0.005] * 3
    s += [0.01]  # dt
    for _ in range(K):
        s += [0.005] * 3 + [0.015] * 3 + [1.0] * 4 + [0.005] * 4
    for _ in range(C):
        s += [0.01] * 3 + [0.05] * 3
    return np.asarray(s)


def make_problem(cfg=2, rep=0, *, C=None, K=None, F=None, track="full", fisheye=False,
                 min_obs=5, pose_noise=1.0, seed=None, shard=0, outlier_frac=0.0, imu_intrinsics=False, calib_noise=1.0) -> Problem:
    """Builds the snapshot of BASELINE.json config `cfg`; C/K/F override its sizes.

    track = "full": every visible observation is kept; "ragged": a contiguous sub-window of clones of
    length ~U[5, C] per feature.  The state (clones, calibration, prior P) depends only on (cfg, rep / seed,
    C, K); `shard` selects an independent feature stream on the same state (feature-sharded multi-GPU runs).
    outlier_frac: fraction of features whose pixels get a gross 15 px offset (exercises the chi2 gate).
    imu_intrinsics: the state also calibrates the IMU intrinsics (BASELINE configs[2] "online cam/IMU calib": 24 more rows of P behind
    the IMU block, N = 248 at 30 clones x 2 cameras); they never get Jacobian columns, only correlations."""
    base = CONFIGS[cfg]
    C = C or base["C"]
    K = K or base["K"]
    F = F if F is not None else base["F"]
    seed = (1000 * cfg + rep) if seed is None else seed
    rng = np.random.default_rng([seed, 1])        # state stream
    rng_P = np.random.default_rng([seed, 2])      # prior covariance stream
    rng_f = np.random.default_rng([seed, 3, shard])  # feature stream

    traj = load_traj_window()
    assert C <= traj.shape[0]
    q_true = traj[:C, 4:8].copy()
    for i in range(C):
        if q_true[i, 3] < 0:
            q_true[i] = -q_true[i]
        q_true[i] /= np.linalg.norm(q_true[i])
    p_true = traj[:C, 1:4].copy()
    clone_true = np.hstack([q_true, p_true])

    # estimate = truth [+] N(0, 0.5 deg, 2 cm); fej = estimate [+] N(0, 0.1 deg, 5 mm)
    clone_est = np.empty_like(clone_true)
    clone_fej = np.empty_like(clone_true)
    for i in range(C):
        d = np.concatenate([rng.normal(0, np.deg2rad(0.5) * pose_noise, 3), rng.normal(0, 0.02 * pose_noise, 3)])
        clone_est[i] = boxplus_pose(clone_true[i], d)
        d = np.concatenate([rng.normal(0, np.deg2rad(0.1) * pose_noise, 3), rng.normal(0, 0.005 * pose_noise, 3)])
        clone_fej[i] = boxplus_pose(clone_est[i], d)

    # cameras (VioManagerOptions.h:262-269: q_ItoC = rot_2_quat(R_CtoI^T), p_IinC = -R_CtoI^T p_CinI)
    calib_true = np.empty((K, 7))
    R_ItoC, p_IinC = [], []
    for k in range(K):
        T = np.asarray(_T_IMU_CAM[k])
        R_CtoI, p_CinI = T[:, :3], T[:, 3]
        calib_true[k, :4] = rot_2_quat(R_CtoI.T)
        calib_true[k, 4:] = -R_CtoI.T @ p_CinI
        R_ItoC.append(quat_2_rot(calib_true[k, :4]))
        p_IinC.append(calib_true[k, 4:].copy())
    intr_true = np.asarray((_INTRINSICS_EQUI if fisheye else _INTRINSICS)[:K], dtype=np.float64)
    distort = equi_distort if fisheye else radtan_distort
    undistort = equi_undistort if fisheye else radtan_undistort

    # estimated calibration = truth [+] 0.3 sigma of the State.cpp:150-164 priors
    calib_est = np.empty_like(calib_true)
    intr_est = intr_true.copy()
    for k in range(K):
        d = np.concatenate([rng.normal(0, 0.3 * 0.005, 3), rng.normal(0, 0.3 * 0.015, 3)]) * calib_noise
        calib_est[k] = boxplus_pose(calib_true[k], d)
        intr_est[k, :4] += rng.normal(0, 0.3 * 1.0, 4) * calib_noise
        intr_est[k, 4:] += rng.normal(0, 0.3 * 0.005 * 0.1, 4) * calib_noise

    R_GtoI = [quat_2_rot(clone_true[i, :4]) for i in range(C)]

    # --- features (vectorised over a batch of candidate points) --------------
    max_dist = 7.0  # sim_max_feature_gen_dist
    Rg = np.stack(R_GtoI)            # [C,3,3]
    Rc = np.stack(R_ItoC)            # [K,3,3]
    pIC = np.stack(p_IinC)           # [K,3]
    uv_l, uvn_l, cl_l, cam_l, pf_l, cnt_l = [], [], [], [], [], []
    nfeat = 0
    batch = max(256, int(F * 1.25) + 16)
    cams_desc = np.arange(K - 1, -1, -1)  # camera groups in descending id (libstdc++ unordered_map order)
    while nfeat < F:
        j = rng_f.integers(0, C, batch)
        k = rng_f.integers(0, K, batch)
        u = rng_f.uniform(0, IMG_W, batch).astype(np.float32).astype(np.float64)
        v = rng_f.uniform(0, IMG_H, batch).astype(np.float32).astype(np.float64)
        depth = rng_f.uniform(5.0, 7.0, batch)
        lo = rng_f.integers(0, C, batch)
        ln = rng_f.integers(5, C + 1, batch)
        noise = rng_f.normal(0, 1.0, (batch, K, C, 2))
        camk = intr_true[k].T  # [8,batch]
        xn, yn = undistort(camk, u, v)
        xn = xn.astype(np.float32).astype(np.float64)  # undistort_cv returns float
        yn = yn.astype(np.float32).astype(np.float64)
        p_FinC = depth[:, None] * np.stack([xn, yn, np.ones_like(xn)], axis=1)
        p_FinI = np.einsum("bji,bj->bi", Rc[k], p_FinC - pIC[k])
        p_FinG = np.einsum("bji,bj->bi", Rg[j], p_FinI) + p_true[j]
        t = np.einsum("cij,bcj->bci", Rg, p_FinG[:, None, :] - p_true[None, :, :])  # [b,C,3] in IMU frames
        valid = np.zeros((batch, K, C), dtype=bool)
        ud = np.zeros((batch, K, C))
        vd = np.zeros((batch, K, C))
        for kk in range(K):
            pc = np.einsum("ij,bcj->bci", Rc[kk], t) + pIC[kk]
            z = pc[..., 2]
            ok = (z <= max_dist) & (z >= 0.1)
            zs = np.where(ok, z, 1.0)
            xf = (pc[..., 0] / zs).astype(np.float32).astype(np.float64)
            yf = (pc[..., 1] / zs).astype(np.float32).astype(np.float64)
            a_, b_ = distort(intr_true[kk], xf, yf)
            a_ = a_.astype(np.float32).astype(np.float64)
            b_ = b_.astype(np.float32).astype(np.float64)
            ok &= (a_ >= 0) & (a_ <= IMG_W) & (b_ >= 0) & (b_ <= IMG_H)
            valid[:, kk, :] = ok
            ud[:, kk, :] = a_
            vd[:, kk, :] = b_
        if track == "ragged":
            c0 = np.maximum(np.minimum(lo, C - ln), 0)
            cc = np.arange(C)[None, :]
            win = (cc >= c0[:, None]) & (cc < (c0 + ln)[:, None])
            valid &= win[:, None, :]
        # noisy raw pixel (float32) and its undistorted normalised coordinate (float32)
        if outlier_frac > 0:
            bad = rng_f.uniform(0, 1, batch) < outlier_frac
            noise = noise + np.where(bad, 15.0, 0.0)[:, None, None, None] * np.sign(noise)
        un = (ud + noise[..., 0]).astype(np.float32)
        vn = (vd + noise[..., 1]).astype(np.float32)
        xu = np.zeros_like(ud)
        yu = np.zeros_like(ud)
        for kk in range(K):
            xu[:, kk, :], yu[:, kk, :] = undistort(intr_true[kk], un[:, kk, :].astype(np.float64), vn[:, kk, :].astype(np.float64))
        count = valid.reshape(batch, -1).sum(axis=1)
        keep = np.nonzero(count >= min_obs)[0][: F - nfeat]
        if keep.size == 0:
            continue
        vk = valid[keep][:, cams_desc, :]
        sel = vk.reshape(len(keep), -1)
        uv_l.append(np.stack([un[keep][:, cams_desc, :].reshape(len(keep), -1)[sel],
                              vn[keep][:, cams_desc, :].reshape(len(keep), -1)[sel]], axis=1))
        uvn_l.append(np.stack([xu[keep][:, cams_desc, :].reshape(len(keep), -1)[sel],
                               yu[keep][:, cams_desc, :].reshape(len(keep), -1)[sel]], axis=1).astype(np.float32))
        cam_grid = np.broadcast_to(cams_desc[None, :, None], vk.shape).reshape(len(keep), -1)
        cl_grid = np.broadcast_to(np.arange(C)[None, None, :], vk.shape).reshape(len(keep), -1)
        cam_l.append(cam_grid[sel])
        cl_l.append(cl_grid[sel])
        cnt_l.append(sel.sum(axis=1))
        pf_l.append(p_FinG[keep])
        nfeat += len(keep)
    counts = np.concatenate(cnt_l)
    offs = np.concatenate([[0], np.cumsum(counts)])
    uv_l = np.concatenate(uv_l, axis=0)
    uvn_l = np.concatenate(uvn_l, axis=0)
    cl_l = np.concatenate(cl_l)
    cam_l = np.concatenate(cam_l)
    pf_l = np.concatenate(pf_l, axis=0)

    # --- prior covariance ---------------------------------------------------
    base = 16 + (24 if imu_intrinsics else 0)
    N = base + 14 * K + 6 * C
    sig = state_sigmas(C, K, imu_intrinsics)
    assert sig.shape[0] == N
    G = np.tril(rng_P.normal(0, 1.0 / np.sqrt(N), (N, N)), -1)
    L = sig[:, None] * (np.eye(N) + 0.1 * G)
    P = L @ L.T
    P = 0.5 * (P + P.T)

    calib_cov_id = np.array([base + 14 * k for k in range(K)], dtype=np.int32)
    intr_cov_id = np.array([base + 14 * k + 6 for k in range(K)], dtype=np.int32)
    clone_cov_id = np.array([base + 14 * K + 6 * c for c in range(C)], dtype=np.int32)

    return Problem(
        cfg=cfg, seed=seed, N=N, C=C, K=K, P=np.ascontiguousarray(P),
        clone_q_p=np.ascontiguousarray(clone_est), clone_q_p_fej=np.ascontiguousarray(clone_fej),
        clone_q_p_true=clone_true, clone_cov_id=clone_cov_id,
        calib_q_p=np.ascontiguousarray(calib_est), calib_q_p_true=calib_true,
        intrinsics=np.ascontiguousarray(intr_est),
        cam_is_fisheye=np.full(K, 1 if fisheye else 0, dtype=np.uint8),
        calib_cov_id=calib_cov_id, intr_cov_id=intr_cov_id,
        meas_offsets=np.asarray(offs, dtype=np.int32),
        uv=np.asarray(uv_l, dtype=np.float32).reshape(-1), uvn=np.asarray(uvn_l, dtype=np.float32).reshape(-1),
        clone_idx=np.asarray(cl_l, dtype=np.int32), cam_idx=np.asarray(cam_l, dtype=np.int32),
        p_FinG_true=np.asarray(pf_l, dtype=np.float64).reshape(-1, 3),
        meta=dict(track=track, fisheye=bool(fisheye), min_obs=min_obs, imu_intrinsics=bool(imu_intrinsics)),
    )


def tight_window_problem(cfg=2, scale=0.05, **kw) -> Problem:
    """make_problem's snapshot `scale` times as uncertain — clone and calibration errors AND the prior covariance scaled alike, so the
    snapshot stays consistent.  SURVEY 8(d)'s snapshot gives every clone an independent 0.57 deg / 5 cm and the calibration its
    start-up priors (tens of pixels of predicted-pixel uncertainty); a running filter's window is tight relative to its newest clone and
    its calibration has converged — the regime in which the gate's residual bound (ovgpu_options::gate_always_factor = 0) decides
    most features (tests/test_rpng_sim_loop.py: 99.8 %)."""
    prob = make_problem(cfg, pose_noise=scale, calib_noise=scale, **kw)
    prob.P = np.ascontiguousarray(prob.P * (scale * scale))
    return prob


def realistic_prior(prob: Problem, sigma_g_p=1.0, sigma_g_th=0.05, q_th=5e-5, q_p=1e-5, seed=0):
    """A prior covariance shaped like a live sliding window instead of make_problem's generic SPD matrix.

    The oldest clone carries the accumulated GLOBAL uncertainty (sigma_g_p metres, sigma_g_th radians: position and
    yaw of a VIO drift without bound), every later clone is the previous one plus the IMU's noise over one camera
    period (q_th rad, q_p m; the chain StateHelper::clone / augment_clone builds, StateHelper.cpp:341-391, with
    Propagator.cpp:76-137 adding Q_d at IMU level, and the lever arm of the rotation error), and the IMU pose is an
    exact copy of the newest clone (the state right after the cloning step: P is singular, its clone / calibration
    block is not).  Consecutive clones are then correlated to 1 - (q / sigma_g)^2: cond(P_DD) = 1e8 ... 1e16 for the
    defaults and beyond, and P is LARGE exactly along the directions the measurements do not see.
    Returns the N x N matrix (prob.P is not modified)."""
    rng = np.random.default_rng([int(seed), 11])
    N, C, K = prob.N, prob.C, prob.K
    sig = state_sigmas(C, K)
    c0 = int(prob.clone_cov_id[0])
    A = np.zeros((N, N))
    Gm = np.tril(rng.normal(0, 1 / np.sqrt(N), (N, N)), -1)
    L = sig[:, None] * (np.eye(N) + 0.1 * Gm)
    A[:c0, :c0] = L[:c0, :c0]  # IMU velocity / biases, dt, calibration: State.cpp:150-164 priors, mildly correlated
    for c in range(C):
        i = c0 + 6 * c
        if c == 0:
            A[i:i + 3, i:i + 3] = np.eye(3) * sigma_g_th
            A[i + 3:i + 6, i + 3:i + 6] = np.eye(3) * sigma_g_p
            A[i:i + 6, :c0] = 0.05 * rng.normal(0, 1, (6, c0)) * np.array([sigma_g_th] * 3 + [sigma_g_p] * 3)[:, None] / np.sqrt(c0)
        else:
            A[i:i + 6, :] = A[i - 6:i, :]
            dp = prob.clone_q_p[c, 4:7] - prob.clone_q_p[c - 1, 4:7]
            A[i + 3:i + 6, :] += -skew(dp) @ A[i - 6:i - 3, :]
            A[i:i + 3, i:i + 3] += np.eye(3) * q_th
            A[i + 3:i + 6, i + 3:i + 6] += np.eye(3) * q_p
    iN = c0 + 6 * (C - 1)
    A[0:6, :] = A[iN:iN + 6, :]
    P = A @ A.T
    return np.ascontiguousarray(0.5 * (P + P.T))


def landmark_from_xyz(rep, p):
    """ov_type::Landmark::set_from_xyz (Landmark.cpp:66-141): xyz -> representation coordinates, rows of p."""
    p = np.asarray(p, dtype=np.float64)
    if rep in (1, 3):  # GLOBAL_ / ANCHORED_FULL_INVERSE_DEPTH: (theta, phi, rho)
        rho = 1.0 / np.linalg.norm(p, axis=-1)
        return np.stack([np.arctan2(p[..., 1], p[..., 0]), np.arccos(rho * p[..., 2]), rho], axis=-1)
    if rep in (4, 5):  # ANCHORED_MSCKF_INVERSE_DEPTH: (alpha, beta, rho); ANCHORED_INVERSE_DEPTH_SINGLE: (bearing x, y, rho), rho the state
        return np.stack([p[..., 0] / p[..., 2], p[..., 1] / p[..., 2], 1.0 / p[..., 2]], axis=-1)
    return p.copy()


def landmark_to_xyz(rep, v):
    """ov_type::Landmark::get_xyz (Landmark.cpp:25-62)."""
    v = np.asarray(v, dtype=np.float64)
    if rep in (1, 3):
        return np.stack([np.cos(v[..., 0]) * np.sin(v[..., 1]), np.sin(v[..., 0]) * np.sin(v[..., 1]), np.cos(v[..., 1])], axis=-1) / v[..., 2:3]
    if rep in (4, 5):
        return np.stack([v[..., 0], v[..., 1], np.ones_like(v[..., 0])], axis=-1) / v[..., 2:3]
    return v.copy()


def make_slam_problem(cfg=2, L=20, *, lm_rep=0, lm_noise=0.05, seed=None, **kw) -> Problem:
    """A snapshot for UpdaterSLAM::update: L landmarks live in the state behind the clones, each observed by one track.
    Estimate = truth + N(0, lm_noise) in xyz (global, or in the anchor camera frame of the track's first measurement for an
    anchored representation lm_rep >= 2), fej = estimate + N(0, lm_noise / 5); both are stored in representation
    coordinates as ov_type::Landmark does.  The prior covariance is rebuilt for the larger state the same way
    make_problem does (landmark sigma = 2 lm_noise in representation coordinates)."""
    prob = make_problem(cfg, F=L, seed=seed, **kw)
    rng = np.random.default_rng([prob.seed, 7])
    N0 = prob.N
    # lm_rep: one representation for all, or one per landmark (ArUco corners in StateOptions::feat_rep_aruco next to landmarks in
    # feat_rep_slam: UpdaterSLAM::update reads the representation from each landmark, UpdaterSLAM.cpp:336-341)
    reps = np.full(L, int(lm_rep), np.int32) if np.isscalar(lm_rep) else np.ascontiguousarray(lm_rep, dtype=np.int32)
    assert reps.shape == (L,)
    mixed = not np.isscalar(lm_rep)
    szs = np.where(reps == 5, 1, 3)  # state dof of each landmark
    N = N0 + int(szs.sum())
    xyz = prob.p_FinG_true + rng.normal(0, lm_noise, (L, 3))
    xyz_fej = xyz + rng.normal(0, lm_noise / 5, (L, 3))
    prob.lm_rep = int(reps[0]) if L else 0
    prob.lm_rep_each = reps if mixed else None
    if (reps >= 2).any():
        first = prob.meas_offsets[:-1]
        prob.lm_anchor_cam = np.where(reps >= 2, prob.cam_idx[first], -1).astype(np.int32)
        prob.lm_anchor_clone = np.where(reps >= 2, prob.clone_idx[first], -1).astype(np.int32)

        def to_anchor(p):
            out = p.copy()
            for l in range(L):
                if reps[l] < 2:
                    continue
                qc, qk = prob.clone_q_p[prob.lm_anchor_clone[l]], prob.calib_q_p[prob.lm_anchor_cam[l]]
                R_GtoI, R_ItoC = quat_2_rot(qc[:4]), quat_2_rot(qk[:4])
                out[l] = R_ItoC @ (R_GtoI @ (p[l] - qc[4:7])) + qk[4:7]
            return out

        xyz, xyz_fej = to_anchor(xyz), to_anchor(xyz_fej)
    val, fej = np.zeros((L, 3)), np.zeros((L, 3))
    for r in np.unique(reps):
        sel = reps == r
        val[sel], fej[sel] = landmark_from_xyz(int(r), xyz[sel]), landmark_from_xyz(int(r), xyz_fej[sel])
    prob.lm_value = np.ascontiguousarray(val)
    prob.lm_fej = np.ascontiguousarray(fej)
    prob.lm_cov_id = (N0 + np.concatenate([[0], np.cumsum(szs)[:-1]])).astype(np.int32) if L else np.zeros(0, np.int32)
    prob.lm_index = np.arange(L, dtype=np.int32)
    depth = np.linalg.norm(xyz, axis=1)
    lm_sig_parts = []
    for l in range(L):
        sg = np.full(3, 2 * lm_noise)
        if reps[l] in (1, 3, 4, 5):  # angles / normalised coordinates and an inverse depth: scale the sigma to the coordinates
            sg = np.array([2 * lm_noise / depth[l], 2 * lm_noise / depth[l], 2 * lm_noise / depth[l] ** 2])
        lm_sig_parts.append(sg[2:3] if szs[l] == 1 else sg)  # single depth: only the inverse depth is a state variable
    lm_sig = np.concatenate(lm_sig_parts) if L else np.zeros(0)
    sig = np.concatenate([state_sigmas(prob.C, prob.K), lm_sig.reshape(-1)])
    G = np.tril(rng.normal(0, 1.0 / np.sqrt(N), (N, N)), -1)
    Lc = sig[:, None] * (np.eye(N) + 0.1 * G)
    P = Lc @ Lc.T
    prob.P = np.ascontiguousarray(0.5 * (P + P.T))
    prob.N = N
    return prob


def algorithmic_flops(prob: Problem):
    """SURVEY.md §8(d) per-feature algorithmic FLOPs, summed over the snapshot, plus the once-per-update EKF term.

    FLOPs(m, d_f, D) = 1035 m + 400 m + 36 m (d_f + 3) + (2 r d_f^2 + r^2 d_f + r^3/3 + 2 r^2) + 2 r D^2, r = 2m - 3.
    Returns (total, compress_part) where compress_part = sum 2 r D^2 (rows of the measurement-compression QR)."""
    D = prob.Dmax
    N = prob.N
    tot = 0.0
    comp = 0.0
    for f in range(prob.F):
        a, b = int(prob.meas_offsets[f]), int(prob.meas_offsets[f + 1])
        m = b - a
        if m < 2:
            continue
        r = 2 * m - 3
        ncl = len(set(prob.clone_idx[a:b].tolist()))
        ncam = len(set(prob.cam_idx[a:b].tolist()))
        d_f = 6 * ncl + 14 * ncam
        tot += 1035 * m + 400 * m + 36 * m * (d_f + 3) + (2 * r * d_f ** 2 + r * r * d_f + r ** 3 / 3 + 2 * r * r) + 2 * r * D * D
        comp += 2 * r * D * D
    tot += 4 * N * D * D + 2.33 * D ** 3 + N * N * D
    return tot, comp


def algorithmic_flops_system(prob: Problem):
    """The share of algorithmic_flops() that the per-feature kernel (k_system) carries: Jacobians a6-a8 (100 m + 400 m), nullspace
    projection a10 (36 m (d_f + 3)) and the chi2 gate a11 (2 r d_f^2 + r^2 d_f + r^3 / 3 + 2 r^2); the triangulation (935 m of
    the 1035 m term) and the compression run in their own kernels."""
    tot = 0.0
    for f in range(prob.F):
        a, b = int(prob.meas_offsets[f]), int(prob.meas_offsets[f + 1])
        m = b - a
        if m < 2:
            continue
        r = 2 * m - 3
        d_f = 6 * len(set(prob.clone_idx[a:b].tolist())) + 14 * len(set(prob.cam_idx[a:b].tolist()))
        tot += 500 * m + 36 * m * (d_f + 3) + (2 * r * d_f ** 2 + r * r * d_f + r ** 3 / 3 + 2 * r * r)
    return tot
