"""Closed-loop harness for BASELINE.json configs[0] (rpng_sim mono, 11-clone window, ~50 MSCKF features per update).

NOT part of the product path: a small host-side sliding-window filter around the update under test, so that the same
measurement stream can be run once with the CPU oracle as updater and once with the GPU library, and the two
trajectories / ATE values compared ("ATE parity").  What is restated minimally:

  * cloning: the newest clone is predicted from the previous one by the true relative motion plus noise
    (R_new = dR R_last, p_new = p_last + dp): error-state Jacobian blockdiag(dR, I), noise Q — the role of
    Propagator::propagate_and_clone + StateHelper::augment_clone (Propagator.cpp:76-137, StateHelper.cpp:579-616);
  * first-estimate Jacobians: a clone's fej value is its estimate at cloning time and never changes;
  * MSCKF feature selection: a track is used at the frame after its last observation (VioManager.cpp:366-429),
    every observation lies inside the window;
  * marginalisation of the oldest clone after the update (StateHelper.cpp:271-339): row / column deletion.

The trajectory is the committed fixture tests/golden/sim_traj_window.txt (64 poses at 10 Hz of the rpng_sim B-spline).
"""
from __future__ import annotations

import numpy as np

from . import synth


def _rot(q):
    return synth.quat_2_rot(q)


class Stream:
    """Truth, tracks and noise of one closed-loop run; identical for every updater under test."""

    def __init__(self, C=12, feats_per_frame=50, seed=7, sigma_px=1.0, q_theta=np.deg2rad(0.15), q_p=0.01, T=None, K=1):
        rng = np.random.default_rng(seed)
        self.K = K
        if T is None:
            traj = synth.load_traj_window()
            self.T = traj.shape[0]
            q = traj[:, 4:8].copy()
            q[q[:, 3] < 0] *= -1
            q /= np.linalg.norm(q, axis=1, keepdims=True)
            self.truth = np.hstack([q, traj[:, 1:4]])
        else:
            self.T = T
            self.truth = _analytic_trajectory(T)
        self.C = C
        if K > 1:
            self._init_rig(rng, C, K, feats_per_frame, sigma_px, q_theta, q_p)
            return
        Tm = np.asarray(synth._T_IMU_CAM[0])
        R_CtoI, p_CinI = Tm[:, :3], Tm[:, 3]
        self.calib_true = np.concatenate([synth.rot_2_quat(R_CtoI.T), -R_CtoI.T @ p_CinI])
        self.intr_true = np.asarray(synth._INTRINSICS[0], dtype=np.float64)
        self.q_theta, self.q_p = q_theta, q_p
        self.clone_noise = rng.normal(0, 1, (self.T, 6)) * np.array([q_theta] * 3 + [q_p] * 3)
        self.init_noise = rng.normal(0, 1, (C, 6)) * np.array([0.01] * 3 + [0.03] * 3)
        self.calib_noise = np.concatenate([rng.normal(0, 0.3 * 0.005, 3), rng.normal(0, 0.3 * 0.015, 3)])
        self.intr_noise = np.concatenate([rng.normal(0, 0.3, 4), rng.normal(0, 0.3 * 0.0005, 4)])
        # tracks: (frames a..b, pixel / normalised observations); used in the update of frame b + 1
        R_ItoC, p_IinC = _rot(self.calib_true[:4]), self.calib_true[4:]
        Rg = [_rot(self.truth[i, :4]) for i in range(self.T)]
        self.tracks = {t: [] for t in range(self.T)}
        for t_use in range(C - 1, self.T):
            b = t_use - 1
            made = 0
            while made < feats_per_frame:
                a = int(rng.integers(max(t_use - (C - 1), 0), b - 3))
                j = int(rng.integers(a, b + 1))
                u, v = rng.uniform(0, synth.IMG_W), rng.uniform(0, synth.IMG_H)
                xn, yn = synth.radtan_undistort(self.intr_true, np.array([u]), np.array([v]))
                depth = rng.uniform(5.0, 7.0)
                p_C = depth * np.array([xn[0], yn[0], 1.0])
                p_G = Rg[j].T @ (R_ItoC.T @ (p_C - p_IinC)) + self.truth[j, 4:]
                obs = []
                for i in range(a, b + 1):
                    pc = R_ItoC @ (Rg[i] @ (p_G - self.truth[i, 4:])) + p_IinC
                    if not (0.1 <= pc[2] <= 7.0):
                        continue
                    ud, vd = synth.radtan_distort(self.intr_true, np.float32(pc[0] / pc[2]).astype(np.float64), np.float32(pc[1] / pc[2]).astype(np.float64))
                    if not (0 <= ud <= synth.IMG_W and 0 <= vd <= synth.IMG_H):
                        continue
                    un, vn = np.float32(ud + rng.normal(0, sigma_px)), np.float32(vd + rng.normal(0, sigma_px))
                    xu, yu = synth.radtan_undistort(self.intr_true, np.array([float(un)]), np.array([float(vn)]))
                    obs.append((i, un, vn, np.float32(xu[0]), np.float32(yu[0])))
                if len(obs) >= 5:
                    self.tracks[t_use].append(obs)
                    made += 1


def _stream_init_rig(self, rng, C, K, feats_per_frame, sigma_px, q_theta, q_p):
    """K-camera rig (BASELINE configs[1..3] shape: stereo, 30-clone window, hundreds of features per frame), generated with numpy
    over all features of a frame at once.  Observations are (frame, un, vn, xu, yu, camera); calib_true / intr_true are [K, .]."""
    calib, intr = [], []
    for k in range(K):
        Tm = np.asarray(synth._T_IMU_CAM[k])
        R_CtoI, p_CinI = Tm[:, :3], Tm[:, 3]
        calib.append(np.concatenate([synth.rot_2_quat(R_CtoI.T), -R_CtoI.T @ p_CinI]))
        intr.append(np.asarray(synth._INTRINSICS[k], dtype=np.float64))
    self.calib_true, self.intr_true = np.stack(calib), np.stack(intr)
    self.q_theta, self.q_p = q_theta, q_p
    self.clone_noise = rng.normal(0, 1, (self.T, 6)) * np.array([q_theta] * 3 + [q_p] * 3)
    self.init_noise = rng.normal(0, 1, (C, 6)) * np.array([0.01] * 3 + [0.03] * 3)
    self.calib_noise = np.stack([np.concatenate([rng.normal(0, 0.3 * 0.005, 3), rng.normal(0, 0.3 * 0.015, 3)]) for _ in range(K)])
    self.intr_noise = np.stack([np.concatenate([rng.normal(0, 0.3, 4), rng.normal(0, 0.3 * 0.0005, 4)]) for _ in range(K)])
    R_ItoC = np.stack([_rot(c[:4]) for c in self.calib_true])
    p_IinC = np.stack([c[4:] for c in self.calib_true])
    Rg = np.stack([_rot(self.truth[i, :4]) for i in range(self.T)])
    pg = self.truth[:, 4:]
    self.tracks = {t: [] for t in range(self.T)}
    for t_use in range(C - 1, self.T):
        b = t_use - 1
        lo = max(t_use - (C - 1), 0)
        made = 0
        while made < feats_per_frame:
            nb = 2 * (feats_per_frame - made) + 16
            a = rng.integers(lo, b - 3, nb)
            j = rng.integers(a, b + 1)
            k0 = rng.integers(0, K, nb)
            u, v = rng.uniform(0, synth.IMG_W, nb), rng.uniform(0, synth.IMG_H, nb)
            xn, yn = synth.radtan_undistort(self.intr_true[k0].T, u, v)
            depth = rng.uniform(5.0, 7.0, nb)
            p_C = depth[:, None] * np.stack([xn, yn, np.ones(nb)], axis=1)
            p_G = np.einsum("bji,bj->bi", Rg[j], np.einsum("bji,bj->bi", R_ItoC[k0], p_C - p_IinC[k0])) + pg[j]
            frames = np.arange(lo, b + 1)
            p_I = np.einsum("fij,bfj->bfi", Rg[frames], p_G[:, None, :] - pg[frames][None, :, :])        # [nb, nf, 3]
            noise = rng.normal(0, sigma_px, (nb, K, len(frames), 2))
            per_feat = [[] for _ in range(nb)]
            for k in range(K):
                pc = np.einsum("ij,bfj->bfi", R_ItoC[k], p_I) + p_IinC[k]
                z = pc[..., 2]
                ok = (z >= 0.1) & (z <= 7.0) & (frames[None, :] >= a[:, None])
                zs = np.where(ok, z, 1.0)
                xf = (pc[..., 0] / zs).astype(np.float32).astype(np.float64)
                yf = (pc[..., 1] / zs).astype(np.float32).astype(np.float64)
                ud, vd = synth.radtan_distort(self.intr_true[k], xf, yf)
                ok &= (ud >= 0) & (ud <= synth.IMG_W) & (vd >= 0) & (vd <= synth.IMG_H)
                un, vn = (ud + noise[:, k, :, 0]).astype(np.float32), (vd + noise[:, k, :, 1]).astype(np.float32)
                xu, yu = synth.radtan_undistort(self.intr_true[k], un.astype(np.float64), vn.astype(np.float64))
                xu, yu = xu.astype(np.float32), yu.astype(np.float32)
                for bi, fi in zip(*np.nonzero(ok)):
                    per_feat[bi].append((int(frames[fi]), un[bi, fi], vn[bi, fi], xu[bi, fi], yu[bi, fi], k))
            for obs in per_feat:
                if len(obs) >= 5 and made < feats_per_frame:
                    self.tracks[t_use].append(sorted(obs, key=lambda o: (o[0], o[5])))
                    made += 1


Stream._init_rig = _stream_init_rig


def _analytic_trajectory(T, dt=0.1):
    """A long smooth trajectory for closed loops beyond the 64-pose fixture: ~1 m/s on a 4 m circle with a vertical wobble, yawing
    with the motion, roll / pitch of a few degrees.  Rows [q_GtoI (JPL), p_IinG] like the fixture."""
    out = np.zeros((T, 7))
    for i in range(T):
        t = dt * i
        yaw, pitch, roll = 0.25 * t, 0.08 * np.sin(0.9 * t), 0.06 * np.cos(0.7 * t)
        cz, sz, cy, sy, cx, sx = np.cos(yaw), np.sin(yaw), np.cos(pitch), np.sin(pitch), np.cos(roll), np.sin(roll)
        R_ItoG = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @ np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
        q = synth.rot_2_quat(R_ItoG.T)
        out[i, :4] = -q if q[3] < 0 else q
        out[i, 4:] = [4.0 * np.cos(0.25 * t), 4.0 * np.sin(0.25 * t), 1.0 + 0.3 * np.sin(0.6 * t)]
    return out


def _compose(last_est, truth_last, truth_new):
    """Newest clone predicted from the previous estimate by the TRUE relative motion."""
    dR = _rot(truth_new[:4]) @ _rot(truth_last[:4]).T
    q = synth.rot_2_quat(dR @ _rot(last_est[:4]))
    if q[3] < 0:
        q = -q
    return np.concatenate([q, last_est[4:] + truth_new[4:] - truth_last[4:]]), dR


def _frame_problem(stream, tracks, frames, N, P, clones, fej, calib, intr):
    """The snapshot handed to the updater at one frame: state + the tracks that ended at the previous frame.  Camera groups of a
    track in the reference's iteration order of Feature::timestamps (reverse first insertion: normally descending camera id,
    SURVEY Q13), time order inside a group."""
    C, K = stream.C, stream.K
    base = 16 + 14 * K
    idx = {f: i for i, f in enumerate(frames)}
    offs, uv, uvn, ci, cam = [0], [], [], [], []
    for obs_all in tracks:
        # Feature::timestamps is an unordered_map keyed by camera: libstdc++ iterates it in REVERSE order of first insertion, and the
        # front end delivers a frame camera 0 first (keys of cameras whose observations all left the window stay in the map)
        first_seen = list(dict.fromkeys((o[5] if len(o) > 5 else 0) for o in sorted(obs_all, key=lambda o: (o[0], o[5] if len(o) > 5 else 0))))
        obs = [o for o in obs_all if o[0] in idx]
        for k in reversed(first_seen):
            for o in obs:
                if (o[5] if len(o) > 5 else 0) != k:
                    continue
                uv += [o[1], o[2]]
                uvn += [o[3], o[4]]
                ci.append(idx[o[0]])
                cam.append(k)
        offs.append(len(ci))
    calib_true = stream.calib_true if K > 1 else stream.calib_true[None, :]
    return synth.Problem(
        cfg=1, seed=0, N=N, C=C, K=K, P=np.ascontiguousarray(P), clone_q_p=np.ascontiguousarray(clones),
        clone_q_p_fej=np.ascontiguousarray(fej), clone_q_p_true=stream.truth[frames], clone_cov_id=(base + 6 * np.arange(C)).astype(np.int32),
        calib_q_p=np.ascontiguousarray(calib), calib_q_p_true=calib_true, intrinsics=np.ascontiguousarray(intr),
        cam_is_fisheye=np.zeros(K, np.uint8), calib_cov_id=(16 + 14 * np.arange(K)).astype(np.int32), intr_cov_id=(22 + 14 * np.arange(K)).astype(np.int32),
        meas_offsets=np.asarray(offs, np.int32), uv=np.asarray(uv, np.float32), uvn=np.asarray(uvn, np.float32),
        clone_idx=np.asarray(ci, np.int32), cam_idx=np.asarray(cam, np.int32), p_FinG_true=np.zeros((len(tracks), 3)))


def _initial_state(stream):
    C, K = stream.C, stream.K
    base = 16 + 14 * K
    sig = synth.state_sigmas(C, K)
    sig[base:] = np.tile([0.01] * 3 + [0.03] * 3, C)
    N = base + 6 * C
    P = np.diag(sig ** 2)
    clones = np.stack([synth.boxplus_pose(stream.truth[i], stream.init_noise[i]) for i in range(C)])
    if K > 1:
        calib = np.stack([synth.boxplus_pose(stream.calib_true[k], stream.calib_noise[k]) for k in range(K)])
        intr = stream.intr_true + stream.intr_noise
    else:
        calib = synth.boxplus_pose(stream.calib_true, stream.calib_noise)[None, :]
        intr = (stream.intr_true + stream.intr_noise)[None, :]
    return base, N, P, clones, calib, intr


def run_resident(stream: Stream, up, track_store=False):
    """The same filter with the covariance RESIDENT on the device between frames: cloning = ovgpu_state_augment_clone +
    ovgpu_state_propagate, marginalisation = ovgpu_state_marginalize, update = ovgpu_set_features + ovgpu_msckf_update;
    the state is uploaded once.  With track_store the observations live on the device as well: every frame appends its
    observations (ovgpu_tracks_append), the batch of the frame's MSCKF features — the schedule of the stream: a track is used
    at the frame after its nominal end, which run() follows too — is assembled on the device (ovgpu_tracks_to_features)
    and the tracks are erased after use.  `up` is an open_vins_amd.updater.UpdaterMSCKF."""
    C = stream.C
    base, N, P, clones, calib, intr = _initial_state(stream)
    frames = list(range(C))
    up.set_problem(_frame_problem(stream, [], frames, N, P, clones, clones.copy(), calib, intr))
    est, used = {frames[-1]: clones[-1].copy()}, {}
    last = clones[-1].copy()
    time_of = lambda f: 10.0 + 0.1 * f  # clone / observation time of frame f
    by_frame, ids_at = {}, {}
    if track_store:
        n_tracks = sum(len(v) for v in stream.tracks.values())
        up.tracks_create(n_tracks + 8, stream.K * C + 4)
        fid = 0
        for t in sorted(stream.tracks):  # ids in the order the host loop meets the tracks
            if t < C:
                continue                  # tracks that end before the first update frame are never used by run() either
            for obs in stream.tracks[t]:
                for o in obs:
                    by_frame.setdefault(o[0], []).append((fid, o[1], o[2], o[3], o[4], o[5] if len(o) > 5 else 0))
                ids_at.setdefault(t, []).append(fid)
                fid += 1

        def feed(f):  # the front end's delivery of frame f: camera by camera, camera 0 first
            for k in range(stream.K):
                o = [x for x in by_frame.get(f, []) if x[5] == k]
                if o:
                    up.tracks_append(time_of(f), [x[0] for x in o], np.full(len(o), k, np.int32), np.array([[x[1], x[2]] for x in o], np.float32),
                                     np.array([[x[3], x[4]] for x in o], np.float32))
        for f in range(C):
            feed(f)
    for t in range(C, stream.T):
        new, dR = _compose(last, stream.truth[t - 1], stream.truth[t])
        new = synth.boxplus_pose(new, stream.clone_noise[t])
        Phi = np.zeros((6, 6))
        Phi[:3, :3], Phi[3:, 3:] = dR, np.eye(3)
        Q = np.diag([stream.q_theta ** 2] * 3 + [stream.q_p ** 2] * 3)
        nid = up.state_augment_clone(base + 6 * (C - 1), new)            # StateHelper::clone of the newest pose
        up.state_propagate(nid, nid + np.arange(6), Phi, Q)              # EKFPropagation of the new block
        up.state_marginalize(base, 6)                                    # the oldest clone leaves the window
        frames = frames[1:] + [t]
        tracks = stream.tracks[t]
        if track_store:
            feed(t)
            ids = ids_at.get(t, [])
            have = len(ids) > 0
            if have:
                up.tracks_to_features(ids, [time_of(f) for f in frames])
        else:
            have = bool(tracks)
            if have:
                dummy = np.zeros((C, 7))
                dummy[:, 3] = 1.0
                up.set_features(_frame_problem(stream, tracks, frames, N, P, dummy, dummy, calib, intr))
        if have:
            out = up.update()
            used[t] = int(np.sum(out["feat_status"] == 0))
            last = out["clone_q_p"][-1].copy()
            if track_store:
                up.tracks_erase(ids)
        else:
            last = up.get_state(P=False)["clone_q_p"][-1].copy()
        est[t] = last.copy()
    ts = sorted(est)
    return dict(frames=np.array(ts), est=np.stack([est[t] for t in ts]), truth=stream.truth[ts], used=used)


def run(stream: Stream, update_fn=None):
    """Runs the sliding-window filter over the stream.  update_fn(prob) -> dict(P, clone_q_p, calib_q_p, intrinsics,
    feat_status); None = no updates (dead reckoning).  Returns per-frame estimates of the newest clone and the truth."""
    C = stream.C
    base, N, P, clones, calib, intr = _initial_state(stream)
    frames = list(range(C))
    fej = clones.copy()
    est, used = {frames[-1]: clones[-1].copy()}, {}
    for t in range(C, stream.T):
        # ---- clone the new pose (window grows to C + 1) ... then drop the oldest so that the update sees C clones
        new, dR = _compose(clones[-1], stream.truth[t - 1], stream.truth[t])
        new = synth.boxplus_pose(new, stream.clone_noise[t])
        J = np.zeros((6, N))
        J[:3, N - 6:N - 3] = dR
        J[3:, N - 3:N] = np.eye(3)
        Q = np.diag([stream.q_theta ** 2] * 3 + [stream.q_p ** 2] * 3)
        P = np.block([[P, P @ J.T], [J @ P, J @ P @ J.T + Q]])
        keep = np.r_[0:base, base + 6:N + 6]  # marginalise the oldest clone (StateHelper.cpp:271-339)
        P = P[np.ix_(keep, keep)]
        clones = np.vstack([clones[1:], new[None, :]])
        fej = np.vstack([fej[1:], new[None, :]])
        frames = frames[1:] + [t]
        # ---- MSCKF update with the tracks that ended at t - 1
        tracks = stream.tracks[t]
        if update_fn is not None and tracks:
            prob = _frame_problem(stream, tracks, frames, N, P, clones, fej, calib, intr)
            out = update_fn(prob)
            P = 0.5 * (out["P"] + out["P"].T)
            clones, calib, intr = out["clone_q_p"].copy(), out["calib_q_p"].copy(), out["intrinsics"].copy()
            used[t] = int(np.sum(out["feat_status"] == 0))
        est[t] = clones[-1].copy()
    ts = sorted(est)
    return dict(frames=np.array(ts), est=np.stack([est[t] for t in ts]), truth=stream.truth[ts], used=used)


def ate(res):
    """Absolute trajectory error of the newest-clone estimates, no alignment (estimate and truth share the frame in
    simulation, ov_eval/src/calc/ResultTrajectory.cpp:82-110): (mean orientation error in deg, mean position error in m)."""
    e_p = np.linalg.norm(res["est"][:, 4:] - res["truth"][:, 4:], axis=1)
    e_o = []
    for qe, qt in zip(res["est"][:, :4], res["truth"][:, :4]):
        R = _rot(qe).T @ _rot(qt)
        e_o.append(np.degrees(np.arccos(np.clip((np.trace(R) - 1) / 2, -1, 1))))
    return float(np.mean(e_o)), float(np.mean(e_p))
