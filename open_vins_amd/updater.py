"""Thin Python binding of the C ABI, named after the reference classes it stands in for.

The host language of the reference is C++; the drop-in C++ shim (same class names and
signatures as ov_msckf::UpdaterMSCKF / ov_core::FeatureInitializer) lives in
open_vins_amd/shim/.  This module only exists so that tests and bench.py can drive the very
same C entry points from Python; it contains no arithmetic.
"""
from __future__ import annotations

import ctypes as C

import numpy as np

from . import capi


def _dp(a):
    return a.ctypes.data_as(capi.c_double_p) if a is not None else None


def _ip(a):
    return a.ctypes.data_as(capi.c_int32_p) if a is not None else None


class UpdaterMSCKF:
    """ov_msckf::UpdaterMSCKF (ov_msckf/src/update/UpdaterMSCKF.h:60-78) on one MI355X.

    ``options`` carries UpdaterOptions + FeatureInitializerOptions + the StateOptions subset
    (capi.default_options).  The constructor builds the chi2 table and the feature initializer
    exactly like UpdaterMSCKF.cpp:42-56 does — inside ovgpu_create.
    """

    def __init__(self, options: capi.Options | None = None, device: int = 0):
        self.lib = capi.load()
        self.options = options if options is not None else capi.default_options()
        self._ctx = C.c_void_p()
        capi.check(self.lib.ovgpu_create(C.byref(self.options), int(device), C.byref(self._ctx)), "ovgpu_create")
        self._views = None

    def close(self):
        if getattr(self, "_ctx", None) is not None and self._ctx:
            self.lib.ovgpu_destroy(self._ctx)
            self._ctx = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- inputs ---------------------------------------------------------
    def set_problem(self, prob):
        """Uploads state + features of a synth.Problem-like snapshot (kept resident in HBM)."""
        self._views = capi.Views(prob)
        capi.check(self.lib.ovgpu_set_state(self._ctx, C.byref(self._views.state)), "ovgpu_set_state")
        capi.check(self.lib.ovgpu_set_features(self._ctx, C.byref(self._views.features)), "ovgpu_set_features")
        self.F, self.N = self._views.features.F, self._views.state.N
        self.Cn, self.K = self._views.state.C, self._views.state.K

    def set_features(self, prob):
        v = capi.Views(prob)
        capi.check(self.lib.ovgpu_set_features(self._ctx, C.byref(v.features)), "ovgpu_set_features")
        self._views_feat = v
        self.F = v.features.F

    def reset_state(self):
        capi.check(self.lib.ovgpu_reset_state(self._ctx), "ovgpu_reset_state")

    def set_camera_poses_from(self, prob):
        """FeatureInitializer's own input (clonesCAM): the clone-camera poses computed by the caller (here: numpy, the
        formulas of UpdaterMSCKF.cpp:106-107) instead of a state snapshot; then the feature batch."""
        from . import synth
        Cn, K = prob.C, prob.K
        R = np.zeros((K * Cn, 9))
        p = np.zeros((K * Cn, 3))
        for k in range(K):
            R_ItoC, p_IinC = synth.quat_2_rot(prob.calib_q_p[k, :4]), prob.calib_q_p[k, 4:]
            for c in range(Cn):
                R_GtoC = R_ItoC @ synth.quat_2_rot(prob.clone_q_p[c, :4])
                R[k * Cn + c] = R_GtoC.reshape(-1)
                p[k * Cn + c] = prob.clone_q_p[c, 4:] - R_GtoC.T @ p_IinC
        self._poses = (np.ascontiguousarray(R), np.ascontiguousarray(p))
        capi.check(self.lib.ovgpu_set_camera_poses(self._ctx, Cn, K, _dp(self._poses[0]), _dp(self._poses[1])), "ovgpu_set_camera_poses")
        self.N, self.Cn, self.K = int(prob.N), Cn, K
        self.set_features(prob)

    # ---- ov_core::FeatureInitializer ------------------------------------
    def triangulate(self):
        F = self.F
        out = dict(p_FinA=np.zeros((F, 3)), p_FinG=np.zeros((F, 3)), anchor_meas=np.zeros(F, np.int32), status=np.zeros(F, np.int32))
        capi.check(self.lib.ovgpu_triangulate(self._ctx, _dp(out["p_FinA"]), _dp(out["p_FinG"]), _ip(out["anchor_meas"]), _ip(out["status"])),
                   "ovgpu_triangulate")
        return out

    def refine(self, p_FinA, anchor_meas):
        """FeatureInitializer::single_gaussnewton alone: refinement from the caller's positions / anchors."""
        F = self.F
        pa = np.ascontiguousarray(p_FinA, dtype=np.float64)
        am = np.ascontiguousarray(anchor_meas, dtype=np.int32)
        out = dict(p_FinA=np.zeros((F, 3)), p_FinG=np.zeros((F, 3)), status=np.zeros(F, np.int32))
        capi.check(self.lib.ovgpu_refine(self._ctx, _dp(pa), _ip(am), _dp(out["p_FinA"]), _dp(out["p_FinG"]), _ip(out["status"])), "ovgpu_refine")
        return out

    def get_triangulation(self):
        """What the triangulation stage of the last compress / update / delayed_init left on the device (no second pass)."""
        F = self.F
        out = dict(p_FinA=np.zeros((F, 3)), p_FinG=np.zeros((F, 3)), anchor_meas=np.zeros(F, np.int32))
        capi.check(self.lib.ovgpu_get_triangulation(self._ctx, _dp(out["p_FinA"]), _dp(out["p_FinG"]), _ip(out["anchor_meas"])), "ovgpu_get_triangulation")
        return out

    # ---- VioManager::retriangulate_active_tracks ------------------------
    def retriangulate(self, clone_index, featid, cam_idx, uv, uvn, cam0=0, img_w=752, img_h=480):
        """One camera frame of the running linear triangulation of the active tracks (SLAM features left out by the caller)."""
        featid = np.ascontiguousarray(featid, dtype=np.int64)
        cam_idx = np.ascontiguousarray(cam_idx, dtype=np.int32)
        uv = np.ascontiguousarray(uv, dtype=np.float32)
        uvn = np.ascontiguousarray(uvn, dtype=np.float32)
        n = len(featid)
        ids = np.zeros(max(n, 1), np.int64)
        pos, uvd = np.zeros((max(n, 1), 3)), np.zeros((max(n, 1), 3))
        nt = C.c_int32(0)
        capi.check(self.lib.ovgpu_retriangulate(self._ctx, int(clone_index), n, featid.ctypes.data_as(capi.c_int64_p), _ip(cam_idx),
                                                uv.ctypes.data_as(capi.c_float_p), uvn.ctypes.data_as(capi.c_float_p), int(cam0), int(img_w), int(img_h),
                                                C.byref(nt), ids.ctypes.data_as(capi.c_int64_p), _dp(pos), _dp(uvd)), "ovgpu_retriangulate")
        return dict(featid=ids[:nt.value].copy(), p_FinG=pos[:nt.value].copy(), uvd=uvd[:nt.value].copy())

    def retriangulate_reset(self):
        capi.check(self.lib.ovgpu_retriangulate_reset(self._ctx), "ovgpu_retriangulate_reset")

    def set_triangulation(self, p_FinG, p_FinA=None, anchor_meas=None, status=None):
        """Positions supplied by the caller (UpdaterSLAM::update path, stage-wise parity tests)."""
        self._given = [np.ascontiguousarray(p_FinG, dtype=np.float64),
                       np.ascontiguousarray(p_FinA, dtype=np.float64) if p_FinA is not None else None,
                       np.ascontiguousarray(anchor_meas, dtype=np.int32) if anchor_meas is not None else None,
                       np.ascontiguousarray(status, dtype=np.int32) if status is not None else None]
        g = self._given
        capi.check(self.lib.ovgpu_set_triangulation(self._ctx, _dp(g[1]), _dp(g[0]), _ip(g[2]), _ip(g[3])), "ovgpu_set_triangulation")

    # ---- UpdaterMSCKF::update -------------------------------------------
    def update(self, check=True):
        F, N = self.F, self.N
        out = dict(feat_status=np.zeros(F, np.int32), chi2=np.zeros(F), chi2_thresh=np.zeros(F), p_FinG=np.zeros((F, 3)),
                   dx=np.zeros(N), P=np.zeros((N, N)))
        stats = capi.UpdateStats()
        rc = self.lib.ovgpu_msckf_update(self._ctx, _ip(out["feat_status"]), _dp(out["chi2"]), _dp(out["chi2_thresh"]), _dp(out["p_FinG"]),
                                         _dp(out["dx"]), _dp(out["P"]), C.byref(stats))
        out["rc"] = rc
        if check:
            capi.check(rc, "ovgpu_msckf_update")
        out["stats"] = stats.as_dict()
        out["route"] = self.lib.ovgpu_last_update_route(self._ctx)  # capi.COMPRESS_GRAM / COMPRESS_TSQR (chosen, or the fallback)
        out.update(self.get_state(P=False))
        return out

    def compress(self):
        """Mode A: returns the compressed (H, r) for the stock StateHelper::EKFUpdate."""
        F = self.F
        Dmax = 6 * self.Cn + 14 * self.K
        out = dict(feat_status=np.zeros(F, np.int32), chi2=np.zeros(F), chi2_thresh=np.zeros(F), p_FinG=np.zeros((F, 3)))
        H = np.zeros((Dmax, Dmax))
        r = np.zeros(Dmax)
        cols = np.zeros(Dmax, np.int32)
        D, rows = C.c_int32(0), C.c_int32(0)
        stats = capi.UpdateStats()
        capi.check(self.lib.ovgpu_msckf_compress(self._ctx, _ip(out["feat_status"]), _dp(out["chi2"]), _dp(out["chi2_thresh"]), _dp(out["p_FinG"]),
                                                 C.byref(D), C.byref(rows), _ip(cols), _dp(H), _dp(r), C.byref(stats)), "ovgpu_msckf_compress")
        d, n = D.value, rows.value
        out["D"], out["rows"] = d, n
        out["H"] = np.ascontiguousarray(H.reshape(-1)[: n * d].reshape(n, d))
        out["r"] = r[:n].copy()
        out["col_cov_id"] = cols[:d].copy()
        out["stats"] = stats.as_dict()
        return out

    def get_state(self, P=True):
        out = dict(clone_q_p=np.zeros((self.Cn, 7)), calib_q_p=np.zeros((self.K, 7)), intrinsics=np.zeros((self.K, 8)))
        Pm = np.zeros((self.N, self.N)) if P else None
        capi.check(self.lib.ovgpu_get_state(self._ctx, _dp(Pm), _dp(out["clone_q_p"]), _dp(out["calib_q_p"]), _dp(out["intrinsics"])),
                   "ovgpu_get_state")
        if P:
            out["P"] = Pm
        return out

    # ---- UpdaterSLAM::update (UpdaterSLAM.cpp:253-479) ------------------
    def set_slam_problem(self, prob):
        """State, landmarks and tracks of a synth.make_slam_problem snapshot (order matters: the landmarks define the
        Jacobian columns and the row count of every feature)."""
        v = capi.Views(prob)
        self._views = v
        self.F, self.N, self.Cn, self.K = v.features.F, v.state.N, v.state.C, v.state.K
        capi.check(self.lib.ovgpu_set_state(self._ctx, C.byref(v.state)), "ovgpu_set_state")
        capi.check(self.lib.ovgpu_set_landmarks(self._ctx, C.byref(v.landmarks)), "ovgpu_set_landmarks")
        capi.check(self.lib.ovgpu_set_features(self._ctx, C.byref(v.features)), "ovgpu_set_features")

    def slam_update(self, lm_index=None):
        """lm_index [F]: the resident landmark each uploaded track observes (default: the snapshot's)."""
        v = self._views
        F, N = self.F, self.N
        Lc = C.c_int32(0)
        capi.check(self.lib.ovgpu_get_landmarks(self._ctx, C.byref(Lc), None, None, None, None, None), "ovgpu_get_landmarks")
        L = Lc.value
        lm_index = v.lm_index if lm_index is None else np.ascontiguousarray(lm_index, dtype=np.int32)
        out = dict(feat_status=np.zeros(F, np.int32), chi2=np.zeros(F), chi2_thresh=np.zeros(F), dx=np.zeros(N), P=np.zeros((N, N)),
                   landmarks=np.zeros((L, 3)))
        stats = capi.UpdateStats()
        rc = self.lib.ovgpu_slam_update(self._ctx, _ip(lm_index), _ip(out["feat_status"]), _dp(out["chi2"]), _dp(out["chi2_thresh"]),
                                        _dp(out["dx"]), _dp(out["P"]), _dp(out["landmarks"]), C.byref(stats))
        out["rc"] = rc
        capi.check(rc, "ovgpu_slam_update")
        out["stats"] = stats.as_dict()
        return out

    def slam_compress(self):
        """Mode A of the SLAM update: the compressed (H, r) incl. the landmark columns."""
        v = self._views
        F, L = self.F, v.landmarks.L
        Dmax = 6 * self.Cn + 14 * self.K + 3 * L
        out = dict(feat_status=np.zeros(F, np.int32), chi2=np.zeros(F), chi2_thresh=np.zeros(F))
        H, r, cols = np.zeros((Dmax, Dmax)), np.zeros(Dmax), np.zeros(Dmax, np.int32)
        D, rows = C.c_int32(0), C.c_int32(0)
        stats = capi.UpdateStats()
        capi.check(self.lib.ovgpu_slam_compress(self._ctx, _ip(v.lm_index), _ip(out["feat_status"]), _dp(out["chi2"]), _dp(out["chi2_thresh"]),
                                                C.byref(D), C.byref(rows), _ip(cols), _dp(H), _dp(r), C.byref(stats)), "ovgpu_slam_compress")
        d, n = D.value, rows.value
        out.update(D=d, rows=n, H=np.ascontiguousarray(H.reshape(-1)[: n * d].reshape(n, d)), r=r[:n].copy(), col_cov_id=cols[:d].copy(),
                   stats=stats.as_dict())
        return out

    # ---- UpdaterSLAM::delayed_init (UpdaterSLAM.cpp:61-251) -------------
    def delayed_init(self, feat_rep=0, feat_rep_each=None):
        """Runs the delayed initialisation on the resident state and tracks (set_problem / set_slam_problem first;
        set_triangulation optionally replaces the triangulation stage).  The state grows by 3 (1: single depth) per accepted
        feature.  feat_rep_each [F]: the representation per feature (UpdaterSLAM.cpp:160-166: feat_rep_aruco for ArUco corners)."""
        F, N = self.F, self.N
        reps = np.full(F, int(feat_rep), np.int32)
        if feat_rep_each is not None:
            reps = np.ascontiguousarray(feat_rep_each, dtype=np.int32)
            capi.check(self.lib.ovgpu_set_feature_reps(self._ctx, _ip(reps)), "ovgpu_set_feature_reps")
        Nmax = N + int(np.where(reps == capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE, 1, 3).sum())
        out = dict(feat_status=np.zeros(F, np.int32), chi2=np.zeros(F), chi2_thresh=np.zeros(F), lm_cov_id=np.zeros(F, np.int32),
                   lm_value=np.zeros((F, 3)), lm_fej=np.zeros((F, 3)), anchor_cam=np.zeros(F, np.int32), anchor_clone=np.zeros(F, np.int32),
                   dx_seq=np.zeros((F, Nmax)))
        Pbuf = np.zeros(Nmax * Nmax)
        N_out = C.c_int32(0)
        stats = capi.UpdateStats()
        rc = self.lib.ovgpu_slam_delayed_init(self._ctx, int(feat_rep), _ip(out["feat_status"]), _dp(out["chi2"]), _dp(out["chi2_thresh"]),
                                              _ip(out["lm_cov_id"]), _dp(out["lm_value"]), _dp(out["lm_fej"]), _ip(out["anchor_cam"]),
                                              _ip(out["anchor_clone"]), _dp(out["dx_seq"]), C.byref(N_out), _dp(Pbuf), C.byref(stats))
        out["rc"] = rc
        capi.check(rc, "ovgpu_slam_delayed_init")
        n = N_out.value
        out["N"] = n
        out["P"] = Pbuf[: n * n].reshape(n, n).copy()
        out["stats"] = stats.as_dict()
        self.N = n
        return out

    def get_landmarks(self):
        L = C.c_int32(0)
        capi.check(self.lib.ovgpu_get_landmarks(self._ctx, C.byref(L), None, None, None, None, None), "ovgpu_get_landmarks")
        n = L.value
        out = dict(value=np.zeros((n, 3)), fej=np.zeros((n, 3)), cov_id=np.zeros(n, np.int32), anchor_cam=np.zeros(n, np.int32),
                   anchor_clone=np.zeros(n, np.int32), feat_rep=np.zeros(n, np.int32))
        if n:
            capi.check(self.lib.ovgpu_get_landmark_reps(self._ctx, C.byref(L), _ip(out["feat_rep"])), "ovgpu_get_landmark_reps")
            capi.check(self.lib.ovgpu_get_landmarks(self._ctx, C.byref(L), _dp(out["value"]), _dp(out["fej"]), _ip(out["cov_id"]),
                                                    _ip(out["anchor_cam"]), _ip(out["anchor_clone"])), "ovgpu_get_landmarks")
        return out

    # ---- FeatureDatabase on the device (FeatureDatabase.cpp:59-126, Feature.cpp:26-53) ----
    def tracks_create(self, max_tracks, max_obs):
        capi.check(self.lib.ovgpu_tracks_create(self._ctx, int(max_tracks), int(max_obs)), "ovgpu_tracks_create")

    def tracks_append(self, timestamp, featid, cam_id, uv, uvn):
        ids = np.ascontiguousarray(featid, dtype=np.int64)
        cam = np.ascontiguousarray(cam_id, dtype=np.int32)
        uv = np.ascontiguousarray(uv, dtype=np.float32)
        uvn = np.ascontiguousarray(uvn, dtype=np.float32)
        rc = self.lib.ovgpu_tracks_append(self._ctx, float(timestamp), len(ids), ids.ctypes.data_as(C.POINTER(C.c_int64)), _ip(cam),
                                          uv.ctypes.data_as(capi.c_float_p), uvn.ctypes.data_as(capi.c_float_p))
        capi.check(rc, "ovgpu_tracks_append")

    def tracks_erase(self, featid):
        ids = np.ascontiguousarray(featid, dtype=np.int64)
        capi.check(self.lib.ovgpu_tracks_erase(self._ctx, len(ids), ids.ctypes.data_as(C.POINTER(C.c_int64))), "ovgpu_tracks_erase")

    def _tracks_ids(self, fn, name, timestamp):
        n = C.c_int32(0)
        capi.check(fn(self._ctx, float(timestamp), 0, None, C.byref(n)), name)
        ids = np.zeros(max(n.value, 1), np.int64)
        capi.check(fn(self._ctx, float(timestamp), n.value, ids.ctypes.data_as(C.POINTER(C.c_int64)), C.byref(n)), name)
        return ids[: n.value]

    def tracks_not_containing_newer(self, timestamp):
        """FeatureDatabase::features_not_containing_newer (FeatureDatabase.cpp:87-126): ids, ascending."""
        return self._tracks_ids(self.lib.ovgpu_tracks_not_containing_newer, "ovgpu_tracks_not_containing_newer", timestamp)

    def tracks_containing_older(self, timestamp):
        """FeatureDatabase::features_containing_older (FeatureDatabase.cpp:128-167)."""
        return self._tracks_ids(self.lib.ovgpu_tracks_containing_older, "ovgpu_tracks_containing_older", timestamp)

    def tracks_containing(self, timestamp):
        """FeatureDatabase::features_containing (FeatureDatabase.cpp:169-209)."""
        return self._tracks_ids(self.lib.ovgpu_tracks_containing, "ovgpu_tracks_containing", timestamp)

    def tracks_oldest_timestamp(self):
        """FeatureDatabase::get_oldest_timestamp (FeatureDatabase.cpp:265-276): -1 for an empty store."""
        t = C.c_double(0.0)
        capi.check(self.lib.ovgpu_tracks_oldest_timestamp(self._ctx, C.byref(t)), "ovgpu_tracks_oldest_timestamp")
        return t.value

    def tracks_cleanup_measurements(self, timestamp, exact=False):
        """FeatureDatabase::cleanup_measurements / cleanup_measurements_exact (FeatureDatabase.cpp:226-263): returns how many tracks
        were left without observations and dropped."""
        n = C.c_int32(0)
        fn = self.lib.ovgpu_tracks_cleanup_measurements_exact if exact else self.lib.ovgpu_tracks_cleanup_measurements
        capi.check(fn(self._ctx, float(timestamp), C.byref(n)), "ovgpu_tracks_cleanup_measurements")
        return n.value

    def tracks_get_feature(self, featid):
        """FeatureDatabase::get_feature_clone (FeatureDatabase.cpp:41-57): the stored observations of one track in append order, or
        None for an unknown id."""
        n = C.c_int32(0)
        capi.check(self.lib.ovgpu_tracks_get_feature(self._ctx, int(featid), 0, C.byref(n), None, None, None, None), "ovgpu_tracks_get_feature")
        k = n.value
        if k == 0:
            return None
        out = dict(timestamps=np.zeros(k), cam_id=np.zeros(k, np.int32), uv=np.zeros((k, 2), np.float32), uvn=np.zeros((k, 2), np.float32))
        capi.check(self.lib.ovgpu_tracks_get_feature(self._ctx, int(featid), k, C.byref(n), _dp(out["timestamps"]), _ip(out["cam_id"]),
                                                     out["uv"].ctypes.data_as(capi.c_float_p), out["uvn"].ctypes.data_as(capi.c_float_p)),
                   "ovgpu_tracks_get_feature")
        return out

    def tracks_count(self):
        n = C.c_int32(0)
        capi.check(self.lib.ovgpu_tracks_count(self._ctx, C.byref(n)), "ovgpu_tracks_count")
        return n.value

    def tracks_group_order(self, order=capi.GROUPS_REFERENCE):
        """Camera-group order of the assembled batch (include/ovgpu.h): GROUPS_REFERENCE (default) is the reference's iteration
        order of Feature::timestamps — reverse order of first insertion — which decides the anchor of a track with tied counts."""
        capi.check(self.lib.ovgpu_tracks_group_order(self._ctx, int(order)), "ovgpu_tracks_group_order")

    def tracks_to_features(self, featid, clone_times):
        ids = np.ascontiguousarray(featid, dtype=np.int64)
        ct = np.ascontiguousarray(clone_times, dtype=np.float64)
        capi.check(self.lib.ovgpu_tracks_to_features(self._ctx, len(ids), ids.ctypes.data_as(C.POINTER(C.c_int64)), _dp(ct)), "ovgpu_tracks_to_features")
        self.F = len(ids)

    def set_feature_options(self, sigma_pix=None, chi2_multipler=None):
        """Per-feature sigma_pix / chi2 multiplier of the resident batch (UpdaterSLAM's ArUco options, UpdaterSLAM.cpp:392-409);
        None keeps the context's value.  Cleared by the next feature upload."""
        def arr(a):
            return None if a is None else np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (self.F,)))
        s, m = arr(sigma_pix), arr(chi2_multipler)
        capi.check(self.lib.ovgpu_set_feature_options(self._ctx, _dp(s) if s is not None else None, _dp(m) if m is not None else None),
                   "ovgpu_set_feature_options")

    def get_features(self):
        F, M = C.c_int32(0), C.c_int32(0)
        capi.check(self.lib.ovgpu_get_features(self._ctx, C.byref(F), C.byref(M), None, None, None, None, None), "ovgpu_get_features")
        f, m = F.value, M.value
        out = dict(meas_offsets=np.zeros(f + 1, np.int32), uv=np.zeros(2 * m, np.float32), uvn=np.zeros(2 * m, np.float32),
                   clone_idx=np.zeros(m, np.int32), cam_idx=np.zeros(m, np.int32))
        capi.check(self.lib.ovgpu_get_features(self._ctx, C.byref(F), C.byref(M), _ip(out["meas_offsets"]), out["uv"].ctypes.data_as(capi.c_float_p),
                                               out["uvn"].ctypes.data_as(capi.c_float_p), _ip(out["clone_idx"]), _ip(out["cam_idx"])), "ovgpu_get_features")
        return out

    # ---- UpdaterSLAM::change_anchors / perform_anchor_change (UpdaterSLAM.cpp:481-647) ----
    def change_anchor(self, lm_index, new_cam, new_clone):
        capi.check(self.lib.ovgpu_slam_change_anchor(self._ctx, int(lm_index), int(new_cam), int(new_clone)), "ovgpu_slam_change_anchor")

    def change_anchors(self, marg_clone, new_clone):
        n = C.c_int32(0)
        capi.check(self.lib.ovgpu_slam_change_anchors(self._ctx, int(marg_clone), int(new_clone), C.byref(n)), "ovgpu_slam_change_anchors")
        return n.value

    # ---- window bookkeeping on the resident covariance (StateHelper::marginalize / clone / EKFPropagation) ----
    def _refresh_dims(self):
        n, c = C.c_int32(0), C.c_int32(0)
        capi.check(self.lib.ovgpu_state_dims(self._ctx, C.byref(n), C.byref(c)), "ovgpu_state_dims")
        self.N, self.Cn = n.value, c.value

    def state_marginalize(self, cov_id, size):
        capi.check(self.lib.ovgpu_state_marginalize(self._ctx, int(cov_id), int(size)), "ovgpu_state_marginalize")
        self._refresh_dims()

    def state_augment_clone(self, src_cov_id, q_p, q_p_fej=None, dt_cov_id=-1, dnc_dt=None):
        q = np.ascontiguousarray(q_p, dtype=np.float64)
        qf = np.ascontiguousarray(q_p_fej if q_p_fej is not None else q_p, dtype=np.float64)
        d = np.ascontiguousarray(dnc_dt, dtype=np.float64) if dnc_dt is not None else None
        new_id = C.c_int32(0)
        capi.check(self.lib.ovgpu_state_augment_clone(self._ctx, int(src_cov_id), _dp(q), _dp(qf), int(dt_cov_id), _dp(d), C.byref(new_id)),
                   "ovgpu_state_augment_clone")
        self._refresh_dims()
        return new_id.value

    def state_propagate(self, new_cov_id, old_cov_ids, Phi, Q):
        Phi = np.ascontiguousarray(Phi, dtype=np.float64)
        Q = np.ascontiguousarray(Q, dtype=np.float64)
        ids = np.ascontiguousarray(old_cov_ids, dtype=np.int32)
        rc = self.lib.ovgpu_state_propagate(self._ctx, int(new_cov_id), Phi.shape[0], Phi.shape[1], _ip(ids), _dp(Phi), _dp(Q))
        capi.check(rc, "ovgpu_state_propagate")

    def marginal_covariance(self, cov_idx):
        idx = np.ascontiguousarray(cov_idx, dtype=np.int32)
        out = np.zeros((len(idx), len(idx)))
        capi.check(self.lib.ovgpu_state_marginal_covariance(self._ctx, len(idx), _ip(idx), _dp(out)), "ovgpu_state_marginal_covariance")
        return out

    def zupt(self, H, res, cov_idx, Q_bias, noise_mult, chi2_multipler=1.0, apply=True):
        """The linear algebra of UpdaterZeroVelocity::try_update (UpdaterZeroVelocity.cpp:183-203, :266-277) on the resident state:
        H [m x 9] over the dofs cov_idx (orientation 3, gyro bias 3, accelerometer bias 3), whitened residual res.  Returns chi2,
        its threshold and — when it passes and apply is set — dx, P after the bias propagation and the update."""
        Hc, rc = self.measurement_compress(H, res)                       # :183-186
        Pm = self.marginal_covariance(cov_idx)                            # :193
        Pm[3:9, 3:9] += Q_bias                                            # :194-196
        S = Hc @ Pm @ Hc.T + noise_mult * np.eye(len(rc))                 # :197
        chi2 = float(rc @ np.linalg.solve(S, rc))                         # :198
        thr = chi2_multipler * float(self.lib.ovgpu_chi2_quantile_95(len(rc)))  # :201-208
        out = dict(chi2=chi2, chi2_thresh=thr, rows=len(rc), accepted=chi2 <= thr)
        if out["accepted"] and apply:
            self.state_propagate(int(cov_idx[3]), list(cov_idx[3:9]), np.eye(6), Q_bias)  # :268-274
            out["dx"], out["P"] = self.ekf_update(Hc, rc, cov_idx, noise_mult)           # :277
        return out

    # ---- standalone helpers (UpdaterHelper::measurement_compress_inplace, StateHelper::EKFUpdate) ----
    def measurement_compress(self, H, res):
        H = np.ascontiguousarray(H, dtype=np.float64)
        res = np.ascontiguousarray(res, dtype=np.float64)
        rows, cols = H.shape
        n = min(rows, cols)
        Ho, ro = np.zeros((max(n, 1), cols)), np.zeros(max(n, 1))
        rows_out = C.c_int32(0)
        capi.check(self.lib.ovgpu_measurement_compress(self._ctx, rows, cols, _dp(H), _dp(res), _dp(Ho), _dp(ro), C.byref(rows_out)),
                   "ovgpu_measurement_compress")
        return Ho[: rows_out.value], ro[: rows_out.value]

    def ekf_update(self, H, res, col_cov_id, sigma2):
        H = np.ascontiguousarray(H, dtype=np.float64)
        res = np.ascontiguousarray(res, dtype=np.float64)
        cols = np.ascontiguousarray(col_cov_id, dtype=np.int32)
        dx, P = np.zeros(self.N), np.zeros((self.N, self.N))
        capi.check(self.lib.ovgpu_ekf_update(self._ctx, H.shape[0], H.shape[1], _ip(cols), _dp(H), _dp(res), float(sigma2), _dp(dx), _dp(P)),
                   "ovgpu_ekf_update")
        return dx, P

    # ---- feature-sharded multi-GPU update ----------------
    def triangle_len(self):
        n = C.c_int64(0)
        capi.check(self.lib.ovgpu_triangle_len(self._ctx, C.byref(n)), "ovgpu_triangle_len")
        return n.value

    def local(self, tri_dev_ptr=None, want_outputs=True):
        F = self.F
        out = dict(feat_status=np.zeros(F, np.int32), chi2=np.zeros(F), chi2_thresh=np.zeros(F), p_FinG=np.zeros((F, 3)))
        stats = capi.UpdateStats()
        if want_outputs:
            rc = self.lib.ovgpu_msckf_local(self._ctx, _ip(out["feat_status"]), _dp(out["chi2"]), _dp(out["chi2_thresh"]), _dp(out["p_FinG"]),
                                            C.c_void_p(tri_dev_ptr) if tri_dev_ptr else None, C.byref(stats))
        else:
            rc = self.lib.ovgpu_msckf_local(self._ctx, None, None, None, None, C.c_void_p(tri_dev_ptr) if tri_dev_ptr else None, None)
        capi.check(rc, "ovgpu_msckf_local")
        out["stats"] = stats.as_dict()
        return out

    def merge_update(self, tris_dev_ptr, G, want_outputs=True):
        N = self.N
        out = dict(dx=np.zeros(N), P=np.zeros((N, N)))
        stats = capi.UpdateStats()
        if want_outputs:
            rc = self.lib.ovgpu_msckf_merge_update(self._ctx, C.c_void_p(tris_dev_ptr), int(G), _dp(out["dx"]), _dp(out["P"]), C.byref(stats))
        else:
            rc = self.lib.ovgpu_msckf_merge_update(self._ctx, C.c_void_p(tris_dev_ptr), int(G), None, None, None)
        capi.check(rc, "ovgpu_msckf_merge_update")
        out["stats"] = stats.as_dict()
        return out

    # the same exchange in Gram form: one all-reduce (sum) instead of an all-gather + merge
    def gram_len(self):
        """Doubles of the Gram buffer (16 ceil((D+1)/16) squared + the accepted-row count), or 0 when this state does not
        fit the Gram route (D > 383) or options.compress_route selects the Householder exchange."""
        if self.options.compress_route != capi.COMPRESS_GRAM or self.triangle_len() > 383 * 384:
            return 0
        n = C.c_int64(0)
        capi.check(self.lib.ovgpu_gram_len(self._ctx, C.byref(n)), "ovgpu_gram_len")
        return n.value

    def local_gram(self, gram_dev_ptr, want_outputs=True):
        F = self.F
        out = dict(feat_status=np.zeros(F, np.int32), chi2=np.zeros(F), chi2_thresh=np.zeros(F), p_FinG=np.zeros((F, 3)))
        stats = capi.UpdateStats()
        if want_outputs:
            rc = self.lib.ovgpu_msckf_local_gram(self._ctx, _ip(out["feat_status"]), _dp(out["chi2"]), _dp(out["chi2_thresh"]), _dp(out["p_FinG"]),
                                                 C.c_void_p(gram_dev_ptr), C.byref(stats))
        else:
            rc = self.lib.ovgpu_msckf_local_gram(self._ctx, None, None, None, None, C.c_void_p(gram_dev_ptr), None)
        capi.check(rc, "ovgpu_msckf_local_gram")
        out["stats"] = stats.as_dict()
        return out

    def gram_update(self, gram_dev_ptr, want_outputs=True):
        N = self.N
        out = dict(dx=np.zeros(N), P=np.zeros((N, N)))
        stats = capi.UpdateStats()
        if want_outputs:
            rc = self.lib.ovgpu_msckf_gram_update(self._ctx, C.c_void_p(gram_dev_ptr), _dp(out["dx"]), _dp(out["P"]), C.byref(stats))
        else:
            rc = self.lib.ovgpu_msckf_gram_update(self._ctx, C.c_void_p(gram_dev_ptr), None, None, None)
        capi.check(rc, "ovgpu_msckf_gram_update")
        out["stats"] = stats.as_dict()
        return out

    # ---- benchmarking hooks ---------------------------------------------
    # ---- native multi-GPU exchange (RCCL inside the library, on the update's own stream)
    def comm_init(self, dist, device):
        """Joins this context to an RCCL communicator of dist.get_world_size() ranks: rank 0 draws the id, torch.distributed carries
        the 128 bytes (the host program's own transport; the C ABI only needs the bytes)."""
        import torch
        rank, world = dist.get_rank(), dist.get_world_size()
        idbuf = C.create_string_buffer(128)
        if rank == 0:
            capi.check(self.lib.ovgpu_comm_unique_id(idbuf), "ovgpu_comm_unique_id")
        t = torch.tensor(list(idbuf.raw), dtype=torch.uint8, device=device)
        dist.broadcast(t, src=0)
        idbuf = C.create_string_buffer(bytes(t.cpu().tolist()), 128)
        capi.check(self.lib.ovgpu_comm_init_rank(self._ctx, idbuf, rank, world), "ovgpu_comm_init_rank")

    def comm_info(self):
        """rank / world as told to ovgpu_comm_init_rank, and as the RCCL communicator reports them (-1: no communicator)."""
        v = [C.c_int32(0) for _ in range(4)]
        capi.check(self.lib.ovgpu_comm_info(self._ctx, *[C.byref(x) for x in v]), "ovgpu_comm_info")
        return dict(rank=v[0].value, world=v[1].value, rccl_rank=v[2].value, rccl_ranks=v[3].value)

    def comm_init_single(self):
        """A communicator of one rank (no collective is issued): the native sharded entry points on a single GPU."""
        idbuf = C.create_string_buffer(128)
        capi.check(self.lib.ovgpu_comm_unique_id(idbuf), "ovgpu_comm_unique_id")
        capi.check(self.lib.ovgpu_comm_init_rank(self._ctx, idbuf, 0, 1), "ovgpu_comm_init_rank")

    def update_sharded_async(self):
        capi.check(self.lib.ovgpu_msckf_update_sharded_async(self._ctx), "ovgpu_msckf_update_sharded_async")

    def update_sharded(self):
        F, N = self.F, self.N
        out = dict(feat_status=np.zeros(F, np.int32), chi2=np.zeros(F), chi2_thresh=np.zeros(F), p_FinG=np.zeros((F, 3)),
                   dx=np.zeros(N), P=np.zeros((N, N)))
        stats = capi.UpdateStats()
        capi.check(self.lib.ovgpu_msckf_update_sharded(self._ctx, _ip(out["feat_status"]), _dp(out["chi2"]), _dp(out["chi2_thresh"]), _dp(out["p_FinG"]),
                                                       _dp(out["dx"]), _dp(out["P"]), C.byref(stats)), "ovgpu_msckf_update_sharded")
        out["stats"] = stats.as_dict()
        out.update(self.get_state(P=False))
        return out

    def update_async(self):
        capi.check(self.lib.ovgpu_msckf_update_async(self._ctx), "ovgpu_msckf_update_async")

    def synchronize(self):
        capi.check(self.lib.ovgpu_synchronize(self._ctx), "ovgpu_synchronize")

    def debug_option(self, name, value=-1):
        """ovgpu_debug_option: returns the old value; value >= 0 sets it."""
        old = C.c_int64(0)
        capi.check(self.lib.ovgpu_debug_option(self._ctx, name.encode(), int(value), C.byref(old)), "ovgpu_debug_option")
        return old.value

    def kernel_times(self, reset=True):
        a, b, n = C.c_double(0), C.c_double(0), C.c_int64(0)
        sy, ns = C.c_double(0), C.c_int64(0)
        capi.check(self.lib.ovgpu_system_time(self._ctx, C.byref(sy), C.byref(ns)), "ovgpu_system_time")
        capi.check(self.lib.ovgpu_kernel_times(self._ctx, 1 if reset else 0, C.byref(a), C.byref(b), C.byref(n)), "ovgpu_kernel_times")
        return dict(ms_compress=a.value, ms_update=b.value, launches=n.value, ms_system=sy.value)


class MultiUpdater:
    """ovgpu_multi_*: ONE host process driving several GPUs (the reference's host is one C++ process); devices = list of HIP device
    indices.  A repeated index puts several ranks on one GPU with the library's loop-back collective instead of RCCL (which refuses
    two ranks on one GPU): how the G > 1 code paths are exercised on a one-GPU box."""

    def __init__(self, options=None, devices=(0,)):
        self.lib = capi.load()
        self.options = options if options is not None else capi.default_options()
        self._m = C.c_void_p()
        devs = np.ascontiguousarray(devices, dtype=np.int32)
        capi.check(self.lib.ovgpu_multi_create(C.byref(self.options), len(devs), _ip(devs), C.byref(self._m)), "ovgpu_multi_create")
        self._views = None

    def close(self):
        if getattr(self, "_m", None) is not None and self._m:
            self.lib.ovgpu_multi_destroy(self._m)
            self._m = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_problem(self, prob):
        self._views = capi.Views(prob)
        capi.check(self.lib.ovgpu_multi_set_state(self._m, C.byref(self._views.state)), "ovgpu_multi_set_state")
        capi.check(self.lib.ovgpu_multi_set_features(self._m, C.byref(self._views.features)), "ovgpu_multi_set_features")
        self.F, self.N = self._views.features.F, self._views.state.N

    def update(self):
        F, N = self.F, self.N
        out = dict(feat_status=np.zeros(F, np.int32), chi2=np.zeros(F), chi2_thresh=np.zeros(F), p_FinG=np.zeros((F, 3)),
                   dx=np.zeros(N), P=np.zeros((N, N)))
        stats = capi.UpdateStats()
        capi.check(self.lib.ovgpu_multi_msckf_update(self._m, _ip(out["feat_status"]), _dp(out["chi2"]), _dp(out["chi2_thresh"]), _dp(out["p_FinG"]),
                                                     _dp(out["dx"]), _dp(out["P"]), C.byref(stats)), "ovgpu_multi_msckf_update")
        out["stats"] = stats.as_dict()
        return out
