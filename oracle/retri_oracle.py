"""ORACLE (test infrastructure, never shipped or imported by open_vins_amd/): numpy restatement of
VioManager::retriangulate_active_tracks (ov_msckf/src/core/VioManagerHelper.cpp:190-387, rpng/open_vins v2.7) for the MSCKF
(non-SLAM) tracks: the maps active_feat_linsys_A / _b / _count carried from frame to frame, statement for statement —
including what their bookkeeping does to a track seen by several cameras in one frame (:264-272).
Plain Python loops: small cases only.  Parity unpinned by the reference (it ships no test of this function); pinned by
tests/test_oracle_invariants.py::test_retriangulation_* (noise-free truth, drop-out, the multi-camera rule)."""
import numpy as np


def skew_x(w):  # ov_core/src/utils/quat_ops.h:135-139
    return np.array([[0.0, -w[2], w[1]], [w[2], 0.0, -w[0]], [-w[1], w[0], 0.0]])


def colpiv_qr_solve(A, b):
    """Eigen::ColPivHouseholderQR::solve for a square system (rank-revealing; least-squares equivalent)."""
    import scipy.linalg
    Q, R, piv = scipy.linalg.qr(A, pivoting=True)
    x = np.zeros(3)
    x[piv] = np.linalg.solve(R, Q.T @ b) if abs(R[2, 2]) > 0 else np.linalg.lstsq(R, Q.T @ b, rcond=None)[0]
    return x


class ActiveTracks:
    """The state VioManager keeps between frames (VioManager.h: active_feat_linsys_A, _b, _count, active_tracks_posinG, _uvd)."""

    def __init__(self, max_cond_number=10000.0, min_dist=0.10, max_dist=60.0):
        self.A, self.b, self.count = {}, {}, {}
        self.posinG, self.uvd = {}, {}
        self.max_cond_number, self.min_dist, self.max_dist = max_cond_number, min_dist, max_dist

    def frame(self, R_GtoI, p_IinG, cams, obs, img_w, img_h):
        """cams: list over cameras IN sensor_ids ORDER of (cam_id, R_ItoC, p_IinC); obs[cam_id] = list of (featid, (u, v) float32 pixel,
        (xn, yn) float32 normalised).  Returns (posinG, uvd) of this frame."""
        A_new, b_new, c_new, pos_new, uv_cam0 = {}, {}, {}, {}, {}
        for cam_id, R_ItoC, p_IinC in cams:
            R_GtoCi = R_ItoC @ R_GtoI                       # :226
            p_CiinG = p_IinG - R_GtoCi.T @ p_IinC           # :227
            for featid, pt_d, pt_n in obs.get(cam_id, []):
                if cam_id == 0:
                    uv_cam0[featid] = pt_d                  # :243-245
                b_i = R_GtoCi.T @ np.array([float(pt_n[0]), float(pt_n[1]), 1.0])  # :255-257
                b_i = b_i / np.linalg.norm(b_i)             # :258
                Bperp = skew_x(b_i)
                Ai = Bperp.T @ Bperp                        # :262
                bi = Ai @ p_CiinG                           # :263
                if featid not in self.A:                    # :264-267 (std::map::insert keeps an existing entry)
                    A_new.setdefault(featid, Ai)
                    b_new.setdefault(featid, bi)
                    c_new.setdefault(featid, 1)
                else:                                       # :268-272
                    A_new[featid] = Ai + self.A[featid]
                    b_new[featid] = bi + self.b[featid]
                    c_new[featid] = 1 + self.count[featid]
                if c_new[featid] > 3:                       # :275
                    A, b = A_new[featid], b_new[featid]
                    p_FinG = colpiv_qr_solve(A, b)          # :280
                    p_FinCi = R_GtoCi @ (p_FinG - p_CiinG)  # :281
                    sv = np.linalg.svd(A, compute_uv=False)  # :284-288
                    condA = sv[0] / sv[-1]
                    if abs(condA) <= self.max_cond_number and self.min_dist <= p_FinCi[2] <= self.max_dist and not np.isnan(np.linalg.norm(p_FinCi)):
                        pos_new[featid] = p_FinG            # :293-296
        self.A, self.b, self.count, self.posinG = A_new, b_new, c_new, pos_new  # :305-309
        # ---- :329-379 with cam0's calibration
        self.uvd = {}
        cam0 = [c for c in cams if c[0] == 0]
        if cam0:
            _, R_ItoC, p_IinC = cam0[0]
            for featid, p in self.posinG.items():
                if featid not in uv_cam0:                   # :349-350
                    continue
                p_FinCi = R_ItoC @ (R_GtoI @ (p - p_IinG)) + p_IinC  # :354-355
                depth = p_FinCi[2]
                ud, vd = float(uv_cam0[featid][0]), float(uv_cam0[featid][1])  # :358-359
                if depth < 0.1:                             # :367
                    continue
                if ud < 0 or int(ud) >= img_w or vd < 0 or int(vd) >= img_h:  # :374
                    continue
                self.uvd[featid] = np.array([ud, vd, depth])
        return self.posinG, self.uvd
