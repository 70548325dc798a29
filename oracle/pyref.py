"""ctypes loader of oracle/_ref/libov_ref.so: the REFERENCE'S OWN update-path sources (rpng/open_vins v2.7) compiled from
/root/reference against the stand-in Eigen / Boost / OpenCV headers of oracle/ref/standin, behind the C driver
oracle/ref/ref_driver.cpp.

TEST INFRASTRUCTURE ONLY: tests/ use it to pin oracle/ov_oracle.cpp (and, through the fixtures it generates under
tests/golden/, the HIP library).  It can only be BUILT where /root/reference exists (this container); the built .so is
git-ignored but travels to the GPU box with the snapshot.  `available()` says whether it can be used here.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from open_vins_amd import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_ref", "libov_ref.so")
REFERENCE = "/root/reference"
_lib = None
dp, ip = capi.c_double_p, capi.c_int32_p


def can_build():
    return os.path.isdir(os.path.join(REFERENCE, "ov_msckf", "src"))


def build():
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(_HERE, "ref")])


def available():
    return os.path.exists(LIB_PATH) or can_build()


def _open(path):
    lib = C.CDLL(path)
    lib.ref_chi2_quantile_95.restype = C.c_double
    lib.ref_chi2_quantile_95.argtypes = [C.c_int]
    return lib


def load():
    global _lib
    if _lib is not None:
        return _lib
    if can_build():
        build()  # make: no-op when up to date
    _lib = _open(LIB_PATH)
    return _lib


# ---- the DROP-IN build: open_vins_amd/shim's translation units compiled against the reference's own headers and linked with the
# reference's own State / StateHelper / types objects + libovgpu behind the SAME C driver (oracle/ref/Makefile: `dropin`).  Mode "a":
# the stock StateHelper::EKFUpdate applies the compressed system; "b": -DOVGPU_SHIM_MODE_B.  Every function of this module runs through
# that library inside `with pyref.using(pyref.dropin_path("a")):` -- the updates then need a GPU (the shims have no CPU fallback).
def dropin_path(mode):
    return os.path.join(_HERE, "_ref", f"libov_dropin_{mode}.so")


def build_dropin(target="dropin"):
    """target: "dropin" (links libovgpu) or "dropin_cpu" (links the oracle-backed test double tests/fake_ovgpu: mode names "a_cpu" / "b_cpu")."""
    if target == "dropin_cpu":
        subprocess.check_call(["make", "-s", "-C", _HERE])  # oracle/libov_oracle.so
    subprocess.check_call(["make", "-s", "-j8", "-C", os.path.join(_HERE, "ref"), target])


def dropin_available(mode="a"):
    return os.path.exists(dropin_path(mode)) or can_build()


class using:
    """Context manager: the wrappers of this module call into another build of the driver (a drop-in library)."""

    def __init__(self, path):
        self.path = path

    def __enter__(self):
        global _lib
        if not self.path.endswith("_cpu.so"):  # (the *_cpu builds link tests/fake_ovgpu, the oracle-backed double of the C ABI, instead)
            capi.load()  # libovgpu (and torch's HIP runtime before it) first: the drop-in library links against it
        self.saved = _lib
        _lib = _open(self.path)
        return _lib

    def __exit__(self, *exc):
        global _lib
        _lib = self.saved
        return False


def _p(a):
    return a.ctypes.data_as(dp) if a is not None else None


def _pi(a):
    return a.ctypes.data_as(ip) if a is not None else None


def chi2_quantile_95(dof):
    return load().ref_chi2_quantile_95(int(dof))


def make_givens(p, q):
    c, s = C.c_double(0), C.c_double(0)
    load().ref_make_givens(C.c_double(p), C.c_double(q), C.byref(c), C.byref(s))
    return c.value, s.value


def cam_distort(cam_d, is_fisheye, uv_norm):
    cam_d = np.ascontiguousarray(cam_d, dtype=np.float64)
    zn = np.ascontiguousarray(uv_norm, dtype=np.float64)
    uv, a, b = np.zeros(2), np.zeros((2, 2)), np.zeros((2, 8))
    load().ref_cam_distort(_p(cam_d), C.c_int(int(is_fisheye)), _p(zn), _p(uv), _p(a), _p(b))
    return uv, a, b


def cam_undistort(cam_d, is_fisheye, uv_dist):
    cam_d = np.ascontiguousarray(cam_d, dtype=np.float64)
    uv = np.ascontiguousarray(uv_dist, dtype=np.float64)
    out = np.zeros(2)
    load().ref_cam_undistort(_p(cam_d), C.c_int(int(is_fisheye)), _p(uv), _p(out))
    return out


def nullspace_project(H_f, H_x, res):
    H_f = np.ascontiguousarray(H_f, dtype=np.float64).copy()
    H_x = np.ascontiguousarray(H_x, dtype=np.float64).copy()
    res = np.ascontiguousarray(res, dtype=np.float64).copy()
    rows, nf = H_f.shape
    load().ref_nullspace_project(_p(H_f), _p(H_x), _p(res), C.c_int(rows), C.c_int(nf), C.c_int(H_x.shape[1]))
    return H_x[nf:].copy(), res[nf:].copy()


def measurement_compress(H_x, res):
    H_x = np.ascontiguousarray(H_x, dtype=np.float64).copy()
    res = np.ascontiguousarray(res, dtype=np.float64).copy()
    lib = load()
    lib.ref_measurement_compress.restype = C.c_int
    r = lib.ref_measurement_compress(_p(H_x), _p(res), C.c_int(H_x.shape[0]), C.c_int(H_x.shape[1]))
    return H_x[:r].copy(), res[:r].copy()


def triangulate(opts, views):
    F = views.features.F
    out = dict(p_FinA=np.zeros((F, 3)), p_FinG=np.zeros((F, 3)), anchor_cam=np.zeros(F, dtype=np.int32),
               anchor_clone=np.zeros(F, dtype=np.int32), status=np.zeros(F, dtype=np.int32))
    rc = load().ref_triangulate(C.byref(opts), C.byref(views.state), C.byref(views.features), _p(out["p_FinA"]), _p(out["p_FinG"]),
                                _pi(out["anchor_cam"]), _pi(out["anchor_clone"]), _pi(out["status"]))
    assert rc == 0
    return out


def feature_jacobian(opts, views, f, rep, p_FinG, p_FinA=None, anchor_cam=-1, anchor_clone=-1):
    """H_f [2m x 3], H_x [2m x N] (view covariance index space), res [2m] of UpdaterHelper::get_feature_jacobian_full."""
    m = int(views.meas_offsets[f + 1] - views.meas_offsets[f])
    N = views.state.N
    H_f, H_x, res = np.zeros((2 * m, 3)), np.zeros((2 * m, N)), np.zeros(2 * m)
    pG = np.ascontiguousarray(p_FinG, dtype=np.float64)
    pA = np.ascontiguousarray(p_FinA if p_FinA is not None else np.zeros(3), dtype=np.float64)
    nf = load().ref_feature_jacobian(C.byref(opts), C.byref(views.state), C.byref(views.features), C.c_int(int(f)), C.c_int(int(rep)), _p(pG), _p(pA),
                                     C.c_int(int(anchor_cam)), C.c_int(int(anchor_clone)), _p(H_f), _p(H_x), _p(res))
    if nf != 3:
        H_f = np.ascontiguousarray(H_f.reshape(-1)[: 2 * m * nf].reshape(2 * m, nf))
    return H_f, H_x, res


def msckf_update(opts, views):
    F, N, Cn, K = views.features.F, views.state.N, views.state.C, views.state.K
    out = dict(feat_status=np.zeros(F, dtype=np.int32), p_FinG=np.zeros((F, 3)), dx=np.zeros(N), P=np.zeros((N, N)),
               clone_q_p=np.zeros((Cn, 7)), calib_q_p=np.zeros((K, 7)), intrinsics=np.zeros((K, 8)))
    rc = load().ref_msckf_update(C.byref(opts), C.byref(views.state), C.byref(views.features), _pi(out["feat_status"]), _p(out["p_FinG"]),
                                 _p(out["dx"]), _p(out["P"]), _p(out["clone_q_p"]), _p(out["calib_q_p"]), _p(out["intrinsics"]))
    assert rc == 0
    return out


def _aruco(F, feat_sigma, feat_chi2mult, opts):
    """Per-feature (sigma, multiplier) arrays -> the reference's two option sets: features that differ from opts are ArUco."""
    if feat_sigma is None and feat_chi2mult is None:
        return None, 1.0, 1.0
    s = np.broadcast_to(np.asarray(opts.sigma_pix if feat_sigma is None else feat_sigma, dtype=np.float64), (F,))
    m = np.broadcast_to(np.asarray(opts.chi2_multipler if feat_chi2mult is None else feat_chi2mult, dtype=np.float64), (F,))
    is_aruco = np.ascontiguousarray(((s != opts.sigma_pix) | (m != opts.chi2_multipler)).astype(np.int32))
    if not is_aruco.any():
        return is_aruco, 1.0, 1.0
    sa, ma = np.unique(s[is_aruco == 1]), np.unique(m[is_aruco == 1])
    assert len(sa) == 1 and len(ma) == 1, "the reference has ONE ArUco option set"
    return is_aruco, float(sa[0]), float(ma[0])


def slam_update(opts, views, feat_sigma=None, feat_chi2mult=None):
    F, N, Cn, K, L = views.features.F, views.state.N, views.state.C, views.state.K, views.landmarks.L
    out = dict(feat_status=np.zeros(F, dtype=np.int32), dx=np.zeros(N), P=np.zeros((N, N)), landmarks=np.zeros((L, 3)),
               clone_q_p=np.zeros((Cn, 7)), calib_q_p=np.zeros((K, 7)), intrinsics=np.zeros((K, 8)))
    is_aruco, sa, ma = _aruco(F, feat_sigma, feat_chi2mult, opts)
    rc = load().ref_slam_update(C.byref(opts), C.byref(views.state), C.byref(views.landmarks), C.byref(views.features), _pi(views.lm_index),
                                _pi(is_aruco), C.c_double(sa), C.c_double(ma), _pi(out["feat_status"]), _p(out["dx"]), _p(out["P"]),
                                _p(out["landmarks"]), _p(out["clone_q_p"]), _p(out["calib_q_p"]), _p(out["intrinsics"]))
    assert rc == 0
    return out


def slam_delayed_init(opts, views, feat_rep=0, feat_sigma=None, feat_chi2mult=None, feat_rep_aruco=-1, feat_is_aruco=None):
    """feat_rep_aruco >= 0: StateOptions::feat_rep_aruco — the features that carry their own sigma / multiplier (the ArUco corners) are
    initialised in it, the others in feat_rep (UpdaterSLAM.cpp:160-166)."""
    F, N, Cn, K = views.features.F, views.state.N, views.state.C, views.state.K
    L0 = views.landmarks.L if views.landmarks is not None else 0
    Nmax = N + 3 * F
    out = dict(feat_status=np.zeros(F, dtype=np.int32), lm_cov_id=np.zeros(F, dtype=np.int32), lm_value=np.zeros((F, 3)), lm_fej=np.zeros((F, 3)),
               anchor_cam=np.zeros(F, dtype=np.int32), anchor_clone=np.zeros(F, dtype=np.int32), clone_q_p=np.zeros((Cn, 7)),
               calib_q_p=np.zeros((K, 7)), intrinsics=np.zeros((K, 8)), landmarks_existing=np.zeros((L0, 3)))
    Pbuf = np.zeros(Nmax * Nmax)
    N_out = C.c_int32(0)
    is_aruco, sa, ma = _aruco(F, feat_sigma, feat_chi2mult, opts)
    if feat_is_aruco is not None:  # the ArUco corners named explicitly (they may share the SLAM options and differ in representation only)
        given = np.ascontiguousarray(feat_is_aruco, dtype=np.int32)
        if is_aruco is None or not is_aruco.any():
            sa, ma = float(opts.sigma_pix), float(opts.chi2_multipler)
        else:
            assert np.array_equal(given, is_aruco), "features with their own sigma / multiplier are the ArUco corners"
        is_aruco = given
    rc = load().ref_slam_delayed_init(C.byref(opts), C.byref(views.state), C.byref(views.landmarks) if L0 else None, C.byref(views.features),
                                      C.c_int(int(feat_rep)), _pi(is_aruco), C.c_double(sa), C.c_double(ma), _pi(out["feat_status"]),
                                      _pi(out["lm_cov_id"]), _p(out["lm_value"]), _p(out["lm_fej"]), _pi(out["anchor_cam"]), _pi(out["anchor_clone"]),
                                      C.byref(N_out), _p(Pbuf), _p(out["clone_q_p"]), _p(out["calib_q_p"]), _p(out["intrinsics"]),
                                      _p(out["landmarks_existing"]) if L0 else None, C.c_int(int(feat_rep_aruco)))
    assert rc == 0
    n = N_out.value
    out["N"] = n
    out["P"] = Pbuf[: n * n].reshape(n, n).copy()
    return out


def anchor_change(opts, views, l, new_cam, new_clone):
    N = views.state.N
    P, val, fej = np.zeros((N, N)), np.zeros(3), np.zeros(3)
    rc = load().ref_anchor_change(C.byref(opts), C.byref(views.state), C.byref(views.landmarks), C.c_int(int(l)), C.c_int(int(new_cam)),
                                  C.c_int(int(new_clone)), _p(P), _p(val), _p(fej))
    return dict(rc=rc, P=P, value=val, fej=fej)


def change_anchors(opts, views):
    N, L = views.state.N, views.landmarks.L
    out = dict(P=np.zeros((N, N)), value=np.zeros((L, 3)), fej=np.zeros((L, 3)), anchor_clone=np.zeros(L, dtype=np.int32))
    rc = load().ref_change_anchors(C.byref(opts), C.byref(views.state), C.byref(views.landmarks), _p(out["P"]), _p(out["value"]), _p(out["fej"]),
                                   _pi(out["anchor_clone"]))
    assert rc == 0
    return out


def marginalize_clone(opts, views, clone):
    N = views.state.N
    out = np.zeros((N - 6, N - 6))
    has_lm = getattr(views, "landmarks", None) is not None and views.landmarks.L > 0
    rc = load().ref_marginalize(C.byref(opts), C.byref(views.state), C.byref(views.landmarks) if has_lm else None, C.c_int(int(clone)), C.c_int(-1), _p(out))
    assert rc == 0
    return out


def augment_clone(opts, views, imu_value, last_w):
    N = views.state.N
    P, clone = np.zeros((N + 6, N + 6)), np.zeros(7)
    iv = np.ascontiguousarray(imu_value, dtype=np.float64)
    w = np.ascontiguousarray(last_w, dtype=np.float64)
    rc = load().ref_augment_clone(C.byref(opts), C.byref(views.state), _p(iv), _p(w), _p(P), _p(clone))
    assert rc == 0
    return P, clone


def propagate_imu(opts, views, Phi, Q):
    N = views.state.N
    P = np.zeros((N, N))
    Phi = np.ascontiguousarray(Phi, dtype=np.float64)
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    rc = load().ref_propagate_imu(C.byref(opts), C.byref(views.state), _p(Phi), _p(Q), _p(P))
    assert rc == 0
    return P


def ekf_update(opts, views, H, res, col_cov_id, sigma2):
    N = views.state.N
    H = np.ascontiguousarray(H, dtype=np.float64)
    res = np.ascontiguousarray(res, dtype=np.float64)
    cols = np.ascontiguousarray(col_cov_id, dtype=np.int32)
    dx, P = np.zeros(N), np.zeros((N, N))
    rc = load().ref_ekf_update(C.byref(opts), C.byref(views.state), _p(H), _p(res), C.c_int(H.shape[0]), C.c_int(H.shape[1]), _pi(cols),
                               C.c_double(float(sigma2)), _p(dx), _p(P))
    return rc, P, dx


def zupt_try_update(opts, views, imu_value, imu_t, imu_wm, imu_am, t_state, t_update, max_velocity=0.5, noise_multiplier=10.0, max_disparity=1.0):
    """UpdaterZeroVelocity::try_update on the view's state with the IMU at imu_value [16] (oracle/ref/ref_driver.cpp: ref_zupt_try_update)."""
    N = views.state.N
    iv = np.ascontiguousarray(imu_value, dtype=np.float64)
    t = np.ascontiguousarray(imu_t, dtype=np.float64)
    wm = np.ascontiguousarray(imu_wm, dtype=np.float64)
    am = np.ascontiguousarray(imu_am, dtype=np.float64)
    out = dict(dx=np.zeros(N), P=np.zeros((N, N)), imu=np.zeros(16))
    acc, ts = C.c_int32(0), C.c_double(0.0)
    lib = load()
    lib.ref_zupt_try_update.restype = C.c_int
    rc = lib.ref_zupt_try_update(C.byref(opts), C.byref(views.state), _p(iv), C.c_int(len(t)), _p(t), _p(wm), _p(am), C.c_double(t_state), C.c_double(t_update),
                                 C.c_double(max_velocity), C.c_double(noise_multiplier), C.c_double(max_disparity), C.byref(acc), _p(out["dx"]), _p(out["P"]),
                                 _p(out["imu"]), C.byref(ts))
    assert rc == 0
    out["accepted"], out["timestamp"] = bool(acc.value), ts.value
    return out
