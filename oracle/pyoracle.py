"""ctypes loader of oracle/libov_oracle.so.

TEST INFRASTRUCTURE ONLY: may be imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg,
never by the product package.  Pinned by the known-answer fixtures of tests/test_known_answer.py (see ov_oracle.h).
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess

import numpy as np

from open_vins_amd import capi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libov_oracle.so")
_lib = None

dp, ip = capi.c_double_p, capi.c_int32_p


def build():
    subprocess.check_call(["make", "-s", "-C", _HERE])


def load():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        build()
    lib = C.CDLL(LIB_PATH)
    O, SV, FV, US = C.POINTER(capi.Options), C.POINTER(capi.StateView), C.POINTER(capi.FeaturesView), C.POINTER(capi.UpdateStats)
    lib.oracle_chi2_quantile_95.restype = C.c_double
    lib.oracle_chi2_quantile_95.argtypes = [C.c_int]
    lib.oracle_cam_distort.restype = None
    lib.oracle_cam_distort.argtypes = [dp, C.c_int, dp, dp, dp, dp]
    lib.oracle_make_givens.restype = None
    lib.oracle_make_givens.argtypes = [C.c_double, C.c_double, dp, dp]
    lib.oracle_nullspace_project.restype = None
    lib.oracle_nullspace_project.argtypes = [dp, dp, dp, C.c_int, C.c_int, C.c_int]
    lib.oracle_measurement_compress.restype = C.c_int
    lib.oracle_measurement_compress.argtypes = [dp, dp, C.c_int, C.c_int]
    lib.oracle_ekf_update.restype = C.c_int
    lib.oracle_ekf_update.argtypes = [dp, C.c_int, dp, dp, C.c_int, C.c_int, ip, C.c_double, dp]
    lib.oracle_apply_dx.restype = None
    lib.oracle_apply_dx.argtypes = [O, SV, dp, dp, dp, dp]
    lib.oracle_triangulate.restype = C.c_int
    lib.oracle_triangulate.argtypes = [O, SV, FV, dp, dp, ip, ip]
    lib.oracle_feature_jacobian.restype = C.c_int
    lib.oracle_feature_jacobian.argtypes = [O, SV, FV, C.c_int, dp, dp, C.c_int, dp, dp, dp, C.POINTER(C.c_int)]
    lib.oracle_column_map.restype = C.c_int
    lib.oracle_column_map.argtypes = [O, SV, ip]
    lib.oracle_msckf_update.restype = C.c_int
    lib.oracle_msckf_update.argtypes = [O, SV, FV, ip, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, ip, US, dp]
    lib.oracle_msckf_update_given.restype = C.c_int
    lib.oracle_msckf_update_given.argtypes = [O, SV, FV, dp, dp, ip, ip, ip, dp, dp, dp, dp, dp, dp, dp, dp, dp, dp, ip, US, dp]
    _lib = lib
    return lib


def _p(a):
    return a.ctypes.data_as(dp)


def _pi(a):
    return a.ctypes.data_as(ip)


def chi2_quantile_95(dof):
    return load().oracle_chi2_quantile_95(int(dof))


def column_map(opts, views):
    lib = load()
    D = lib.oracle_column_map(C.byref(opts), C.byref(views.state), None)
    cols = np.zeros(D, dtype=np.int32)
    lib.oracle_column_map(C.byref(opts), C.byref(views.state), _pi(cols))
    return cols


def triangulate(opts, views):
    lib = load()
    F = views.features.F
    pA = np.zeros((F, 3))
    pG = np.zeros((F, 3))
    anchor = np.zeros(F, dtype=np.int32)
    status = np.zeros(F, dtype=np.int32)
    lib.oracle_triangulate(C.byref(opts), C.byref(views.state), C.byref(views.features), _p(pA), _p(pG), _pi(anchor), _pi(status))
    return dict(p_FinA=pA, p_FinG=pG, anchor_meas=anchor, status=status)


def feature_jacobian(opts, views, f, p_FinG, p_FinA=None, anchor_meas=-1):
    lib = load()
    m = int(views.meas_offsets[f + 1] - views.meas_offsets[f])
    D = lib.oracle_column_map(C.byref(opts), C.byref(views.state), None)
    H_f = np.zeros((2 * m, 3))
    H_x = np.zeros((2 * m, D))
    res = np.zeros(2 * m)
    nf = C.c_int(3)
    pG = np.ascontiguousarray(p_FinG, dtype=np.float64)
    pA = np.ascontiguousarray(p_FinA if p_FinA is not None else np.zeros(3), dtype=np.float64)
    lib.oracle_feature_jacobian(C.byref(opts), C.byref(views.state), C.byref(views.features), int(f), _p(pG), _p(pA),
                                int(anchor_meas), _p(H_f), _p(H_x), _p(res), C.byref(nf))
    n = nf.value
    if n != 3:
        H_f = np.ascontiguousarray(H_f.reshape(-1)[: 2 * m * n].reshape(2 * m, n))
    return H_f, H_x, res


def nullspace_project(H_f, H_x, res):
    lib = load()
    H_f = np.ascontiguousarray(H_f, dtype=np.float64).copy()
    H_x = np.ascontiguousarray(H_x, dtype=np.float64).copy()
    res = np.ascontiguousarray(res, dtype=np.float64).copy()
    rows, nf = H_f.shape
    lib.oracle_nullspace_project(_p(H_f), _p(H_x), _p(res), rows, nf, H_x.shape[1])
    return H_f, H_x[nf:].copy(), res[nf:].copy()


def measurement_compress(H_x, res):
    lib = load()
    H_x = np.ascontiguousarray(H_x, dtype=np.float64).copy()
    res = np.ascontiguousarray(res, dtype=np.float64).copy()
    r = lib.oracle_measurement_compress(_p(H_x), _p(res), H_x.shape[0], H_x.shape[1])
    return H_x[:r].copy(), res[:r].copy()


def ekf_update(P, H, res, col_cov_id, sigma2):
    lib = load()
    P = np.ascontiguousarray(P, dtype=np.float64).copy()
    H = np.ascontiguousarray(H, dtype=np.float64)
    res = np.ascontiguousarray(res, dtype=np.float64)
    cols = np.ascontiguousarray(col_cov_id, dtype=np.int32)
    dx = np.zeros(P.shape[0])
    st = lib.oracle_ekf_update(_p(P), P.shape[0], _p(H), _p(res), H.shape[0], H.shape[1], _pi(cols), float(sigma2), _p(dx))
    return st, P, dx


def apply_dx(opts, views, dx):
    """Box-plus of the clone / calibration tables with dx (oracle_apply_dx: JPLQuat.h:114-125, PoseJPL.h:74-91, Vec.h:55-58)."""
    lib = load()
    Cn, K = views.state.C, views.state.K
    out = dict(clone_q_p=np.zeros((Cn, 7)), calib_q_p=np.zeros((K, 7)), intrinsics=np.zeros((K, 8)))
    dx = np.ascontiguousarray(dx, dtype=np.float64)
    lib.oracle_apply_dx(C.byref(opts), C.byref(views.state), _p(dx), _p(out["clone_q_p"]), _p(out["calib_q_p"]), _p(out["intrinsics"]))
    return out


def msckf_update(opts, views, want_compressed=False, given=None):
    """Runs the complete reference-order update on the CPU; returns a dict of outputs.

    given = dict(p_FinG=..., p_FinA=..., anchor_meas=..., status=...) injects a triangulation (see ov_oracle.h)."""
    lib = load()
    F, N, Cn, K = views.features.F, views.state.N, views.state.C, views.state.K
    D = lib.oracle_column_map(C.byref(opts), C.byref(views.state), None)
    out = dict(
        feat_status=np.zeros(F, dtype=np.int32), chi2=np.zeros(F), chi2_thresh=np.zeros(F), p_FinG=np.zeros((F, 3)),
        dx=np.zeros(N), P=np.zeros((N, N)), clone_q_p=np.zeros((Cn, 7)), calib_q_p=np.zeros((K, 7)), intrinsics=np.zeros((K, 8)),
    )
    Hc = np.zeros((D, D)) if want_compressed else None
    rc = np.zeros(D) if want_compressed else None
    if want_compressed:
        # rows_comp <= D only when the stack has more rows than columns; otherwise rows = ct_meas <= D too
        pass
    rows = C.c_int32(0)
    stats = capi.UpdateStats()
    secs = np.zeros(4)
    g_pA = g_pG = g_an = g_st = None
    if given is not None:
        g_pG = np.ascontiguousarray(given["p_FinG"], dtype=np.float64)
        g_pA = np.ascontiguousarray(given["p_FinA"], dtype=np.float64) if given.get("p_FinA") is not None else None
        g_an = np.ascontiguousarray(given["anchor_meas"], dtype=np.int32) if given.get("anchor_meas") is not None else None
        g_st = np.ascontiguousarray(given["status"], dtype=np.int32) if given.get("status") is not None else None
    rcode = lib.oracle_msckf_update_given(
        C.byref(opts), C.byref(views.state), C.byref(views.features),
        _p(g_pA) if g_pA is not None else None, _p(g_pG) if g_pG is not None else None,
        _pi(g_an) if g_an is not None else None, _pi(g_st) if g_st is not None else None,
        _pi(out["feat_status"]), _p(out["chi2"]), _p(out["chi2_thresh"]),
        _p(out["p_FinG"]), _p(out["dx"]), _p(out["P"]), _p(out["clone_q_p"]), _p(out["calib_q_p"]), _p(out["intrinsics"]),
        _p(Hc) if want_compressed else None, _p(rc) if want_compressed else None, C.byref(rows), C.byref(stats), _p(secs))
    assert rcode == 0
    out["stats"] = stats.as_dict()
    out["stage_seconds"] = dict(zip(["triangulate", "system", "compress", "update"], secs.tolist()))
    out["rows_comp"] = rows.value
    out["D"] = D
    if want_compressed:
        out["H_comp"] = Hc[: rows.value]
        out["r_comp"] = rc[: rows.value]
    return out


def _opt_arr(a, F):
    if a is None:
        return None, None
    a = np.ascontiguousarray(np.broadcast_to(np.asarray(a, dtype=np.float64), (F,)))
    return a, _p(a)


def slam_update(opts, views, want_stack=False, feat_sigma=None, feat_chi2mult=None):
    """UpdaterSLAM::update (oracle_slam_update); views must carry landmarks.  feat_sigma / feat_chi2mult: per-feature
    sigma_pix and chi2 multiplier (the ArUco options of UpdaterSLAM.cpp:392-409)."""
    lib = load()
    F, N, M = views.features.F, views.state.N, views.features.M
    L = views.landmarks.L
    Dmax = 6 * views.state.C + 14 * views.state.K + 3 * L
    out = dict(feat_status=np.zeros(F, dtype=np.int32), chi2=np.zeros(F), chi2_thresh=np.zeros(F), dx=np.zeros(N), P=np.zeros((N, N)),
               landmarks=np.zeros((L, 3)))
    D, rows = C.c_int32(0), C.c_int32(0)
    cols = np.zeros(Dmax, dtype=np.int32)
    H = np.zeros((2 * max(M, 1), Dmax)) if want_stack else None
    r = np.zeros(2 * max(M, 1)) if want_stack else None
    stats = capi.UpdateStats()
    _fs, fs_p = _opt_arr(feat_sigma, F)
    _fm, fm_p = _opt_arr(feat_chi2mult, F)
    lib.oracle_slam_update.restype = C.c_int
    rc = lib.oracle_slam_update(C.byref(opts), C.byref(views.state), C.byref(views.landmarks), C.byref(views.features), _pi(views.lm_index),
                                _pi(out["feat_status"]), _p(out["chi2"]), _p(out["chi2_thresh"]), _p(out["dx"]), _p(out["P"]), _p(out["landmarks"]),
                                C.byref(D), _pi(cols), _p(H) if want_stack else None, _p(r) if want_stack else None, C.byref(rows), C.byref(stats),
                                fs_p, fm_p)
    assert rc == 0
    d, n = D.value, rows.value
    out["D"], out["rows"], out["col_cov_id"] = d, n, cols[:d].copy()
    out["stats"] = stats.as_dict()
    if want_stack:
        out["H"] = np.ascontiguousarray(H.reshape(-1)[: n * d].reshape(n, d))
        out["r"] = r[:n].copy()
    return out


def slam_delayed_init(opts, views, feat_rep=0, tri=None, feat_sigma=None, feat_chi2mult=None, feat_rep_each=None):
    """UpdaterSLAM::delayed_init + StateHelper::initialize (oracle_slam_delayed_init).  `tri` (the dict of
    triangulate()) replaces the triangulation stage.  views may carry landmarks already in the state."""
    lib = load()
    F, N, Cn, K = views.features.F, views.state.N, views.state.C, views.state.K
    reps = np.full(F, int(feat_rep), np.int32) if feat_rep_each is None else np.ascontiguousarray(feat_rep_each, dtype=np.int32)
    Nmax = N + int(np.where(reps == capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE, 1, 3).sum())
    L0 = views.landmarks.L if views.landmarks is not None else 0
    out = dict(feat_status=np.zeros(F, dtype=np.int32), chi2=np.zeros(F), chi2_thresh=np.zeros(F), lm_cov_id=np.zeros(F, dtype=np.int32),
               lm_value=np.zeros((F, 3)), lm_fej=np.zeros((F, 3)), anchor_cam=np.zeros(F, dtype=np.int32), anchor_clone=np.zeros(F, dtype=np.int32),
               dx_seq=np.zeros((F, Nmax)), clone_q_p=np.zeros((Cn, 7)), calib_q_p=np.zeros((K, 7)), intrinsics=np.zeros((K, 8)),
               landmarks_existing=np.zeros((L0, 3)))
    Pbuf = np.zeros(Nmax * Nmax)
    N_out = C.c_int32(0)
    g = {}
    if tri is not None:
        g = dict(pA=np.ascontiguousarray(tri["p_FinA"], dtype=np.float64), pG=np.ascontiguousarray(tri["p_FinG"], dtype=np.float64),
                 an=np.ascontiguousarray(tri["anchor_meas"], dtype=np.int32), st=np.ascontiguousarray(tri["status"], dtype=np.int32))
    _fs, fs_p = _opt_arr(feat_sigma, F)
    _fm, fm_p = _opt_arr(feat_chi2mult, F)
    lib.oracle_slam_delayed_init.restype = C.c_int
    rc = lib.oracle_slam_delayed_init(C.byref(opts), C.byref(views.state), C.byref(views.landmarks) if views.landmarks is not None else None,
                                      C.byref(views.features), C.c_int(int(feat_rep)), _p(g["pA"]) if g else None, _p(g["pG"]) if g else None,
                                      _pi(g["an"]) if g else None, _pi(g["st"]) if g else None, _pi(out["feat_status"]), _p(out["chi2"]),
                                      _p(out["chi2_thresh"]), _pi(out["lm_cov_id"]), _p(out["lm_value"]), _p(out["lm_fej"]), _pi(out["anchor_cam"]),
                                      _pi(out["anchor_clone"]), _p(out["dx_seq"]), C.byref(N_out), _p(Pbuf), _p(out["clone_q_p"]),
                                      _p(out["calib_q_p"]), _p(out["intrinsics"]), _p(out["landmarks_existing"]) if L0 else None, fs_p, fm_p,
                                      _pi(reps) if feat_rep_each is not None else None)
    out["rc"] = rc
    n = N_out.value
    out["N"] = n
    out["P"] = Pbuf[: n * n].reshape(n, n).copy()
    return out


def marginalize(P, marg_id, marg_size):
    lib = load()
    P = np.ascontiguousarray(P, dtype=np.float64)
    N = P.shape[0]
    out = np.zeros((N - marg_size, N - marg_size))
    lib.oracle_marginalize.restype = None
    lib.oracle_marginalize(_p(P), C.c_int(N), C.c_int(int(marg_id)), C.c_int(int(marg_size)), _p(out))
    return out


def augment_clone(P, old_loc, size=6, dt_id=-1, dnc_dt=None):
    lib = load()
    P = np.ascontiguousarray(P, dtype=np.float64)
    N = P.shape[0]
    out = np.zeros((N + size, N + size))
    d = np.ascontiguousarray(dnc_dt if dnc_dt is not None else np.zeros(size), dtype=np.float64)
    lib.oracle_augment_clone.restype = None
    lib.oracle_augment_clone(_p(P), C.c_int(N), C.c_int(int(old_loc)), C.c_int(int(size)), C.c_int(int(dt_id)), _p(d), _p(out))
    return out


def propagate(P, start_id, old_ids, Phi, Q):
    lib = load()
    P = np.ascontiguousarray(P, dtype=np.float64).copy()
    Phi = np.ascontiguousarray(Phi, dtype=np.float64)
    Q = np.ascontiguousarray(Q, dtype=np.float64)
    ids = np.ascontiguousarray(old_ids, dtype=np.int32)
    lib.oracle_propagate.restype = C.c_int
    rc = lib.oracle_propagate(_p(P), C.c_int(P.shape[0]), C.c_int(int(start_id)), C.c_int(Phi.shape[0]), C.c_int(Phi.shape[1]), _pi(ids), _p(Phi), _p(Q))
    return rc, P


def anchor_change(opts, views, l, new_cam, new_clone):
    """UpdaterSLAM::perform_anchor_change for landmark l of views.landmarks (oracle_anchor_change)."""
    lib = load()
    N = views.state.N
    P = np.zeros((N, N))
    val, fej = np.zeros(3), np.zeros(3)
    lib.oracle_anchor_change.restype = C.c_int
    rc = lib.oracle_anchor_change(C.byref(opts), C.byref(views.state), C.byref(views.landmarks), C.c_int(int(l)), C.c_int(int(new_cam)),
                                  C.c_int(int(new_clone)), _p(P), _p(val), _p(fej))
    return dict(rc=rc, P=P, value=val, fej=fej)
