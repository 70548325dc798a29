"""The rpng_sim closed loop out of the reference's own classes (oracle/ref/ref_sim.cpp inside oracle/_ref/libov_ref.so): Simulator,
Propagator, FeatureDatabase, State / StateHelper and UpdaterMSCKF are the reference's; the loop stops in front of every MSCKF update
and hands it out as a synth.Problem-like snapshot, so that the SAME filter can be driven with the reference's update, the oracle's
or the HIP library's.  TEST INFRASTRUCTURE ONLY (see oracle/pyref.py).
"""
from __future__ import annotations

import ctypes as C
import os
from types import SimpleNamespace

import numpy as np

from oracle import pyref

TRAJ_REFERENCE = "/root/reference/ov_data/sim/tum_corridor1_512_16_okvis.txt"
TRAJ_FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "sim_traj_corridor_90s.txt")


class Config(C.Structure):
    _fields_ = [("traj_path", C.c_char_p), ("num_cameras", C.c_int32), ("max_clones", C.c_int32), ("max_msckf_in_update", C.c_int32),
                ("num_pts", C.c_int32), ("use_fej", C.c_int32), ("integration", C.c_int32), ("calib_cam_extrinsics", C.c_int32),
                ("calib_cam_intrinsics", C.c_int32), ("calib_cam_timeoffset", C.c_int32), ("calib_imu_intrinsics", C.c_int32),
                ("calib_imu_g_sensitivity", C.c_int32), ("feat_rep_msckf", C.c_int32), ("use_stereo", C.c_int32), ("do_perturbation", C.c_int32),
                ("seed_state_init", C.c_int32), ("seed_perturb", C.c_int32), ("seed_measurements", C.c_int32), ("sigma_px", C.c_double),
                ("chi2_multipler", C.c_double), ("freq_cam", C.c_double), ("freq_imu", C.c_double), ("distance_threshold", C.c_double),
                ("min_feature_gen_dist", C.c_double), ("max_feature_gen_dist", C.c_double),
                ("max_slam_features", C.c_int32), ("feat_rep_slam", C.c_int32), ("dt_slam_delay", C.c_double)]


def rpng_sim_config(traj_path=None, **kw):
    """config/rpng_sim/estimator_config.yaml as BASELINE configs[0] reads it: one camera, 11 clones, max_msckf_in_update
    raised from 10 to 50, no SLAM features, every calibration flag on (N = 126 with 12 clones), rk4 integration, FEJ, seeds 0."""
    traj = traj_path or TRAJ_FIXTURE  # the committed 95 s excerpt (tools/make_traj_fixture.py): the same stream here and on the GPU box
    c = Config(traj_path=traj.encode(), num_cameras=1, max_clones=11, max_msckf_in_update=50, num_pts=250, use_fej=1, integration=1,
               calib_cam_extrinsics=1, calib_cam_intrinsics=1, calib_cam_timeoffset=1, calib_imu_intrinsics=1, calib_imu_g_sensitivity=1,
               feat_rep_msckf=0, use_stereo=1, do_perturbation=0, seed_state_init=0, seed_perturb=0, seed_measurements=0, sigma_px=1.0,
               chi2_multipler=1.0, freq_cam=10.0, freq_imu=400.0, distance_threshold=1.1, min_feature_gen_dist=5.0, max_feature_gen_dist=7.0,
               max_slam_features=0, feat_rep_slam=0, dt_slam_delay=2.0)  # (the shipped file has max_slam: 50; BASELINE configs[0] is the MSCKF-only filter)
    for k, v in kw.items():
        if not hasattr(c, k):
            raise AttributeError(k)
        setattr(c, k, v)
    c._keep = traj.encode()
    return c


class RefSim:
    def __init__(self, cfg: Config):
        self.lib = pyref.load()
        self.lib.ref_sim_create.restype = C.c_void_p
        self.lib.ref_sim_create.argtypes = [C.POINTER(Config)]
        self.cfg = cfg
        self.h = self.lib.ref_sim_create(C.byref(cfg))
        assert self.h, "the simulator could not start (trajectory file?)"
        self.h = C.c_void_p(self.h)

    def close(self):
        if self.h:
            self.lib.ref_sim_destroy(self.h)
            self.h = None

    def perturb(self, eps):
        """Moves the initial position estimate by eps m along x (see ref_sim_perturb: the control run of the closed-loop tests)."""
        self.lib.ref_sim_perturb(self.h, C.c_double(eps))

    def advance(self):
        """Runs the simulation until an MSCKF update is ready; False when the trajectory ended."""
        return self.lib.ref_sim_advance(self.h) == 1

    def pending(self, with_cov=True):
        """The waiting update as a synth.Problem-like snapshot (+ featid).  with_cov = False: P stays zero (the filter's own updaters
        will do the update: nothing reads it, and a covariance resident on a device is not brought back for it)."""
        N, Cn, K, F, M = (C.c_int32() for _ in range(5))
        self.lib.ref_sim_dims(self.h, C.byref(N), C.byref(Cn), C.byref(K), C.byref(F), C.byref(M))
        N, Cn, K, F, M = N.value, Cn.value, K.value, F.value, M.value
        p = SimpleNamespace(N=N, C=Cn, K=K, P=np.zeros((N, N)), clone_q_p=np.zeros((Cn, 7)), clone_q_p_fej=np.zeros((Cn, 7)),
                            clone_cov_id=np.zeros(Cn, np.int32), calib_q_p=np.zeros((K, 7)), intrinsics=np.zeros((K, 8)),
                            cam_is_fisheye=np.zeros(K, np.uint8), calib_cov_id=np.zeros(K, np.int32), intr_cov_id=np.zeros(K, np.int32),
                            meas_offsets=np.zeros(F + 1, np.int32), uv=np.zeros(2 * max(M, 1), np.float32), uvn=np.zeros(2 * max(M, 1), np.float32),
                            clone_idx=np.zeros(max(M, 1), np.int32), cam_idx=np.zeros(max(M, 1), np.int32), featid=np.zeros(F, np.uint64),
                            lm_value=None)
        dp, ip, fp = C.POINTER(C.c_double), C.POINTER(C.c_int32), C.POINTER(C.c_float)
        a = lambda x, t: x.ctypes.data_as(t)
        self.lib.ref_sim_export(self.h, a(p.P, dp) if with_cov else None, a(p.clone_q_p, dp), a(p.clone_q_p_fej, dp), a(p.clone_cov_id, ip), a(p.calib_q_p, dp),
                                a(p.intrinsics, dp), a(p.calib_cov_id, ip), a(p.intr_cov_id, ip), a(p.meas_offsets, ip), a(p.uv, fp), a(p.uvn, fp),
                                a(p.clone_idx, ip), a(p.cam_idx, ip), p.featid.ctypes.data_as(C.POINTER(C.c_uint64)))
        p.uv, p.uvn, p.clone_idx, p.cam_idx = p.uv[: 2 * M], p.uvn[: 2 * M], p.clone_idx[:M], p.cam_idx[:M]
        p.F, p.M = F, M
        return p

    def update_reference(self, F):
        used = np.zeros(max(F, 1), np.int32)
        self.lib.ref_sim_update_reference(self.h, used.ctypes.data_as(C.POINTER(C.c_int32)))
        return used[:F]

    def update_external(self, dx, P, feat_status, p_FinG=None):
        dx = np.ascontiguousarray(dx, np.float64)
        P = np.ascontiguousarray(P, np.float64)
        st = np.ascontiguousarray(feat_status, np.int32)
        pg = np.ascontiguousarray(p_FinG, np.float64) if p_FinG is not None else None
        dp = C.POINTER(C.c_double)
        self.lib.ref_sim_update_external(self.h, dx.ctypes.data_as(dp), P.ctypes.data_as(dp), st.ctypes.data_as(C.POINTER(C.c_int32)),
                                         pg.ctypes.data_as(dp) if pg is not None else None)

    def times(self):
        """Wall seconds inside propagate_and_clone / UpdaterMSCKF::update / marginalize_old_clone so far, features and observations handed to the updater."""
        out = np.zeros(5)
        self.lib.ref_sim_times(self.h, out.ctypes.data_as(C.POINTER(C.c_double)))
        return dict(propagate_s=out[0], update_s=out[1], marginalize_s=out[2], features=int(out[3]), observations=int(out[4]))

    def finish(self):
        self.lib.ref_sim_finish(self.h)

    def state(self):
        est, gt, extra = np.zeros(17), np.zeros(17), np.zeros(4)
        dp = C.POINTER(C.c_double)
        ok = self.lib.ref_sim_state(self.h, est.ctypes.data_as(dp), gt.ctypes.data_as(dp), extra.ctypes.data_as(dp))
        return est, gt, extra, ok == 1
