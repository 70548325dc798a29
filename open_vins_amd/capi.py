"""ctypes mirror of include/ovgpu.h and loader of the in-tree HIP library.

The product path has no CPU fallback: ``load()`` raises when libovgpu.so is
missing, and ``ovgpu_create`` fails with OVGPU_ERR_NO_DEVICE when no GPU is
present.  Nothing here imports or touches oracle/.
"""
from __future__ import annotations

import ctypes as C
import os
import sys

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "csrc", "libovgpu.so")

OK = 0
ERR_INVALID, ERR_NO_DEVICE, ERR_HIP, ERR_NEGATIVE_DIAGONAL, ERR_NOT_SPD, ERR_CAPACITY, ERR_NO_STATE = 1, 2, 3, 4, 5, 6, 7
FEAT_USED, FEAT_TOO_FEW_MEAS, FEAT_TRI_FAILED, FEAT_GN_FAILED, FEAT_CHI2_REJECTED = 0, 1, 2, 3, 4
REP_GLOBAL_3D, REP_GLOBAL_FULL_INVERSE_DEPTH, REP_ANCHORED_3D = 0, 1, 2
REP_ANCHORED_FULL_INVERSE_DEPTH, REP_ANCHORED_MSCKF_INVERSE_DEPTH, REP_ANCHORED_INVERSE_DEPTH_SINGLE = 3, 4, 5
COMPRESS_GRAM, COMPRESS_TSQR, COMPRESS_PCHOLQR = 0, 1, 3  # (2: the unpivoted Gram factor, retired with ABI 8)
GROUPS_REFERENCE, GROUPS_DESCENDING, GROUPS_ASCENDING = 0, 1, 2

c_double_p = C.POINTER(C.c_double)
c_float_p = C.POINTER(C.c_float)
c_int64_p = C.POINTER(C.c_int64)
c_int32_p = C.POINTER(C.c_int32)
c_uint8_p = C.POINTER(C.c_uint8)


class Options(C.Structure):
    """ovgpu_options"""
    _fields_ = [
        ("chi2_multipler", C.c_double), ("sigma_pix", C.c_double),
        ("triangulate_1d", C.c_int32), ("refine_features", C.c_int32), ("max_runs", C.c_int32), ("_pad0", C.c_int32),
        ("init_lamda", C.c_double), ("max_lamda", C.c_double), ("min_dx", C.c_double), ("min_dcost", C.c_double),
        ("lam_mult", C.c_double), ("min_dist", C.c_double), ("max_dist", C.c_double), ("max_baseline", C.c_double),
        ("max_cond_number", C.c_double),
        ("do_fej", C.c_int32), ("do_calib_camera_pose", C.c_int32), ("do_calib_camera_intrinsics", C.c_int32),
        ("feat_rep_msckf", C.c_int32),
        ("compress_route", C.c_int32), ("gram_no_whiten", C.c_int32), ("no_prior_overlap", C.c_int32), ("tsqr_workers", C.c_int32),
        ("tsqr_no_pipeline", C.c_int32), ("tsqr_overlap", C.c_int32), ("gate_always_factor", C.c_int32), ("no_timing", C.c_int32),
        ("no_fast_feature_kernel", C.c_int32), ("no_single_launch_cholesky", C.c_int32), ("prior_pivot_tol", C.c_double),
        ("feature_kernel_shape", C.c_int32), ("gram_fp32", C.c_int32),
    ]


def default_options(**kw) -> Options:
    """Reference defaults (UpdaterOptions.h:32-48, FeatureInitializerOptions.h:33-69) with the rpng_sim
    estimator settings for the StateOptions subset (config/rpng_sim/estimator_config.yaml:5,10-11,24)."""
    o = Options()
    o.chi2_multipler, o.sigma_pix = 5.0, 1.0
    o.triangulate_1d, o.refine_features, o.max_runs = 0, 1, 5
    o.init_lamda, o.max_lamda, o.min_dx, o.min_dcost, o.lam_mult = 1e-3, 1e10, 1e-6, 1e-6, 10.0
    o.min_dist, o.max_dist, o.max_baseline, o.max_cond_number = 0.10, 60.0, 40.0, 10000.0
    o.do_fej, o.do_calib_camera_pose, o.do_calib_camera_intrinsics, o.feat_rep_msckf = 1, 1, 1, REP_GLOBAL_3D
    for k, v in kw.items():
        if not hasattr(o, k):
            raise AttributeError(k)
        setattr(o, k, v)
    return o


class StateView(C.Structure):
    """ovgpu_state_view"""
    _fields_ = [
        ("N", C.c_int32), ("C", C.c_int32), ("K", C.c_int32), ("_pad0", C.c_int32),
        ("P", c_double_p), ("clone_q_p", c_double_p), ("clone_q_p_fej", c_double_p), ("clone_cov_id", c_int32_p),
        ("calib_q_p", c_double_p), ("intrinsics", c_double_p), ("cam_is_fisheye", c_uint8_p),
        ("calib_cov_id", c_int32_p), ("intr_cov_id", c_int32_p),
    ]


class FeaturesView(C.Structure):
    """ovgpu_features_view"""
    _fields_ = [
        ("F", C.c_int32), ("M", C.c_int32), ("meas_offsets", c_int32_p), ("uv", c_float_p), ("uvn", c_float_p),
        ("clone_idx", c_int32_p), ("cam_idx", c_int32_p),
    ]


class LandmarksView(C.Structure):
    """ovgpu_landmarks_view"""
    _fields_ = [("L", C.c_int32), ("feat_rep", C.c_int32), ("p_value", c_double_p), ("p_fej", c_double_p), ("cov_id", c_int32_p),
                ("anchor_cam", c_int32_p), ("anchor_clone", c_int32_p), ("feat_rep_each", c_int32_p)]


class UpdateStats(C.Structure):
    """ovgpu_update_stats"""
    _fields_ = [
        ("n_used", C.c_int32), ("n_rows", C.c_int32), ("D", C.c_int32), ("n_rows_comp", C.c_int32),
        ("status", C.c_int32), ("n_gate_bound", C.c_int32),
        ("ms_triangulate", C.c_float), ("ms_system", C.c_float), ("ms_compress", C.c_float), ("ms_update", C.c_float),
        ("ms_total", C.c_float), ("_pad1", C.c_float),
    ]

    def as_dict(self):
        return {k: getattr(self, k) for k, _ in self._fields_ if not k.startswith("_")}


def _ptr(a, ctype):
    if a is None:
        return None
    return a.ctypes.data_as(C.POINTER(ctype))


class Views:
    """Keeps the numpy arrays alive next to the ctypes views built from a synth.Problem-like object."""

    def __init__(self, prob):
        f64 = lambda a: np.ascontiguousarray(a, dtype=np.float64)
        i32 = lambda a: np.ascontiguousarray(a, dtype=np.int32)
        self.P = f64(prob.P)
        self.clone_q_p = f64(prob.clone_q_p)
        self.clone_q_p_fej = f64(prob.clone_q_p_fej)
        self.clone_cov_id = i32(prob.clone_cov_id)
        self.calib_q_p = f64(prob.calib_q_p)
        self.intrinsics = f64(prob.intrinsics)
        self.cam_is_fisheye = np.ascontiguousarray(prob.cam_is_fisheye, dtype=np.uint8)
        self.calib_cov_id = i32(prob.calib_cov_id)
        self.intr_cov_id = i32(prob.intr_cov_id)
        self.meas_offsets = i32(prob.meas_offsets)
        self.uv = np.ascontiguousarray(prob.uv, dtype=np.float32)
        self.uvn = np.ascontiguousarray(prob.uvn, dtype=np.float32)
        self.clone_idx = i32(prob.clone_idx)
        self.cam_idx = i32(prob.cam_idx)
        st = StateView()
        st.N, st.C, st.K = int(prob.N), int(prob.C), int(prob.K)
        st.P = _ptr(self.P, C.c_double)
        st.clone_q_p = _ptr(self.clone_q_p, C.c_double)
        st.clone_q_p_fej = _ptr(self.clone_q_p_fej, C.c_double)
        st.clone_cov_id = _ptr(self.clone_cov_id, C.c_int32)
        st.calib_q_p = _ptr(self.calib_q_p, C.c_double)
        st.intrinsics = _ptr(self.intrinsics, C.c_double)
        st.cam_is_fisheye = _ptr(self.cam_is_fisheye, C.c_uint8)
        st.calib_cov_id = _ptr(self.calib_cov_id, C.c_int32)
        st.intr_cov_id = _ptr(self.intr_cov_id, C.c_int32)
        self.state = st
        fv = FeaturesView()
        fv.F = len(self.meas_offsets) - 1
        fv.M = int(self.meas_offsets[-1])
        fv.meas_offsets = _ptr(self.meas_offsets, C.c_int32)
        fv.uv = _ptr(self.uv, C.c_float)
        fv.uvn = _ptr(self.uvn, C.c_float)
        fv.clone_idx = _ptr(self.clone_idx, C.c_int32)
        fv.cam_idx = _ptr(self.cam_idx, C.c_int32)
        self.features = fv
        # SLAM landmarks (synth.make_slam_problem); absent for MSCKF snapshots
        self.landmarks = None
        if getattr(prob, "lm_value", None) is not None:
            self.lm_value = f64(prob.lm_value)
            self.lm_fej = f64(prob.lm_fej)
            self.lm_cov_id = i32(prob.lm_cov_id)
            self.lm_index = i32(prob.lm_index)
            lv = LandmarksView()
            lv.L = int(self.lm_cov_id.shape[0])
            lv.p_value = _ptr(self.lm_value, C.c_double)
            lv.p_fej = _ptr(self.lm_fej, C.c_double)
            lv.cov_id = _ptr(self.lm_cov_id, C.c_int32)
            lv.feat_rep = int(getattr(prob, "lm_rep", 0) or 0)
            # one representation per landmark (ABI 7; UpdaterSLAM.cpp:336-341), or lm_rep for all
            self.lm_rep_each = i32(prob.lm_rep_each) if getattr(prob, "lm_rep_each", None) is not None else None
            lv.feat_rep_each = _ptr(self.lm_rep_each, C.c_int32)
            self.lm_anchor_cam = i32(prob.lm_anchor_cam) if getattr(prob, "lm_anchor_cam", None) is not None else None
            self.lm_anchor_clone = i32(prob.lm_anchor_clone) if getattr(prob, "lm_anchor_clone", None) is not None else None
            lv.anchor_cam = _ptr(self.lm_anchor_cam, C.c_int32)
            lv.anchor_clone = _ptr(self.lm_anchor_clone, C.c_int32)
            self.landmarks = lv


_lib = None


def declare(lib):
    """Attaches argtypes/restypes for every symbol include/ovgpu.h declares."""
    vp = C.c_void_p
    ctxp = C.c_void_p
    S = {
        "ovgpu_default_options": (None, [C.POINTER(Options)]),
        "ovgpu_create": (C.c_int, [C.POINTER(Options), C.c_int, C.POINTER(ctxp)]),
        "ovgpu_destroy": (None, [ctxp]),
        "ovgpu_last_error": (C.c_char_p, []),
        "ovgpu_chi2_quantile_95": (C.c_double, [C.c_int]),
        "ovgpu_abi_version": (C.c_int, []),
        "ovgpu_set_state": (C.c_int, [ctxp, C.POINTER(StateView)]),
        "ovgpu_set_camera_poses": (C.c_int, [ctxp, C.c_int, C.c_int, c_double_p, c_double_p]),
        "ovgpu_set_features": (C.c_int, [ctxp, C.POINTER(FeaturesView)]),
        "ovgpu_triangulate": (C.c_int, [ctxp, c_double_p, c_double_p, c_int32_p, c_int32_p]),
        "ovgpu_state_marginal_covariance": (C.c_int, [ctxp, C.c_int32, c_int32_p, c_double_p]),
        "ovgpu_retriangulate": (C.c_int, [ctxp, C.c_int32, C.c_int32, c_int64_p, c_int32_p, c_float_p, c_float_p, C.c_int32, C.c_int32, C.c_int32, c_int32_p,
                                          c_int64_p, c_double_p, c_double_p]),
        "ovgpu_retriangulate_reset": (C.c_int, [ctxp]),
        "ovgpu_refine": (C.c_int, [ctxp, c_double_p, c_int32_p, c_double_p, c_double_p, c_int32_p]),
        "ovgpu_get_triangulation": (C.c_int, [ctxp, c_double_p, c_double_p, c_int32_p]),
        "ovgpu_set_triangulation": (C.c_int, [ctxp, c_double_p, c_double_p, c_int32_p, c_int32_p]),
        "ovgpu_msckf_update": (C.c_int, [ctxp, c_int32_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                         C.POINTER(UpdateStats)]),
        "ovgpu_msckf_compress": (C.c_int, [ctxp, c_int32_p, c_double_p, c_double_p, c_double_p, c_int32_p, c_int32_p,
                                           c_int32_p, c_double_p, c_double_p, C.POINTER(UpdateStats)]),
        "ovgpu_get_state": (C.c_int, [ctxp, c_double_p, c_double_p, c_double_p, c_double_p]),
        "ovgpu_measurement_compress": (C.c_int, [ctxp, C.c_int, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p, c_int32_p]),
        "ovgpu_ekf_update": (C.c_int, [ctxp, C.c_int, C.c_int, c_int32_p, c_double_p, c_double_p, C.c_double, c_double_p, c_double_p]),
        "ovgpu_set_landmarks": (C.c_int, [ctxp, C.POINTER(LandmarksView)]),
        "ovgpu_tracks_create": (C.c_int, [ctxp, C.c_int32, C.c_int32]),
        "ovgpu_tracks_append": (C.c_int, [ctxp, C.c_double, C.c_int32, C.POINTER(C.c_int64), c_int32_p, c_float_p, c_float_p]),
        "ovgpu_tracks_erase": (C.c_int, [ctxp, C.c_int32, C.POINTER(C.c_int64)]),
        "ovgpu_tracks_not_containing_newer": (C.c_int, [ctxp, C.c_double, C.c_int32, C.POINTER(C.c_int64), c_int32_p]),
        "ovgpu_tracks_containing_older": (C.c_int, [ctxp, C.c_double, C.c_int32, C.POINTER(C.c_int64), c_int32_p]),
        "ovgpu_tracks_containing": (C.c_int, [ctxp, C.c_double, C.c_int32, C.POINTER(C.c_int64), c_int32_p]),
        "ovgpu_tracks_oldest_timestamp": (C.c_int, [ctxp, c_double_p]),
        "ovgpu_tracks_cleanup_measurements": (C.c_int, [ctxp, C.c_double, c_int32_p]),
        "ovgpu_tracks_cleanup_measurements_exact": (C.c_int, [ctxp, C.c_double, c_int32_p]),
        "ovgpu_tracks_get_feature": (C.c_int, [ctxp, C.c_int64, C.c_int32, c_int32_p, c_double_p, c_int32_p, c_float_p, c_float_p]),
        "ovgpu_tracks_count": (C.c_int, [ctxp, c_int32_p]),
        "ovgpu_tracks_to_features": (C.c_int, [ctxp, C.c_int32, C.POINTER(C.c_int64), c_double_p]),
        "ovgpu_tracks_group_order": (C.c_int, [ctxp, C.c_int32]),
        "ovgpu_set_feature_options": (C.c_int, [ctxp, c_double_p, c_double_p]),
        "ovgpu_set_feature_reps": (C.c_int, [ctxp, c_int32_p]),
        "ovgpu_get_landmark_reps": (C.c_int, [ctxp, c_int32_p, c_int32_p]),
        "ovgpu_get_features": (C.c_int, [ctxp, c_int32_p, c_int32_p, c_int32_p, c_float_p, c_float_p, c_int32_p, c_int32_p]),
        "ovgpu_slam_change_anchor": (C.c_int, [ctxp, C.c_int32, C.c_int32, C.c_int32]),
        "ovgpu_slam_change_anchors": (C.c_int, [ctxp, C.c_int32, C.c_int32, c_int32_p]),
        "ovgpu_state_marginalize": (C.c_int, [ctxp, C.c_int32, C.c_int32]),
        "ovgpu_state_augment_clone": (C.c_int, [ctxp, C.c_int32, c_double_p, c_double_p, C.c_int32, c_double_p, c_int32_p]),
        "ovgpu_state_propagate": (C.c_int, [ctxp, C.c_int32, C.c_int32, C.c_int32, c_int32_p, c_double_p, c_double_p]),
        "ovgpu_state_dims": (C.c_int, [ctxp, c_int32_p, c_int32_p]),
        "ovgpu_get_landmarks": (C.c_int, [ctxp, c_int32_p, c_double_p, c_double_p, c_int32_p, c_int32_p, c_int32_p]),
        "ovgpu_slam_delayed_init": (C.c_int, [ctxp, C.c_int32, c_int32_p, c_double_p, c_double_p, c_int32_p, c_double_p, c_double_p, c_int32_p,
                                              c_int32_p, c_double_p, c_int32_p, c_double_p, C.POINTER(UpdateStats)]),
        "ovgpu_slam_compress": (C.c_int, [ctxp, c_int32_p, c_int32_p, c_double_p, c_double_p, c_int32_p, c_int32_p, c_int32_p, c_double_p,
                                          c_double_p, C.POINTER(UpdateStats)]),
        "ovgpu_slam_update": (C.c_int, [ctxp, c_int32_p, c_int32_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p,
                                        C.POINTER(UpdateStats)]),
        "ovgpu_triangle_len": (C.c_int, [ctxp, C.POINTER(C.c_int64)]),
        "ovgpu_msckf_local": (C.c_int, [ctxp, c_int32_p, c_double_p, c_double_p, c_double_p, vp, C.POINTER(UpdateStats)]),
        "ovgpu_msckf_merge_update": (C.c_int, [ctxp, vp, C.c_int, c_double_p, c_double_p, C.POINTER(UpdateStats)]),
        "ovgpu_gram_len": (C.c_int, [ctxp, C.POINTER(C.c_int64)]),
        "ovgpu_msckf_local_gram": (C.c_int, [ctxp, c_int32_p, c_double_p, c_double_p, c_double_p, vp, C.POINTER(UpdateStats)]),
        "ovgpu_msckf_gram_update": (C.c_int, [ctxp, vp, c_double_p, c_double_p, C.POINTER(UpdateStats)]),
        "ovgpu_cam_distort": (C.c_int, [ctxp, C.c_int, c_double_p, C.c_int, c_double_p, c_double_p, c_double_p, c_double_p]),
        "ovgpu_reset_state": (C.c_int, [ctxp]),
        "ovgpu_msckf_update_async": (C.c_int, [ctxp]),
        "ovgpu_synchronize": (C.c_int, [ctxp]),
        "ovgpu_stream": (C.c_uint64, [ctxp]),
        "ovgpu_last_update_route": (C.c_int, [ctxp]),
        "ovgpu_comm_unique_id": (C.c_int, [vp]),
        "ovgpu_comm_init_rank": (C.c_int, [ctxp, vp, C.c_int, C.c_int]),
        "ovgpu_comm_destroy": (C.c_int, [ctxp]),
        "ovgpu_comm_info": (C.c_int, [ctxp, C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32), C.POINTER(C.c_int32)]),
        "ovgpu_msckf_update_sharded": (C.c_int, [ctxp, c_int32_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.POINTER(UpdateStats)]),
        "ovgpu_msckf_update_sharded_async": (C.c_int, [ctxp]),
        "ovgpu_multi_create": (C.c_int, [C.POINTER(Options), C.c_int, c_int32_p, C.POINTER(vp)]),
        "ovgpu_multi_destroy": (None, [vp]),
        "ovgpu_multi_size": (C.c_int, [vp]),
        "ovgpu_multi_ctx": (vp, [vp, C.c_int]),
        "ovgpu_multi_set_state": (C.c_int, [vp, C.POINTER(StateView)]),
        "ovgpu_multi_set_features": (C.c_int, [vp, C.POINTER(FeaturesView)]),
        "ovgpu_multi_msckf_update": (C.c_int, [vp, c_int32_p, c_double_p, c_double_p, c_double_p, c_double_p, c_double_p, C.POINTER(UpdateStats)]),
        "ovgpu_debug_cycles": (C.c_int, [ctxp, C.c_int, C.POINTER(C.c_longlong)]),
        "ovgpu_debug_box_probe": (C.c_int, [ctxp, c_double_p]),
        "ovgpu_debug_option": (C.c_int, [ctxp, C.c_char_p, C.c_int64, C.POINTER(C.c_int64)]),
        "ovgpu_kernel_times": (C.c_int, [ctxp, C.c_int, c_double_p, c_double_p, C.POINTER(C.c_int64)]),
        "ovgpu_system_time": (C.c_int, [ctxp, c_double_p, C.POINTER(C.c_int64)]),
    }
    for name, (res, args) in S.items():
        fn = getattr(lib, name)  # AttributeError if the library does not export it
        fn.restype = res
        fn.argtypes = args
    return sorted(S)


def load():
    """Loads the in-tree libovgpu.so (built by __graft_entry__.build() / make -C open_vins_amd/csrc)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} is missing: build the HIP extension first (python -c 'import __graft_entry__ as g; g.build()'). "
                "There is no CPU fallback.")
        # torch first when it is installed: its wheel bundles its own HIP runtime, and a process that loads libovgpu.so (linked
        # against /opt/rocm's) BEFORE importing torch ends up with two runtimes of which torch's sees no device (measured on the
        # GPU box: torch.cuda.is_available() turns False).  The sharded update and the benchmark need both in one process.
        if "torch" not in sys.modules and not os.environ.get("OVGPU_NO_TORCH_PRELOAD"):
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        lib = C.CDLL(LIB_PATH)
        declare(lib)
        _lib = lib
    return _lib


class OvgpuError(RuntimeError):
    def __init__(self, code, where):
        lib = load()
        msg = lib.ovgpu_last_error()
        super().__init__(f"{where} failed with ovgpu_status {code}: {msg.decode() if msg else ''}")
        self.code = code


def check(code, where):
    if code != OK:
        raise OvgpuError(code, where)
