"""Golden fixtures computed by the REFERENCE'S OWN CODE (oracle/_ref, see oracle/ref/): inputs + outputs of its update entry points on
small seeded snapshots, written to tests/golden/ref_*.npz.  tests/test_ref_fixtures.py holds the oracle (CPU) and the HIP library
(-m gpu) to these files; they need neither /root/reference nor oracle/_ref at test time.

Run here (the container that has /root/reference):   python tools/make_ref_fixtures.py [name-substring ...]   (no argument: every fixture)
Inputs are stored in full (Problem fields as `in_*`, option fields as `opt_*`), so a change of synth.py cannot silently shift them.
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from open_vins_amd import capi, synth  # noqa: E402
from oracle import pyref  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
PROB_FIELDS = ["N", "C", "K", "P", "clone_q_p", "clone_q_p_fej", "clone_cov_id", "calib_q_p", "intrinsics", "cam_is_fisheye", "calib_cov_id",
               "intr_cov_id", "meas_offsets", "uv", "uvn", "clone_idx", "cam_idx", "lm_value", "lm_fej", "lm_cov_id", "lm_index", "lm_rep",
               "lm_anchor_cam", "lm_anchor_clone", "lm_rep_each"]
OPT_FIELDS = ["chi2_multipler", "sigma_pix", "triangulate_1d", "refine_features", "max_runs", "init_lamda", "max_lamda", "min_dx", "min_dcost",
              "lam_mult", "min_dist", "max_dist", "max_baseline", "max_cond_number", "do_fej", "do_calib_camera_pose",
              "do_calib_camera_intrinsics", "feat_rep_msckf"]


def pack(kind, prob, opts, out, **extra):
    d = dict(kind=np.array(kind))
    for k in PROB_FIELDS:
        v = getattr(prob, k, None)
        if v is not None:
            d["in_" + k] = np.asarray(v)
    for k in OPT_FIELDS:
        d["opt_" + k] = np.asarray(getattr(opts, k))
    for k, v in out.items():
        if isinstance(v, (np.ndarray, int, float, np.integer, np.floating)):
            d["out_" + k] = np.asarray(v)
    for k, v in extra.items():
        d["x_" + k] = np.asarray(v)
    return d


def save(name, d):
    if len(sys.argv) > 1 and not any(s in name for s in sys.argv[1:]):
        return
    path = os.path.join(GOLDEN, f"ref_{name}.npz")
    np.savez_compressed(path, **d)
    print(f"{name:34s} {os.path.getsize(path) / 1024:8.1f} KiB  kind={d['kind']}")


def msckf(name, prob, compact=False, **okw):
    opts = capi.default_options(**okw)
    out = pyref.msckf_update(opts, capi.Views(prob))
    assert (out["feat_status"] == capi.FEAT_USED).sum() >= 1
    if compact:  # a large state: keep P' as its diagonal and its action on four fixed vectors
        W = np.random.default_rng(99).normal(size=(prob.N, 4))
        out = dict(out)
        P = out.pop("P")
        out["P_diag"], out["P_W"] = np.diag(P).copy(), P @ W
        save(name, pack("msckf_compact", prob, opts, out, W=W))
    else:
        save(name, pack("msckf", prob, opts, out))


def aruco(F, seed):
    tag = np.random.default_rng(seed).random(F) < 0.4
    return np.where(tag, 2.5, 1.0), np.where(tag, 3.0, 1.0)


def main():
    os.makedirs(GOLDEN, exist_ok=True)
    R = capi
    # ---- UpdaterMSCKF::update -------------------------------------------------------------------------------------------
    msckf("msckf_global3d_fej", synth.make_problem(2, F=14, C=12, K=2, track="ragged", outlier_frac=0.2, seed=11), chi2_multipler=1.0)
    msckf("msckf_anchored_invdepth_fej", synth.make_problem(2, F=12, C=12, K=2, seed=12), chi2_multipler=1.0, do_fej=1,
          feat_rep_msckf=R.REP_ANCHORED_FULL_INVERSE_DEPTH)
    msckf("msckf_anchored_msckf_nofej_equi", synth.make_problem(2, F=12, C=12, K=2, fisheye=True, seed=213), chi2_multipler=1.0, do_fej=0,
          feat_rep_msckf=R.REP_ANCHORED_MSCKF_INVERSE_DEPTH)
    msckf("msckf_single_depth_maps_to_msckf", synth.make_problem(2, F=10, C=12, K=1, seed=14), chi2_multipler=5.0,
          feat_rep_msckf=R.REP_ANCHORED_INVERSE_DEPTH_SINGLE)
    msckf("msckf_rows_le_cols", synth.make_problem(2, F=1, C=12, K=2, track="ragged", min_obs=3, seed=6), chi2_multipler=1.0)   # Q9
    msckf("msckf_no_calibration", synth.make_problem(2, F=12, C=12, K=2, seed=15), chi2_multipler=1.0, do_calib_camera_pose=0,
          do_calib_camera_intrinsics=0)
    msckf("msckf_imu_intrinsics_state", synth.make_problem(2, F=16, C=12, K=2, seed=16, imu_intrinsics=True), chi2_multipler=1.0)
    msckf("msckf_1d_triangulation", synth.make_problem(2, F=12, C=12, K=2, seed=17), chi2_multipler=1.0, triangulate_1d=1)
    msckf("msckf_dof_beyond_table", synth.make_problem(4, C=64, K=4, F=3, seed=18), compact=True, chi2_multipler=1.0)        # Q8

    # ---- UpdaterSLAM::update --------------------------------------------------------------------------------------------
    for rep, nm in ((R.REP_GLOBAL_3D, "global3d"), (R.REP_ANCHORED_MSCKF_INVERSE_DEPTH, "anchored_msckf"),
                    (R.REP_ANCHORED_INVERSE_DEPTH_SINGLE, "single_depth"), (R.REP_ANCHORED_FULL_INVERSE_DEPTH, "anchored_full")):
        prob = synth.make_slam_problem(2, L=5, lm_rep=rep, C=9, K=2, seed=20 + rep)
        opts = capi.default_options(chi2_multipler=1.0)
        v = capi.Views(prob)
        sig, mult = aruco(v.features.F, 3)
        out = pyref.slam_update(opts, v, feat_sigma=sig, feat_chi2mult=mult)
        assert (out["feat_status"] == capi.FEAT_USED).sum() >= 3
        save(f"slam_update_{nm}_aruco", pack("slam_update", prob, opts, out, feat_sigma=sig, feat_chi2mult=mult))

    # ... SLAM landmarks (feat_rep_slam) and ArUco corners (feat_rep_aruco != feat_rep_slam) in ONE update (UpdaterSLAM.cpp:336-341, :427-447):
    # the anchored MSCKF inverse depth next to global xyz; the 1-dof single depth next to anchored xyz; all six at once
    for each, nm in (([4, 0, 4, 0, 4, 4], "msckf_and_global3d"), ([5, 2, 5, 2, 5, 5], "single_depth_and_anchored3d"), ([0, 1, 2, 3, 4, 5], "all_six")):
        each = np.array(each, np.int32)
        prob = synth.make_slam_problem(2, L=6, lm_rep=each, C=9, K=2, seed=60 + int(each[1]))
        opts = capi.default_options(chi2_multipler=1.0)
        v = capi.Views(prob)
        tag = each == each[1]  # the corners: the second representation, with the ArUco option set
        sig, mult = np.where(tag, 2.5, 1.0), np.where(tag, 3.0, 1.0)
        out = pyref.slam_update(opts, v, feat_sigma=sig, feat_chi2mult=mult)
        assert (out["feat_status"] == capi.FEAT_USED).sum() >= 4
        save(f"slam_update_mixed_{nm}", pack("slam_update", prob, opts, out, feat_sigma=sig, feat_chi2mult=mult))

    # ---- UpdaterSLAM::delayed_init (chain of StateHelper::initialize) --------------------------------------------------------
    # ... with the ArUco corners initialised in feat_rep_aruco, the other features in feat_rep_slam (:160-166)
    for rep_slam, rep_aruco, nm in ((R.REP_ANCHORED_MSCKF_INVERSE_DEPTH, R.REP_GLOBAL_3D, "msckf_and_global3d"),
                                    (R.REP_GLOBAL_3D, R.REP_ANCHORED_INVERSE_DEPTH_SINGLE, "global3d_and_single_depth")):
        prob = synth.make_problem(2, F=6, C=12, K=2, outlier_frac=0.3, seed=70 + rep_slam)
        opts = capi.default_options(chi2_multipler=1.0)
        v = capi.Views(prob)
        sig, mult = aruco(6, 5)
        each = np.where(sig != 1.0, rep_aruco, rep_slam).astype(np.int32)
        assert 1 <= (each == rep_aruco).sum() < 6
        out = pyref.slam_delayed_init(opts, v, feat_rep=rep_slam, feat_sigma=sig, feat_chi2mult=mult, feat_rep_aruco=rep_aruco)
        acc = out["lm_cov_id"] >= 0
        assert 2 <= acc.sum() and len(set(each[acc].tolist())) == 2, (acc, each)
        save(f"delayed_init_mixed_{nm}", pack("delayed_init", prob, opts, out, feat_rep=rep_slam, feat_sigma=sig, feat_chi2mult=mult, feat_rep_each=each))

    for rep, nm in ((R.REP_GLOBAL_3D, "global3d"), (R.REP_ANCHORED_MSCKF_INVERSE_DEPTH, "anchored_msckf"),
                    (R.REP_ANCHORED_INVERSE_DEPTH_SINGLE, "single_depth")):
        prob = synth.make_problem(2, F=6, C=12, K=2, outlier_frac=0.3, seed=30 + rep)
        opts = capi.default_options(chi2_multipler=1.0)
        v = capi.Views(prob)
        sig, mult = aruco(6, 5)
        out = pyref.slam_delayed_init(opts, v, feat_rep=rep, feat_sigma=sig, feat_chi2mult=mult)
        acc = (out["lm_cov_id"] >= 0).sum()
        assert 2 <= acc, acc
        save(f"delayed_init_{nm}", pack("delayed_init", prob, opts, out, feat_rep=rep, feat_sigma=sig, feat_chi2mult=mult))

    # ---- UpdaterSLAM::perform_anchor_change ----------------------------------------------------------------------------------
    for rep, nm in ((R.REP_ANCHORED_3D, "anchored3d"), (R.REP_ANCHORED_MSCKF_INVERSE_DEPTH, "anchored_msckf"),
                    (R.REP_ANCHORED_INVERSE_DEPTH_SINGLE, "single_depth")):
        prob = synth.make_slam_problem(2, L=4, lm_rep=rep, C=9, K=2, seed=40 + rep)
        opts = capi.default_options(chi2_multipler=1.0)
        v = capi.Views(prob)
        l, new_cam, new_clone = 1, 1 - int(prob.lm_anchor_cam[1]), prob.C - 1
        out = pyref.anchor_change(opts, v, l, new_cam, new_clone)
        assert out["rc"] == 0
        save(f"anchor_change_{nm}", pack("anchor_change", prob, opts, out, l=l, new_cam=new_cam, new_clone=new_clone))

    # ---- StateHelper::EKFPropagation -> augment_clone (+ dt Jacobian) -> marginalize -------------------------------------------
    prob = synth.make_problem(2, F=2, C=6, K=1, seed=50)
    opts = capi.default_options()
    rng = np.random.default_rng(50)
    Phi = np.eye(15) + 0.05 * rng.normal(size=(15, 15))
    Q = np.diag(rng.uniform(1e-6, 1e-4, 15))
    Q[0, 3] = 1e-5  # only the upper triangle is read (StateHelper.cpp:87)
    P1 = pyref.propagate_imu(opts, capi.Views(prob), Phi, Q)
    p1 = synth.make_problem(2, F=2, C=6, K=1, seed=50)
    p1.P = P1
    imu = np.concatenate([synth.boxplus_pose(prob.clone_q_p[-1], 0.01 * rng.normal(size=6)), rng.normal(size=3), 0.01 * rng.normal(size=6)])
    last_w = 0.3 * rng.normal(size=3)
    P2, clone = pyref.augment_clone(opts, capi.Views(p1), imu, last_w)
    # marginalise the oldest clone of the grown window
    p2 = synth.make_problem(2, F=2, C=6, K=1, seed=50)
    p2.P, p2.N, p2.C = P2, prob.N + 6, prob.C + 1
    p2.clone_q_p = np.vstack([prob.clone_q_p, clone[None, :]])
    p2.clone_q_p_fej = np.vstack([prob.clone_q_p_fej, clone[None, :]])
    p2.clone_cov_id = np.concatenate([prob.clone_cov_id, [prob.N]]).astype(np.int32)
    P3 = pyref.marginalize_clone(opts, capi.Views(p2), 0)
    save("window_propagate_clone_marginalize", pack("window", prob, opts, dict(P1=P1, P2=P2, P3=P3, clone=clone), Phi=Phi, Q=Q, imu=imu, last_w=last_w))


if __name__ == "__main__":
    main()
