"""The oracle against the REFERENCE'S OWN CODE (oracle/_ref/libov_ref.so: rpng/open_vins' update-path sources compiled from
/root/reference against the stand-in Eigen / Boost / OpenCV headers of oracle/ref/standin, driven by oracle/ref/ref_driver.cpp).

This is the pin of oracle/ov_oracle.cpp: every entry point the oracle restates -- triangulation, Jacobians, nullspace projection,
gate, compression, EKF update, SLAM update, delayed initialisation, anchor change, window bookkeeping -- is compared with what the
reference's classes compute on the same inputs.  CPU only.  The library can only be BUILT where /root/reference exists; a prebuilt
oracle/_ref travels with the snapshot.  Where neither exists the tests skip, and tests/test_ref_fixtures.py holds the oracle (and
the GPU) to the fixtures this library generated (tests/golden/ref_*.npz, tools/make_ref_fixtures.py).
"""
import ctypes

import numpy as np
import pytest

from open_vins_amd import capi, synth
from oracle import pyoracle, pyref

pytestmark = pytest.mark.skipif(not pyref.available(), reason="oracle/_ref cannot be built here (/root/reference absent) and no prebuilt library")

# oracle and reference are two float64 evaluations of the same algorithm, same operation order up to the summation order inside
# products: round-off level agreement is the bar
TOL_POS, TOL_DX, TOL_P, TOL_VAL = 1e-10, 1e-11, 1e-12, 1e-11


def _rel(a, b):
    m = np.isfinite(a) & np.isfinite(b)
    assert m.any()
    return np.linalg.norm(a[m] - b[m]) / max(np.linalg.norm(b[m]), 1e-300)


def _anchors_of(v, anchor_meas):
    fv = v.features
    ci = np.ctypeslib.as_array(fv.clone_idx, (fv.M,))
    ki = np.ctypeslib.as_array(fv.cam_idx, (fv.M,))
    return ki[anchor_meas], ci[anchor_meas]


def test_chi2_table_of_the_stand_in_matches_scipy_and_the_oracle():
    from scipy.stats import chi2
    for k in list(range(1, 500)) + [500, 509, 800, 2397]:
        q = pyref.chi2_quantile_95(k)
        assert abs(q / chi2.ppf(0.95, k) - 1) < 1e-13
        assert abs(q / pyoracle.chi2_quantile_95(k) - 1) < 1e-13


def test_givens_rotation_semantics():
    lib = pyoracle.load()
    for p, q in ((1.0, 2.0), (-3.0, 1e-3), (0.0, 2.0), (2.0, 0.0), (-1.0, -1.0), (0.0, -2.0), (-2.0, 0.0), (1e-200, 1e200), (3.0, -4.0)):
        c, s = ctypes.c_double(), ctypes.c_double()
        lib.oracle_make_givens(ctypes.c_double(p), ctypes.c_double(q), ctypes.byref(c), ctypes.byref(s))
        assert pyref.make_givens(p, q) == (c.value, s.value)
        cc, ss = pyref.make_givens(p, q)  # G^T (p, q) = (r, 0), r >= 0: rows (x, y) <- (c x - s y, s x + c y)
        assert abs(ss * p + cc * q) <= 1e-15 * np.hypot(p, q) and cc * p - ss * q >= 0


@pytest.mark.parametrize("fisheye", [0, 1])
def test_camera_models_pixels_bit_exact(fisheye):
    """CamRadtan / CamEqui::distort_d through the reference's classes: the float casts of CamBase.h:130-135 give bit-identical pixels;
    Jacobians identical too (same expressions)."""
    rng = np.random.default_rng(1 + fisheye)
    cam = np.array([458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])
    if fisheye:
        cam[4:] = [-0.013, 0.02, -0.012, 0.0025]
    lib = pyoracle.load()
    for _ in range(2000):
        zn = rng.uniform(-0.9, 0.9, 2)
        uv, A, B = pyref.cam_distort(cam, fisheye, zn)
        uv2, A2, B2 = np.zeros(2), np.zeros(4), np.zeros(16)
        lib.oracle_cam_distort(pyoracle._p(cam), fisheye, pyoracle._p(zn), pyoracle._p(uv2), pyoracle._p(A2), pyoracle._p(B2))
        assert np.array_equal(uv, uv2)
        np.testing.assert_allclose(A.ravel(), A2, rtol=1e-14, atol=0)
        np.testing.assert_allclose(B.ravel(), B2, rtol=1e-14, atol=1e-300)


@pytest.mark.parametrize("fisheye", [0, 1])
def test_undistortion_of_the_synthetic_front_end(fisheye):
    """synth.py's undistortion (upstream of the path: it makes uv_norm) against CamBase::undistort_d over the stand-in of
    cv::undistortPoints: same fixed-point iterations, float output."""
    rng = np.random.default_rng(7)
    cam = np.array([458.654, 457.296, 367.215, 248.375, -0.28340811, 0.07395907, 0.00019359, 1.76187114e-05])
    if fisheye:
        cam[4:] = [-0.013, 0.02, -0.012, 0.0025]
    und = synth.equi_undistort if fisheye else synth.radtan_undistort
    for _ in range(300):
        uv = np.float32(rng.uniform([40, 40], [700, 440]))
        ref = pyref.cam_undistort(cam, fisheye, uv.astype(np.float64))
        mine = np.float32(und(cam, float(uv[0]), float(uv[1])))
        assert np.abs(ref - mine.astype(np.float64)).max() < 2e-7  # one float ulp at |x| < 1


def test_nullspace_projection_and_compression_bit_exact():
    """UpdaterHelper::nullspace_project_inplace / measurement_compress_inplace: the oracle applies the same Givens rotations in the
    same order, so the results are identical to the last bit (including the rows <= cols early return)."""
    rng = np.random.default_rng(0)
    for rows, cols in ((24, 40), (9, 6), (117, 208)):
        Hf, Hx, r = rng.normal(size=(rows, 3)), rng.normal(size=(rows, cols)), rng.normal(size=rows)
        _, a1, a2 = pyoracle.nullspace_project(Hf, Hx, r)
        b1, b2 = pyref.nullspace_project(Hf, Hx, r)
        assert np.array_equal(a1, b1) and np.array_equal(a2, b2)
    for rows, cols in ((300, 40), (41, 40), (40, 40), (30, 40), (1, 5)):
        Hx, r = rng.normal(size=(rows, cols)), rng.normal(size=rows)
        a1, a2 = pyoracle.measurement_compress(Hx, r)
        b1, b2 = pyref.measurement_compress(Hx, r)
        assert a1.shape == b1.shape == (min(rows, cols), cols)
        assert np.array_equal(a1, b1) and np.array_equal(a2, b2)


@pytest.mark.parametrize("okw,pkw", [(dict(), dict()), (dict(), dict(track="ragged")), (dict(), dict(fisheye=True)), (dict(), dict(K=1, C=12)),
                                     (dict(), dict(cfg=4, F=40)), (dict(triangulate_1d=1), dict()), (dict(refine_features=0), dict()),
                                     (dict(triangulate_1d=1, refine_features=0), dict(track="ragged")), (dict(), dict(outlier_frac=0.3, pose_noise=3.0))])
def test_triangulation_against_the_reference(okw, pkw):
    """FeatureInitializer::single_triangulation(_1d) + single_gaussnewton, called as UpdaterMSCKF.cpp:117-142 calls them: identical
    verdicts (which of the two stages rejected), identical anchors (the tie rule over unordered_map iteration order included),
    positions at round-off."""
    pkw = dict(pkw)
    prob = synth.make_problem(pkw.pop("cfg", 2), F=pkw.pop("F", 60), **pkw)
    opts = capi.default_options(chi2_multipler=1.0, **okw)
    v = capi.Views(prob)
    a, b = pyoracle.triangulate(opts, v), pyref.triangulate(opts, v)
    assert np.array_equal(a["status"], b["status"])
    cam, clone = _anchors_of(v, a["anchor_meas"])
    assert np.array_equal(cam, b["anchor_cam"]) and np.array_equal(clone, b["anchor_clone"])
    ok = a["status"] == capi.FEAT_USED
    assert ok.sum() >= 10
    assert np.abs(a["p_FinG"] - b["p_FinG"])[ok].max() < TOL_POS and np.abs(a["p_FinA"] - b["p_FinA"])[ok].max() < TOL_POS


@pytest.mark.parametrize("rep", range(6))
@pytest.mark.parametrize("fej", [0, 1])
def test_feature_jacobian_against_the_reference(rep, fej):
    """UpdaterHelper::get_feature_jacobian_full (+ _representation): H_f, H_x (compared column by column through the covariance
    ids), residual -- every representation, with and without first-estimate Jacobians."""
    prob = synth.make_problem(2, F=6, C=12, fisheye=bool(rep % 2))
    opts = capi.default_options(do_fej=fej, feat_rep_msckf=rep)
    v = capi.Views(prob)
    tri = pyoracle.triangulate(opts, v)
    cols = pyoracle.column_map(opts, v)
    jrep = capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH if rep == capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE else rep  # UpdaterMSCKF.cpp:180-183
    cam, clone = _anchors_of(v, tri["anchor_meas"])
    for f in range(6):
        Hf, Hx, res = pyoracle.feature_jacobian(opts, v, f, tri["p_FinG"][f], tri["p_FinA"][f], int(tri["anchor_meas"][f]))
        Hf2, Hx2, res2 = pyref.feature_jacobian(opts, v, f, jrep, tri["p_FinG"][f], tri["p_FinA"][f], int(cam[f]), int(clone[f]))
        assert np.array_equal(res, res2)  # float-rounded pixels minus float pixels: exact
        if Hf.shape[1] == 1:  # the oracle's entry hands the single depth its own column (UpdaterHelper.cpp:178-189): d p / d rho
            Hf2 = Hf2[:, 2:]
        np.testing.assert_allclose(Hf, Hf2, rtol=0, atol=1e-12 * np.abs(Hf2).max())
        np.testing.assert_allclose(Hx, Hx2[:, cols], rtol=0, atol=1e-12 * np.abs(Hx2).max())
        rest = np.setdiff1d(np.arange(prob.N), cols)
        assert not Hx2[:, rest].any()


def _msckf_case(seed):
    rng = np.random.default_rng(1000 + seed)
    C, K, F = int(rng.integers(3, 41)), int(rng.integers(1, 5)), int(rng.integers(1, 90))
    kw = dict(C=C, K=K, F=F, track=("full", "ragged")[int(rng.integers(2))], fisheye=bool(rng.integers(2)), seed=int(rng.integers(1 << 20)),
              outlier_frac=float(rng.choice([0.0, 0.0, 0.3])), min_obs=int(rng.integers(2, 6)))
    rep = int(rng.integers(0, 6))
    flags = dict(do_fej=int(rng.integers(2)), do_calib_camera_pose=int(rng.integers(2)), do_calib_camera_intrinsics=int(rng.integers(2)))
    prob = synth.make_problem(int(rng.choice([2, 4])), rep, **kw)
    opts = capi.default_options(chi2_multipler=float(rng.choice([1.0, 5.0])), feat_rep_msckf=rep, **flags)
    return prob, opts


def _check_msckf(a, b):
    assert np.array_equal(a["feat_status"], b["feat_status"])
    used = a["feat_status"] == capi.FEAT_USED
    tri_ok = (a["feat_status"] == capi.FEAT_USED) | (a["feat_status"] == capi.FEAT_CHI2_REJECTED)
    if tri_ok.any():
        assert np.abs(a["p_FinG"] - b["p_FinG"])[tri_ok].max() < TOL_POS
    if used.any():
        assert _rel(b["dx"], a["dx"]) < TOL_DX
        assert _rel(b["P"], a["P"]) < TOL_P
    else:
        m = np.isfinite(b["P"])
        assert np.array_equal(b["P"][m], a["P"][m])  # nothing accepted: the state is untouched
    for k in ("clone_q_p", "calib_q_p", "intrinsics"):
        assert np.abs(a[k] - b[k]).max() < 1e-10 * max(1.0, np.abs(a[k]).max())


@pytest.mark.parametrize("seed", range(40))
def test_msckf_update_against_the_reference_random_shapes(seed):
    """The complete UpdaterMSCKF::update, the same 40 seeded shapes the GPU parity suite sweeps (3-40 clones, 1-4 cameras, 1-89 features,
    ragged / full tracks, both lens models, six representations, FEJ and calibration flags, outliers): identical accept / reject sets
    with the stage that rejected, triangulated positions, dx (recovered by box-minus from the reference's state), P', the posterior
    clone / calibration tables."""
    prob, opts = _msckf_case(seed)
    v = capi.Views(prob)
    _check_msckf(pyoracle.msckf_update(opts, v), pyref.msckf_update(opts, v))


def test_msckf_update_with_a_residual_beyond_the_chi2_table():
    """2m - 3 >= 500 leaves the precomputed table (UpdaterMSCKF.cpp:216-222): 64 clones x 4 cameras, a 256-observation track."""
    prob = synth.make_problem(4, C=64, K=4, F=6)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    assert (2 * np.diff(v.meas_offsets) - 3).max() >= 500
    a = pyoracle.msckf_update(opts, v)
    _check_msckf(a, pyref.msckf_update(opts, v))
    assert a["chi2_thresh"].max() > pyoracle.chi2_quantile_95(499)


def test_msckf_update_whose_stack_is_not_compressed():
    """rows <= cols skips measurement_compress_inplace (UpdaterHelper.cpp:459-460)."""
    prob = synth.make_problem(2, F=1, track="ragged", min_obs=3, seed=5)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    a = pyoracle.msckf_update(opts, v, want_compressed=True)
    assert a["feat_status"][0] == capi.FEAT_USED and a["rows_comp"] < a["D"]
    _check_msckf(a, pyref.msckf_update(opts, v))


def test_msckf_update_with_imu_intrinsics_in_the_state():
    """BASELINE configs[2] "online cam/IMU calib": 24 more rows of P (dw, da, tg, R_GYROtoIMU: State.cpp:65-88) that never get Jacobian
    columns but are corrected through their correlations."""
    prob = synth.make_problem(2, F=40, C=12, imu_intrinsics=True)
    assert prob.N == 16 + 24 + 6 * 12 + 14 * 2
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    a, b = pyoracle.msckf_update(opts, v), pyref.msckf_update(opts, v)
    _check_msckf(a, b)
    blk = slice(15, 39)
    assert np.abs(a["dx"][blk]).max() > 1e-6 and np.isfinite(b["dx"][blk]).all()


@pytest.mark.parametrize("rep", range(6))
def test_slam_update_against_the_reference(rep):
    prob = synth.make_slam_problem(2, L=10, lm_rep=rep)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    a, b = pyoracle.slam_update(opts, v), pyref.slam_update(opts, v)
    assert np.array_equal(a["feat_status"], b["feat_status"]) and (a["feat_status"] == capi.FEAT_USED).sum() >= 6
    assert _rel(b["dx"], a["dx"]) < TOL_DX and _rel(b["P"], a["P"]) < TOL_P
    assert np.abs(a["landmarks"] - b["landmarks"]).max() < TOL_VAL


@pytest.mark.parametrize("seed", range(12))
def test_slam_update_against_the_reference_random_shapes(seed):
    rng = np.random.default_rng(2000 + seed)
    kw = dict(C=int(rng.integers(6, 31)), K=int(rng.integers(1, 4)), track=("full", "ragged")[int(rng.integers(2))], fisheye=bool(rng.integers(2)),
              seed=int(rng.integers(1 << 20)))
    prob = synth.make_slam_problem(2, L=int(rng.integers(1, 13)), lm_rep=int(rng.integers(0, 6)), **kw)
    opts = capi.default_options(chi2_multipler=float(rng.choice([1.0, 5.0])), do_fej=int(rng.integers(2)),
                                do_calib_camera_pose=int(rng.integers(2)), do_calib_camera_intrinsics=int(rng.integers(2)))
    v = capi.Views(prob)
    a, b = pyoracle.slam_update(opts, v), pyref.slam_update(opts, v)
    assert np.array_equal(a["feat_status"], b["feat_status"])
    if (a["feat_status"] == capi.FEAT_USED).any():
        assert _rel(b["dx"], a["dx"]) < TOL_DX and _rel(b["P"], a["P"]) < TOL_P
    assert np.abs(a["landmarks"] - b["landmarks"]).max() < TOL_VAL


@pytest.mark.parametrize("rep", [capi.REP_GLOBAL_3D, capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH, capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE])
def test_slam_update_with_aruco_options_against_the_reference(rep):
    """UpdaterSLAM.cpp:392-409, :444: tag corners (feature id < max_aruco_features) carry their own sigma and chi2 multiplier; R_big is
    diagonal but not isotropic."""
    prob = synth.make_slam_problem(2, L=12, lm_rep=rep, C=16)  # (a 16-clone window: the reference's EKFUpdate through the stand-in Eigen is cubic in the state)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    tag = np.random.default_rng(3).random(v.features.F) < 0.4
    sig, mult = np.where(tag, 2.5, 1.0), np.where(tag, 3.0, 1.0)
    a = pyoracle.slam_update(opts, v, feat_sigma=sig, feat_chi2mult=mult)
    b = pyref.slam_update(opts, v, feat_sigma=sig, feat_chi2mult=mult)
    assert np.array_equal(a["feat_status"], b["feat_status"])
    assert _rel(b["dx"], a["dx"]) < TOL_DX and _rel(b["P"], a["P"]) < TOL_P
    assert _rel(pyref.slam_update(opts, v)["dx"], a["dx"]) > 1e-3  # the options matter


MIXED_REPS = [  # feat_rep_slam next to feat_rep_aruco (StateOptions.h:89-95), and every representation at once
    [capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH, capi.REP_GLOBAL_3D],
    [capi.REP_ANCHORED_3D, capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE],
    [capi.REP_GLOBAL_FULL_INVERSE_DEPTH, capi.REP_ANCHORED_FULL_INVERSE_DEPTH],
    [capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE, capi.REP_GLOBAL_FULL_INVERSE_DEPTH],
    [0, 1, 2, 3, 4, 5],
]  # (on a 16-clone window: the reference's EKFUpdate through the stand-in Eigen is cubic in the state, ~20 s per call at 30 clones)


@pytest.mark.parametrize("reps", MIXED_REPS, ids=lambda r: "-".join(map(str, r)))
def test_slam_update_mixed_representations_against_the_reference(reps):
    """UpdaterSLAM.cpp:336-341, :427-447: the representation is read from each landmark, and SLAM landmarks and ArUco corners kept in
    different representations (feat_rep_slam != feat_rep_aruco) share ONE Hx_big / R_big and one EKFUpdate."""
    L = 12
    each = np.array([reps[l % len(reps)] for l in range(L)], np.int32)
    prob = synth.make_slam_problem(2, L=L, lm_rep=each, C=16)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    tag = each == reps[-1]  # the corners carry the ArUco option set as well
    sig, mult = np.where(tag, 2.5, 1.0), np.where(tag, 3.0, 1.0)
    for kw in (({}, dict(feat_sigma=sig, feat_chi2mult=mult)) if len(reps) > 2 else ({},)):  # (the ArUco option set with the six-way mix)
        a, b = pyoracle.slam_update(opts, v, **kw), pyref.slam_update(opts, v, **kw)
        assert np.array_equal(a["feat_status"], b["feat_status"]) and (a["feat_status"] == capi.FEAT_USED).sum() >= 6
        assert _rel(b["dx"], a["dx"]) < TOL_DX and _rel(b["P"], a["P"]) < TOL_P
        assert np.abs(a["landmarks"] - b["landmarks"]).max() < TOL_VAL
    # ... and it is not what two passes give: the second pass is linearised at the state the first one left
    first = pyoracle.slam_update(opts, v)
    assert first["stats"]["n_rows"] == sum(2 * int(prob.meas_offsets[f + 1] - prob.meas_offsets[f]) - (2 if each[f] == 5 else 0)
                                           for f in range(L) if first["feat_status"][f] == capi.FEAT_USED)


@pytest.mark.parametrize("rep_slam,rep_aruco", [(capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH, capi.REP_GLOBAL_3D), (capi.REP_GLOBAL_3D, capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE),
                                                (capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE, capi.REP_ANCHORED_3D)])
def test_delayed_init_mixed_representations_against_the_reference(rep_slam, rep_aruco):
    """UpdaterSLAM.cpp:160-166: an ArUco corner is initialised in feat_rep_aruco, every other feature in feat_rep_slam, one chain."""
    prob = synth.make_problem(2, F=16, outlier_frac=0.2)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    tag = np.random.default_rng(5).random(16) < 0.4
    each = np.where(tag, rep_aruco, rep_slam).astype(np.int32)
    a = pyoracle.slam_delayed_init(opts, v, feat_rep=rep_slam, feat_rep_each=each)
    b = pyref.slam_delayed_init(opts, v, feat_rep=rep_slam, feat_rep_aruco=rep_aruco, feat_is_aruco=tag)
    assert 4 <= (a["lm_cov_id"] >= 0).sum() < 16
    _check_delayed_init(a, b)
    sig, mult = np.where(tag, 2.5, 1.0), np.where(tag, 3.0, 1.0)
    _check_delayed_init(pyoracle.slam_delayed_init(opts, v, feat_rep=rep_slam, feat_rep_each=each, feat_sigma=sig, feat_chi2mult=mult),
                        pyref.slam_delayed_init(opts, v, feat_rep=rep_slam, feat_rep_aruco=rep_aruco, feat_sigma=sig, feat_chi2mult=mult))


def test_delayed_init_beside_landmarks_of_other_representations():
    """New landmarks in one representation next to resident ones in others (an ArUco batch initialised after SLAM landmarks exist)."""
    each = np.array([0, 4, 5, 2, 4, 5], np.int32)
    prob = synth.make_slam_problem(2, L=6, lm_rep=each, C=14)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    for rep in (capi.REP_GLOBAL_3D, capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE):
        a = pyoracle.slam_delayed_init(opts, v, feat_rep=rep)
        b = pyref.slam_delayed_init(opts, v, feat_rep=rep)
        assert (a["lm_cov_id"] >= 0).sum() >= 3
        _check_delayed_init(a, b)
        np.testing.assert_allclose(a["landmarks_existing"], b["landmarks_existing"], rtol=TOL_VAL, atol=TOL_VAL)


def test_change_anchors_mixed_representations_against_the_reference():
    """UpdaterSLAM::change_anchors skips the global landmarks and moves each anchored one in ITS representation (:493-500)."""
    each = np.array([4, 0, 5, 2, 1, 3, 4, 0, 5, 2], np.int32)
    prob = synth.make_slam_problem(2, L=10, lm_rep=each)
    opts = capi.default_options(chi2_multipler=1.0)
    b = pyref.change_anchors(opts, capi.Views(prob))
    ref = synth.make_slam_problem(2, L=10, lm_rep=each)
    moved = np.flatnonzero((prob.lm_anchor_clone == 0) & (each >= 2))
    assert len(moved) >= 2
    for l in moved:
        o = pyoracle.anchor_change(opts, capi.Views(ref), int(l), int(ref.lm_anchor_cam[l]), ref.C - 1)
        ref.P, ref.lm_value[l], ref.lm_fej[l], ref.lm_anchor_clone[l] = o["P"], o["value"], o["fej"], ref.C - 1
    assert np.array_equal(b["anchor_clone"], ref.lm_anchor_clone)
    assert _rel(b["P"], ref.P) < 1e-13
    np.testing.assert_allclose(b["value"], ref.lm_value, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(b["fej"], ref.lm_fej, rtol=1e-13, atol=1e-15)


def _check_delayed_init(a, b):
    assert np.array_equal(a["feat_status"], b["feat_status"])
    assert a["N"] == b["N"] and np.array_equal(a["lm_cov_id"], b["lm_cov_id"])
    acc = a["lm_cov_id"] >= 0
    if acc.any():
        np.testing.assert_allclose(a["lm_value"][acc], b["lm_value"][acc], rtol=TOL_VAL, atol=TOL_VAL)
        np.testing.assert_allclose(a["lm_fej"][acc], b["lm_fej"][acc], rtol=TOL_VAL, atol=TOL_VAL)
        anchored = acc & (a["anchor_cam"] >= 0)
        assert np.array_equal(a["anchor_cam"][anchored], b["anchor_cam"][anchored]) and np.array_equal(a["anchor_clone"][anchored], b["anchor_clone"][anchored])
    assert _rel(b["P"], a["P"]) < TOL_P
    for k in ("clone_q_p", "calib_q_p", "intrinsics"):
        assert np.abs(a[k] - b[k]).max() < 1e-10 * max(1.0, np.abs(a[k]).max())


@pytest.mark.parametrize("rep", range(6))
def test_delayed_init_against_the_reference(rep):
    """UpdaterSLAM::delayed_init: triangulation, then the chain of StateHelper::initialize -- Givens split of [H_f | H_x | r], gate
    of the lower block against chi2(2m), initialize_invertible's covariance augmentation, EKFUpdate with the rest --
    each on the state the previous feature left."""
    prob = synth.make_problem(2, F=16, outlier_frac=0.2)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    a, b = pyoracle.slam_delayed_init(opts, v, feat_rep=rep), pyref.slam_delayed_init(opts, v, feat_rep=rep)
    assert 4 <= (a["lm_cov_id"] >= 0).sum() < 16
    _check_delayed_init(a, b)


@pytest.mark.parametrize("seed", range(12))
def test_delayed_init_against_the_reference_random_shapes(seed):
    rng = np.random.default_rng(3000 + seed)
    kw = dict(C=int(rng.integers(6, 31)), K=int(rng.integers(1, 4)), F=int(rng.integers(1, 21)), track=("full", "ragged")[int(rng.integers(2))],
              fisheye=bool(rng.integers(2)), seed=int(rng.integers(1 << 20)), outlier_frac=float(rng.choice([0.0, 0.3])))
    rep = int(rng.integers(0, 6))
    prob = synth.make_problem(2, **kw)
    opts = capi.default_options(chi2_multipler=float(rng.choice([1.0, 5.0])), do_fej=int(rng.integers(2)),
                                do_calib_camera_pose=int(rng.integers(2)), do_calib_camera_intrinsics=int(rng.integers(2)))
    v = capi.Views(prob)
    _check_delayed_init(pyoracle.slam_delayed_init(opts, v, feat_rep=rep), pyref.slam_delayed_init(opts, v, feat_rep=rep))


@pytest.mark.parametrize("rep", [capi.REP_GLOBAL_3D, capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE])
def test_delayed_init_with_aruco_options_against_the_reference(rep):
    prob = synth.make_problem(2, F=16, outlier_frac=0.2)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    tag = np.random.default_rng(5).random(16) < 0.4
    sig, mult = np.where(tag, 2.5, 1.0), np.where(tag, 3.0, 1.0)
    _check_delayed_init(pyoracle.slam_delayed_init(opts, v, feat_rep=rep, feat_sigma=sig, feat_chi2mult=mult),
                        pyref.slam_delayed_init(opts, v, feat_rep=rep, feat_sigma=sig, feat_chi2mult=mult))


def test_delayed_init_beside_landmarks_already_in_the_state():
    prob = synth.make_slam_problem(2, L=6, lm_rep=capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH, C=14)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    a = pyoracle.slam_delayed_init(opts, v, feat_rep=capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH)
    b = pyref.slam_delayed_init(opts, v, feat_rep=capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH)
    assert (a["lm_cov_id"] >= 0).sum() >= 3
    _check_delayed_init(a, b)
    np.testing.assert_allclose(a["landmarks_existing"], b["landmarks_existing"], rtol=TOL_VAL, atol=TOL_VAL)


@pytest.mark.parametrize("rep", [capi.REP_ANCHORED_3D, capi.REP_ANCHORED_FULL_INVERSE_DEPTH, capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH,
                                 capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE])
def test_anchor_change_against_the_reference(rep):
    """UpdaterSLAM::perform_anchor_change (protected: reached through a deriving class in the driver): same camera / newest clone as
    change_anchors asks, and the other camera / an arbitrary clone.  This comparison found the one misreading the round-3 oracle had:
    Landmark::get_xyz(true) reads the CURRENT value for the two MSCKF inverse-depth representations (Landmark.cpp:47-59)."""
    prob = synth.make_slam_problem(2, L=10, lm_rep=rep)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    assert np.abs(prob.lm_value - prob.lm_fej).max() > 1e-4
    for l in (0, 3, 7):
        for new_cam, new_clone in ((int(prob.lm_anchor_cam[l]), prob.C - 1), (1 - int(prob.lm_anchor_cam[l]), 5)):
            a, b = pyoracle.anchor_change(opts, v, l, new_cam, new_clone), pyref.anchor_change(opts, v, l, new_cam, new_clone)
            assert a["rc"] == 0 and b["rc"] == 0
            assert _rel(b["P"], a["P"]) < 1e-14
            np.testing.assert_allclose(a["value"], b["value"], rtol=1e-13, atol=1e-15)
            np.testing.assert_allclose(a["fej"], b["fej"], rtol=1e-13, atol=1e-15)


@pytest.mark.parametrize("rep", [capi.REP_ANCHORED_3D, capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH, capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE])
def test_change_anchors_chain_against_the_reference(rep):
    """UpdaterSLAM::change_anchors as VioManager.cpp:585 calls it (window one clone over max_clones): every landmark anchored in the
    oldest clone moves to the newest; the oracle moves them one after the other on the covariance the previous move left."""
    prob = synth.make_slam_problem(2, L=10, lm_rep=rep)
    opts = capi.default_options(chi2_multipler=1.0)
    b = pyref.change_anchors(opts, capi.Views(prob))
    ref = synth.make_slam_problem(2, L=10, lm_rep=rep)
    moved = np.flatnonzero(prob.lm_anchor_clone == 0)
    assert len(moved) >= 3
    for l in moved:
        o = pyoracle.anchor_change(opts, capi.Views(ref), int(l), int(ref.lm_anchor_cam[l]), ref.C - 1)
        ref.P, ref.lm_value[l], ref.lm_fej[l], ref.lm_anchor_clone[l] = o["P"], o["value"], o["fej"], ref.C - 1
    assert np.array_equal(b["anchor_clone"], ref.lm_anchor_clone)
    assert _rel(b["P"], ref.P) < 1e-13
    np.testing.assert_allclose(b["value"], ref.lm_value, rtol=1e-13, atol=1e-15)
    np.testing.assert_allclose(b["fej"], ref.lm_fej, rtol=1e-13, atol=1e-15)


def test_window_bookkeeping_against_the_reference():
    """StateHelper::marginalize, clone / augment_clone (with the time-offset Jacobian), EKFPropagation on the reference's State."""
    prob = synth.make_problem(2, F=4)
    opts = capi.default_options()
    v = capi.Views(prob)
    for c in (0, 7, prob.C - 1):
        assert np.array_equal(pyoracle.marginalize(prob.P, int(prob.clone_cov_id[c]), 6), pyref.marginalize_clone(opts, v, c))
    rng = np.random.default_rng(0)
    imu = np.concatenate([prob.clone_q_p[-1], rng.normal(size=3), rng.normal(size=6) * 0.01])
    w = rng.normal(size=3) * 0.3
    a = pyoracle.augment_clone(prob.P, 0, 6, 15, np.concatenate([w, imu[7:10]]))
    bP, bclone = pyref.augment_clone(opts, v, imu, w)
    assert np.abs(a - bP).max() <= 1e-18 + 1e-15 * np.abs(a).max() and np.array_equal(bclone, imu[:7])
    Phi = np.eye(15) + 0.01 * rng.normal(size=(15, 15))
    Qh = rng.normal(size=(15, 15)) * 1e-3
    rc, a = pyoracle.propagate(prob.P, 0, np.arange(15), Phi, Qh @ Qh.T)
    assert rc == 0 and _rel(pyref.propagate_imu(opts, v, Phi, Qh @ Qh.T), a) < 1e-14


def test_ekf_update_against_the_reference():
    """StateHelper::EKFUpdate with a caller's system (what UpdaterZeroVelocity hands it): K = P H^T S^-1 on the upper triangle, mirrored."""
    prob = synth.make_problem(2, F=4, C=8)
    opts = capi.default_options()
    v = capi.Views(prob)
    cols = pyoracle.column_map(opts, v)
    rng = np.random.default_rng(2)
    H, r = rng.normal(size=(30, len(cols))), rng.normal(size=30) * 0.1
    st, Pa, dxa = pyoracle.ekf_update(prob.P, H, r, cols, 1.3)
    rc, Pb, dxb = pyref.ekf_update(opts, v, H, r, cols, 1.3)
    assert st == 0 and rc == 0
    assert _rel(dxb, dxa) < TOL_DX and _rel(Pb, Pa) < TOL_P and np.array_equal(Pb, Pb.T)
