"""GPU parity tests (`-m gpu`): SLAM landmarks of DIFFERENT representations in one state (ABI 7, round 5).

UpdaterSLAM reads the representation from each landmark (UpdaterSLAM.cpp:336-341) and initialises an ArUco corner in
StateOptions::feat_rep_aruco, every other feature in feat_rep_slam (:160-166): with feat_rep_aruco != feat_rep_slam ONE Hx_big / R_big
(:427-447) holds rows of landmarks of both, of 3 and of 1 state dof.  The HIP path through the C ABI against the CPU oracle — itself pinned to
the reference's own UpdaterSLAM on these states (tests/test_ref_build.py: *_mixed_representations_*) — and against a fixture the reference
generated (tests/test_ref_fixtures.py: ref_slam_update_mixed_reps).  Tolerances as tests/test_gpu_parity.py.
"""
import numpy as np
import pytest

from open_vins_amd import capi, synth

pytestmark = pytest.mark.gpu

TOL_CHI2 = 1e-8

MIXES = [
    [capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH, capi.REP_GLOBAL_3D],
    [capi.REP_GLOBAL_3D, capi.REP_ANCHORED_FULL_INVERSE_DEPTH],
    [capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE, capi.REP_GLOBAL_3D],
    [capi.REP_ANCHORED_3D, capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE],
    [0, 1, 2, 3, 4, 5],
]


@pytest.fixture(scope="module")
def Updater():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    from open_vins_amd.updater import UpdaterMSCKF
    return UpdaterMSCKF


def _rel(a, b):
    return np.linalg.norm(a - b) / max(np.linalg.norm(b), 1e-300)


def _each(reps, L):
    return np.array([reps[l % len(reps)] for l in range(L)], np.int32)


@pytest.mark.parametrize("reps", MIXES, ids=lambda r: "-".join(map(str, r)))
def test_slam_update_mixed_representations(Updater, oracle, reps):
    """One ovgpu_slam_update over landmarks of several representations: statuses, chi2, thresholds (dof 2m, or 2m - 2 for a single-depth
    landmark), dx, P', the landmarks' values; then the same with the ArUco option set on the corners, and mode A's compressed system."""
    L = 12
    each = _each(reps, L)
    prob = synth.make_slam_problem(2, L=L, lm_rep=each)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    tag = each == reps[-1]
    sig, mult = np.where(tag, 2.5, 1.0), np.where(tag, 3.0, 1.0)
    up = Updater(opts)
    for kw in ({}, dict(feat_sigma=sig, feat_chi2mult=mult)):
        ref = oracle.slam_update(opts, v, **kw)
        up.set_slam_problem(prob)
        if kw:
            up.set_feature_options(sig, mult)
        out = up.slam_update()
        assert np.array_equal(out["feat_status"], ref["feat_status"]) and (ref["feat_status"] == capi.FEAT_USED).sum() >= 6
        gate = np.isfinite(ref["chi2"])
        np.testing.assert_allclose(out["chi2"][gate], ref["chi2"][gate], rtol=TOL_CHI2)
        np.testing.assert_allclose(out["chi2_thresh"][gate], ref["chi2_thresh"][gate], rtol=1e-12)
        assert out["stats"]["n_rows"] == ref["stats"]["n_rows"] and out["stats"]["D"] == ref["stats"]["D"]
        assert _rel(out["dx"], ref["dx"]) < 1e-7 and _rel(out["P"], ref["P"]) < 1e-8
        np.testing.assert_allclose(out["landmarks"], ref["landmarks"], rtol=1e-9, atol=1e-11)
    lm = up.get_landmarks()
    assert np.array_equal(lm["feat_rep"], each)
    # mode A: the compressed stack through the stock EKFUpdate (here: the oracle's) gives the same posterior
    ref = oracle.slam_update(opts, v)
    up.set_slam_problem(prob)
    comp = up.slam_compress()
    assert np.array_equal(comp["feat_status"], ref["feat_status"]) and comp["D"] == ref["stats"]["D"]
    st, P1, dx1 = oracle.ekf_update(prob.P, comp["H"], comp["r"], comp["col_cov_id"], opts.sigma_pix ** 2)
    assert st == 0 and _rel(dx1, ref["dx"]) < 1e-7 and _rel(P1, ref["P"]) < 1e-8
    up.close()


@pytest.mark.parametrize("seed", range(8))
def test_slam_update_mixed_representations_random_shapes(Updater, oracle, seed):
    rng = np.random.default_rng(7000 + seed)
    L = int(rng.integers(2, 14))
    kw = dict(C=int(rng.integers(6, 31)), K=int(rng.integers(1, 4)), track=("full", "ragged")[int(rng.integers(2))], fisheye=bool(rng.integers(2)),
              seed=int(rng.integers(1 << 20)))
    each = rng.integers(0, 6, L).astype(np.int32)
    prob = synth.make_slam_problem(2, L=L, lm_rep=each, **kw)
    opts = capi.default_options(chi2_multipler=float(rng.choice([1.0, 5.0])), do_fej=int(rng.integers(2)),
                                do_calib_camera_pose=int(rng.integers(2)), do_calib_camera_intrinsics=int(rng.integers(2)))
    ref = oracle.slam_update(opts, capi.Views(prob))
    up = Updater(opts)
    up.set_slam_problem(prob)
    out = up.slam_update()
    assert np.array_equal(out["feat_status"], ref["feat_status"])
    gate = np.isfinite(ref["chi2"])
    np.testing.assert_allclose(out["chi2"][gate], ref["chi2"][gate], rtol=TOL_CHI2)
    if (ref["feat_status"] == capi.FEAT_USED).any():
        assert _rel(out["dx"], ref["dx"]) < 1e-7 and _rel(out["P"], ref["P"]) < 1e-8
    np.testing.assert_allclose(out["landmarks"], ref["landmarks"], rtol=1e-9, atol=1e-11)
    up.close()


def _check_delayed_init(out, ref, post):
    assert np.array_equal(out["feat_status"], ref["feat_status"])
    gate = np.isfinite(ref["chi2"])
    np.testing.assert_allclose(out["chi2"][gate], ref["chi2"][gate], rtol=1e-7)
    np.testing.assert_allclose(out["chi2_thresh"][gate], ref["chi2_thresh"][gate], rtol=1e-12)
    assert out["N"] == ref["N"] and np.array_equal(out["lm_cov_id"], ref["lm_cov_id"])
    acc = ref["lm_cov_id"] >= 0
    anchored = ref["anchor_cam"] >= 0
    assert np.array_equal(out["anchor_cam"][anchored], ref["anchor_cam"][anchored]) and np.array_equal(out["anchor_clone"][anchored], ref["anchor_clone"][anchored])
    np.testing.assert_allclose(out["lm_value"][acc], ref["lm_value"][acc], rtol=1e-8, atol=1e-10)
    np.testing.assert_allclose(out["lm_fej"][acc], ref["lm_fej"][acc], rtol=1e-12, atol=1e-14)
    assert _rel(out["dx_seq"], ref["dx_seq"]) < 1e-6 and not out["dx_seq"][~acc].any()
    assert _rel(out["P"], ref["P"]) < 1e-7
    for k in ("clone_q_p", "calib_q_p", "intrinsics"):
        assert np.abs(post[k] - ref[k]).max() < 1e-9


@pytest.mark.parametrize("rep_slam,rep_aruco", [(capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH, capi.REP_GLOBAL_3D), (capi.REP_GLOBAL_3D, capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE),
                                                (capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE, capi.REP_ANCHORED_3D)])
def test_delayed_init_per_feature_representations(Updater, oracle, rep_slam, rep_aruco):
    """ovgpu_set_feature_reps: ArUco corners initialised in feat_rep_aruco, the other features in feat_rep_slam, in ONE chain
    (UpdaterSLAM.cpp:160-166): the covariance grows by 3 or 1 per accepted feature as each one's representation says."""
    prob = synth.make_problem(2, F=16, outlier_frac=0.2)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    tri = oracle.triangulate(opts, v)
    tag = np.random.default_rng(5).random(16) < 0.4
    each = np.where(tag, rep_aruco, rep_slam).astype(np.int32)
    sig, mult = np.where(tag, 2.5, 1.0), np.where(tag, 3.0, 1.0)
    ref = oracle.slam_delayed_init(opts, v, feat_rep=rep_slam, tri=tri, feat_rep_each=each, feat_sigma=sig, feat_chi2mult=mult)
    acc = ref["lm_cov_id"] >= 0
    assert ref["rc"] == 0 and 4 <= acc.sum() < 16
    up = Updater(opts)
    up.set_problem(prob)
    up.set_triangulation(tri["p_FinG"], tri["p_FinA"], tri["anchor_meas"], tri["status"])
    up.set_feature_options(sig, mult)
    out = up.delayed_init(rep_slam, feat_rep_each=each)
    _check_delayed_init(out, ref, up.get_state(P=True))
    lm = up.get_landmarks()
    assert np.array_equal(lm["cov_id"], ref["lm_cov_id"][acc]) and np.array_equal(lm["feat_rep"], each[acc])
    # the per-feature representations belong to the batch: the next one is initialised in the call's
    up.set_problem(prob)
    up.set_triangulation(tri["p_FinG"], tri["p_FinA"], tri["anchor_meas"], tri["status"])
    out2 = up.delayed_init(rep_slam)
    ref2 = oracle.slam_delayed_init(opts, v, feat_rep=rep_slam, tri=tri)
    assert out2["N"] == ref2["N"] and np.array_equal(out2["lm_cov_id"], ref2["lm_cov_id"])
    up.close()


def test_delayed_init_beside_landmarks_of_other_representations_then_update(Updater, oracle):
    """New landmarks (one representation per call) next to resident ones of three others; the resident ones take every correction of the
    chain; then one SLAM update over old and new landmarks together."""
    each = np.array([0, 4, 5, 2, 4, 5], np.int32)
    prob = synth.make_slam_problem(2, L=6, lm_rep=each, C=14, seed=21)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    tri = oracle.triangulate(opts, v)
    for rep in (capi.REP_GLOBAL_FULL_INVERSE_DEPTH, capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE):
        ref = oracle.slam_delayed_init(opts, v, feat_rep=rep, tri=tri)
        acc = ref["lm_cov_id"] >= 0
        assert acc.sum() >= 3
        up = Updater(opts)
        up.set_slam_problem(prob)
        up.set_triangulation(tri["p_FinG"], tri["p_FinA"], tri["anchor_meas"], tri["status"])
        out = up.delayed_init(rep)
        _check_delayed_init(out, ref, up.get_state(P=True))
        lm = up.get_landmarks()
        assert lm["value"].shape[0] == 6 + acc.sum() and np.array_equal(lm["feat_rep"], np.r_[each, np.full(acc.sum(), rep, np.int32)])
        np.testing.assert_allclose(lm["value"][:6], ref["landmarks_existing"], rtol=1e-8, atol=1e-10)
        # ---- one update over all of them: the six old landmarks by their tracks, the new ones by the same tracks again
        nxt = synth.make_slam_problem(2, L=6, lm_rep=each, C=14, seed=21)
        post = up.get_state(P=True)
        nxt.N, nxt.P, nxt.clone_q_p, nxt.calib_q_p, nxt.intrinsics = out["N"], post["P"], post["clone_q_p"], post["calib_q_p"], post["intrinsics"]
        nxt.lm_value, nxt.lm_fej, nxt.lm_cov_id = lm["value"], lm["fej"], lm["cov_id"]
        nxt.lm_anchor_cam, nxt.lm_anchor_clone, nxt.lm_rep_each = lm["anchor_cam"], lm["anchor_clone"], lm["feat_rep"]
        nxt.lm_index = np.arange(6, dtype=np.int32)
        ref_u = oracle.slam_update(opts, capi.Views(nxt))
        up.set_features(nxt)
        out_u = up.slam_update(lm_index=nxt.lm_index)
        assert np.array_equal(out_u["feat_status"], ref_u["feat_status"])
        assert _rel(out_u["dx"], ref_u["dx"]) < 1e-7 and _rel(out_u["P"], ref_u["P"]) < 1e-8
        up.close()


def test_change_anchors_and_marginalize_in_a_mixed_state(Updater, oracle):
    """UpdaterSLAM::change_anchors on a state of global and anchored landmarks (the global ones are skipped, :493-496; each anchored one moves
    in ITS representation); StateHelper::marginalize of a 1-dof and of a 3-dof landmark; an update of what is left."""
    each = np.array([4, 0, 5, 2, 1, 3, 4, 0, 5, 2], np.int32)
    prob = synth.make_slam_problem(2, L=10, lm_rep=each)
    opts = capi.default_options(chi2_multipler=1.0)
    moved = np.flatnonzero((prob.lm_anchor_clone == 0) & (each >= 2))
    assert len(moved) >= 2
    up = Updater(opts)
    up.set_slam_problem(prob)
    assert up.change_anchors(0, prob.C - 1) == len(moved)
    ref = synth.make_slam_problem(2, L=10, lm_rep=each)
    for l in moved:
        o = oracle.anchor_change(opts, capi.Views(ref), int(l), int(ref.lm_anchor_cam[l]), ref.C - 1)
        assert o["rc"] == 0
        ref.P, ref.lm_value[l], ref.lm_fej[l], ref.lm_anchor_clone[l] = o["P"], o["value"], o["fej"], ref.C - 1
    lm = up.get_landmarks()
    np.testing.assert_allclose(lm["value"], ref.lm_value, rtol=1e-12, atol=1e-13)
    np.testing.assert_allclose(lm["fej"], ref.lm_fej, rtol=1e-12, atol=1e-13)
    np.testing.assert_array_equal(lm["anchor_clone"], ref.lm_anchor_clone)
    assert _rel(up.get_state(P=True)["P"], ref.P) < 1e-12
    with pytest.raises(RuntimeError):  # a global landmark has no anchor to change
        up.change_anchor(1, 0, 1)
    with pytest.raises(RuntimeError):  # a 1-dof landmark is not a 3-dof block
        up.state_marginalize(int(prob.lm_cov_id[2]), 3)
    # ---- landmark 2 (single depth, 1 dof) and landmark 4 (3 dof) leave the state
    up.state_marginalize(int(prob.lm_cov_id[4]), 3)
    up.state_marginalize(int(prob.lm_cov_id[2]), 1)
    lm2 = up.get_landmarks()
    keepf = np.array([0, 1, 3, 5, 6, 7, 8, 9])
    assert np.array_equal(lm2["feat_rep"], each[keepf])
    post = up.get_state(P=True)
    win = synth.make_slam_problem(2, L=10, lm_rep=each)
    cnt = (prob.meas_offsets[1:] - prob.meas_offsets[:-1])[keepf]
    ids = np.concatenate([np.arange(prob.meas_offsets[f], prob.meas_offsets[f + 1]) for f in keepf])
    win.meas_offsets = np.concatenate([[0], np.cumsum(cnt)]).astype(np.int32)
    win.uv, win.uvn = prob.uv.reshape(-1, 2)[ids].reshape(-1), prob.uvn.reshape(-1, 2)[ids].reshape(-1)
    win.clone_idx, win.cam_idx = prob.clone_idx[ids], prob.cam_idx[ids]
    win.N, win.P = prob.N - 4, post["P"]
    win.lm_value, win.lm_fej, win.lm_cov_id = lm2["value"], lm2["fej"], lm2["cov_id"]
    win.lm_anchor_cam, win.lm_anchor_clone, win.lm_rep_each = lm2["anchor_cam"], lm2["anchor_clone"], lm2["feat_rep"]
    win.lm_index = np.arange(8, dtype=np.int32)
    idx = np.r_[0:prob.lm_cov_id[2], prob.lm_cov_id[2] + 1:prob.lm_cov_id[4], prob.lm_cov_id[4] + 3:prob.N]
    assert _rel(post["P"], ref.P[np.ix_(idx, idx)]) < 1e-12
    ref_u = oracle.slam_update(opts, capi.Views(win))
    up.set_features(win)
    out = up.slam_update(lm_index=win.lm_index)
    assert np.array_equal(out["feat_status"], ref_u["feat_status"]) and (ref_u["feat_status"] == capi.FEAT_USED).sum() >= 4
    assert _rel(out["dx"], ref_u["dx"]) < 1e-7 and _rel(out["P"], ref_u["P"]) < 1e-8
    up.close()


def test_mixed_representation_edge_cases(Updater, oracle):
    """Bad inputs of the ABI 7 entry points: an unknown representation in feat_rep_each / ovgpu_set_feature_reps, per-feature representations
    without a batch, anchors missing for an anchored landmark; and the empty cases (no landmarks, per-feature reps cleared with NULL)."""
    import ctypes as C
    opts = capi.default_options(chi2_multipler=1.0)
    each = np.array([0, 4, 5, 2], np.int32)
    prob = synth.make_slam_problem(2, L=4, lm_rep=each)
    up = Updater(opts)
    v = capi.Views(prob)
    capi.check(up.lib.ovgpu_set_state(up._ctx, C.byref(v.state)), "ovgpu_set_state")
    bad = np.array([0, 4, 9, 2], np.int32)
    lv = capi.LandmarksView.from_buffer_copy(v.landmarks)
    lv.feat_rep_each = bad.ctypes.data_as(capi.c_int32_p)
    assert up.lib.ovgpu_set_landmarks(up._ctx, C.byref(lv)) == capi.ERR_INVALID
    lv = capi.LandmarksView.from_buffer_copy(v.landmarks)
    lv.anchor_cam = None  # landmarks 1 .. 3 are anchored
    assert up.lib.ovgpu_set_landmarks(up._ctx, C.byref(lv)) == capi.ERR_INVALID
    assert up.lib.ovgpu_set_feature_reps(up._ctx, each.ctypes.data_as(capi.c_int32_p)) == capi.ERR_NO_STATE  # no batch is resident
    up.set_slam_problem(prob)
    assert up.lib.ovgpu_set_feature_reps(up._ctx, bad.ctypes.data_as(capi.c_int32_p)) == capi.ERR_INVALID
    capi.check(up.lib.ovgpu_set_feature_reps(up._ctx, each.ctypes.data_as(capi.c_int32_p)), "ovgpu_set_feature_reps")
    capi.check(up.lib.ovgpu_set_feature_reps(up._ctx, None), "ovgpu_set_feature_reps")  # cleared: the call's feat_rep for every feature again
    lm = up.get_landmarks()
    assert np.array_equal(lm["feat_rep"], each) and np.array_equal(lm["anchor_cam"] >= 0, each >= 2)
    out = up.slam_update()
    ref = oracle.slam_update(opts, v)
    assert np.array_equal(out["feat_status"], ref["feat_status"]) and _rel(out["dx"], ref["dx"]) < 1e-7
    up.close()
