"""CPU tests that pin the oracle (oracle/ov_oracle.cpp).

The reference has no golden vectors for this path, so beside the known-answer fixtures of
tests/test_known_answer.py the oracle is checked against independent numpy / scipy computations of the same
quantities and against the algebraic invariants the reference's own derivations rely on (docs/update-null.dox,
docs/update-compress.dox).
"""
import ctypes as C

import numpy as np
import pytest
from scipy.stats import chi2 as sp_chi2

from open_vins_amd import capi, synth


def _views(cfg=2, **kw):
    prob = synth.make_problem(cfg, **kw)
    return prob, capi.Views(prob)


# --------------------------------------------------------------------------- chi2 table (UpdaterMSCKF.cpp:52-55)
def test_chi2_table_matches_scipy(oracle):
    for k in list(range(1, 500, 7)) + [117, 237, 397, 499, 500, 797, 1200]:
        assert oracle.chi2_quantile_95(k) == pytest.approx(sp_chi2.ppf(0.95, k), rel=1e-11)
    # values probed in SURVEY.md §8c(iv)
    assert oracle.chi2_quantile_95(1) == pytest.approx(3.841458820694124, rel=1e-12)
    assert oracle.chi2_quantile_95(117) == pytest.approx(143.24614728377486, rel=1e-12)


# --------------------------------------------------------------------------- Eigen makeGivens semantics
def test_make_givens_annihilates_second(oracle):
    lib = oracle.load()
    rng = np.random.default_rng(0)
    cases = [(0.0, 0.0), (1.5, 0.0), (-1.5, 0.0), (0.0, 2.0), (0.0, -2.0)] + [tuple(rng.normal(size=2)) for _ in range(50)]
    for p, q in cases:
        c, s = C.c_double(0), C.c_double(0)
        lib.oracle_make_givens(p, q, C.byref(c), C.byref(s))
        c, s = c.value, s.value
        assert c * c + s * s == pytest.approx(1.0, abs=1e-15)
        # applied as (x, y) <- (c x - s y, s x + c y): gives (r, 0)
        assert s * p + c * q == pytest.approx(0.0, abs=1e-15 * max(1.0, abs(p) + abs(q)))
        assert abs(c * p - s * q) == pytest.approx(np.hypot(p, q), rel=1e-14, abs=1e-300)


# --------------------------------------------------------------------------- nullspace projection (UpdaterHelper.cpp:426-454)
def test_nullspace_projection_invariants(oracle):
    rng = np.random.default_rng(1)
    rows, cols = 40, 23
    H_f = rng.normal(size=(rows, 3))
    H_x = rng.normal(size=(rows, cols))
    res = rng.normal(size=rows)
    Hf2, Hx2, r2 = oracle.nullspace_project(H_f, H_x, res)
    assert np.abs(Hf2[3:]).max() < 1e-13  # H_f rows >= 3 are annihilated
    # H' = N^T H with N an orthonormal basis of the left nullspace of H_f (any basis gives the same Gram matrices)
    Q, _ = np.linalg.qr(H_f, mode="complete")
    N = Q[:, 3:]
    A = np.hstack([H_x, res[:, None]])
    A2 = np.hstack([Hx2, r2[:, None]])
    np.testing.assert_allclose(A2.T @ A2, (N.T @ A).T @ (N.T @ A), rtol=1e-11, atol=1e-11)


# --------------------------------------------------------------------------- measurement compression (UpdaterHelper.cpp:456-487)
def test_compression_invariants(oracle):
    rng = np.random.default_rng(2)
    rows, cols = 300, 37
    H = rng.normal(size=(rows, cols))
    r = rng.normal(size=rows)
    Hc, rc = oracle.measurement_compress(H, r)
    assert Hc.shape == (cols, cols)
    assert np.abs(np.tril(Hc, -1)).max() < 1e-12
    np.testing.assert_allclose(Hc.T @ Hc, H.T @ H, rtol=1e-11, atol=1e-10)
    np.testing.assert_allclose(Hc.T @ rc, H.T @ r, rtol=1e-11, atol=1e-10)
    Q, _ = np.linalg.qr(H, mode="complete")
    assert rc @ rc == pytest.approx(r @ r - np.sum((Q[:, cols:].T @ r) ** 2), rel=1e-11)
    # fat system: returned untouched (UpdaterHelper.cpp:459-460)
    Hf, rf = oracle.measurement_compress(H[:10], r[:10])
    np.testing.assert_array_equal(Hf, H[:10])
    np.testing.assert_array_equal(rf, r[:10])


# --------------------------------------------------------------------------- EKF update (StateHelper.cpp:116-197)
def test_ekf_update_equals_information_form(oracle):
    rng = np.random.default_rng(3)
    N, D, rows = 30, 12, 50
    L = rng.normal(size=(N, N)) * 0.1 + np.eye(N)
    P = L @ L.T
    cols = np.sort(rng.choice(N, D, replace=False)).astype(np.int32)
    H = rng.normal(size=(rows, D))
    res = rng.normal(size=rows)
    sigma2 = 0.7
    st, P1, dx1 = oracle.ekf_update(P, H, res, cols, sigma2)
    assert st == 0
    Hfull = np.zeros((rows, N))
    Hfull[:, cols] = H
    Pinf = np.linalg.inv(np.linalg.inv(P) + Hfull.T @ Hfull / sigma2)
    np.testing.assert_allclose(P1, Pinf, rtol=1e-9, atol=1e-12)
    np.testing.assert_allclose(dx1, Pinf @ Hfull.T @ res / sigma2, rtol=1e-9, atol=1e-12)
    assert np.array_equal(P1, P1.T)  # exactly symmetric (Q11)
    # compressed system gives the same update (docs/update-compress.dox)
    Hc, rc = oracle.measurement_compress(H, res)
    st, P2, dx2 = oracle.ekf_update(P, Hc, rc, cols, sigma2)
    np.testing.assert_allclose(P2, P1, rtol=1e-10, atol=1e-13)
    np.testing.assert_allclose(dx2, dx1, rtol=1e-10, atol=1e-13)


# --------------------------------------------------------------------------- camera models (CamRadtan.h / CamEqui.h)
@pytest.mark.parametrize("fisheye", [0, 1])
def test_camera_jacobians_vs_finite_differences(oracle, fisheye):
    lib = oracle.load()
    cam = np.array(synth._INTRINSICS_EQUI[0] if fisheye else synth._INTRINSICS[0], dtype=np.float64)
    dist = synth.equi_distort if fisheye else synth.radtan_distort
    rng = np.random.default_rng(4)
    dp = capi.c_double_p
    for _ in range(20):
        uvn = rng.uniform(-0.5, 0.5, 2)
        uv = np.zeros(2)
        dzn = np.zeros(4)
        dze = np.zeros(16)
        lib.oracle_cam_distort(cam.ctypes.data_as(dp), fisheye, uvn.ctypes.data_as(dp), uv.ctypes.data_as(dp), dzn.ctypes.data_as(dp),
                               dze.ctypes.data_as(dp))
        # distort_d carries a float round trip (Q2): within a float ulp of the double evaluation
        ud, vd = dist(cam, uvn[0], uvn[1])
        assert abs(uv[0] - ud) < 1e-4 and abs(uv[1] - vd) < 1e-4
        assert uv[0] == np.float64(np.float32(uv[0]))  # really float-rounded
        # Jacobians are pure double (Q2): central differences of the double model
        h = 1e-6
        J = np.zeros((2, 2))
        for k in range(2):
            e = np.zeros(2)
            e[k] = h
            a = np.array(dist(cam, *(uvn + e)))
            b = np.array(dist(cam, *(uvn - e)))
            J[:, k] = (a - b) / (2 * h)
        np.testing.assert_allclose(dzn.reshape(2, 2), J, rtol=1e-6, atol=1e-5)
        Jz = np.zeros((2, 8))
        for k in range(8):
            e = np.zeros(8)
            e[k] = 1e-6 * max(1.0, abs(cam[k]))
            a = np.array(dist(cam + e, *uvn))
            b = np.array(dist(cam - e, *uvn))
            Jz[:, k] = (a - b) / (2 * e[k])
        np.testing.assert_allclose(dze.reshape(2, 8), Jz, rtol=1e-5, atol=1e-6)


# --------------------------------------------------------------------------- triangulation + GN (FeatureInitializer.cpp)
def test_triangulation_recovers_noise_free_truth(oracle):
    prob = synth.make_problem(2, F=60, pose_noise=0.0)
    # noise-free: regenerate measurements exactly from the truth with the estimated == true calibration
    prob.calib_q_p = prob.calib_q_p_true.copy()
    prob.intrinsics = np.asarray(synth._INTRINSICS[: prob.K], dtype=np.float64)
    R = [synth.quat_2_rot(q) for q in prob.clone_q_p_true[:, :4]]
    Rc = [synth.quat_2_rot(q) for q in prob.calib_q_p_true[:, :4]]
    for f in range(prob.F):
        for i in range(prob.meas_offsets[f], prob.meas_offsets[f + 1]):
            c, k = prob.clone_idx[i], prob.cam_idx[i]
            pc = Rc[k] @ (R[c] @ (prob.p_FinG_true[f] - prob.clone_q_p_true[c, 4:])) + prob.calib_q_p_true[k, 4:]
            prob.uvn[2 * i], prob.uvn[2 * i + 1] = np.float32(pc[0] / pc[2]), np.float32(pc[1] / pc[2])
    prob.clone_q_p = prob.clone_q_p_true.copy()
    v = capi.Views(prob)
    out = oracle.triangulate(capi.default_options(), v)
    ok = out["status"] == capi.FEAT_USED
    assert ok.sum() >= 0.9 * prob.F
    err = np.linalg.norm(out["p_FinG"][ok] - prob.p_FinG_true[ok], axis=1)
    assert err.max() < 2e-4  # float32 bearings at 5-7 m depth over a < 1 m baseline
    # 1d variant agrees to the same level
    out1 = oracle.triangulate(capi.default_options(triangulate_1d=1), v)
    ok1 = out1["status"] == capi.FEAT_USED
    assert ok1.sum() >= 0.9 * prob.F
    assert np.linalg.norm(out1["p_FinG"][ok1] - prob.p_FinG_true[ok1], axis=1).max() < 2e-4


def test_anchor_rule_first_group_with_most_measurements(oracle):
    prob, v = _views(2, F=30, track="ragged")
    out = oracle.triangulate(capi.default_options(), v)
    for f in range(prob.F):
        a, b = prob.meas_offsets[f], prob.meas_offsets[f + 1]
        cams = prob.cam_idx[a:b]
        groups = []
        i = 0
        while i < len(cams):
            j = i
            while j < len(cams) and cams[j] == cams[i]:
                j += 1
            groups.append((i, j))
            i = j
        best = max(groups, key=lambda g: (g[1] - g[0], -g[0]))  # most measurements, first wins ties
        assert out["anchor_meas"][f] == a + best[1] - 1  # last measurement of that camera (FeatureInitializer.cpp:46)


# --------------------------------------------------------------------------- feature Jacobians (UpdaterHelper.cpp:192-424)
def _project(prob, opts, p_FinG, i, clone_q_p, calib_q_p, intr):
    c, k = prob.clone_idx[i], prob.cam_idx[i]
    R = synth.quat_2_rot(clone_q_p[c, :4])
    Rc = synth.quat_2_rot(calib_q_p[k, :4])
    pc = Rc @ (R @ (p_FinG - clone_q_p[c, 4:])) + calib_q_p[k, 4:]
    dist = synth.equi_distort if prob.cam_is_fisheye[k] else synth.radtan_distort
    return np.array(dist(intr[k], pc[0] / pc[2], pc[1] / pc[2]))


@pytest.mark.parametrize("fisheye", [False, True])
def test_global3d_jacobian_vs_finite_differences(oracle, fisheye):
    """With FEJ off, H_x and H_f are the derivatives of the predicted pixel wrt the error states
    (left JPL perturbation for rotations: R(dtheta (+) q) ~ (I - [dtheta x]) R)."""
    prob = synth.make_problem(2, F=4, C=8, fisheye=fisheye)
    opts = capi.default_options(do_fej=0)
    v = capi.Views(prob)
    cols = oracle.column_map(opts, v)
    f = 1
    pG = prob.p_FinG_true[f] + 0.01
    H_f, H_x, res = oracle.feature_jacobian(opts, v, f, pG)
    a, b = prob.meas_offsets[f], prob.meas_offsets[f + 1]
    z0 = np.concatenate([_project(prob, opts, pG, i, prob.clone_q_p, prob.calib_q_p, prob.intrinsics) for i in range(a, b)])
    # residual = measurement - prediction (with the float round trip of distort_d, Q2)
    meas = prob.uv.reshape(-1, 2)[a:b].reshape(-1).astype(np.float64)
    np.testing.assert_allclose(res, meas - z0, atol=2e-4)
    h = 1e-6
    # feature
    for k in range(3):
        e = np.zeros(3)
        e[k] = h
        zp = np.concatenate([_project(prob, opts, pG + e, i, prob.clone_q_p, prob.calib_q_p, prob.intrinsics) for i in range(a, b)])
        zm = np.concatenate([_project(prob, opts, pG - e, i, prob.clone_q_p, prob.calib_q_p, prob.intrinsics) for i in range(a, b)])
        np.testing.assert_allclose(H_f[:, k], (zp - zm) / (2 * h), rtol=2e-5, atol=2e-4)
    # state columns: perturb the owning variable through its box-plus
    N = prob.N
    col_of = {int(c): j for j, c in enumerate(cols)}
    checked = 0
    for cov in list(prob.clone_cov_id[:3]) + [prob.calib_cov_id[0], prob.intr_cov_id[0], prob.calib_cov_id[-1], prob.intr_cov_id[-1]]:
        size = 8 if cov in prob.intr_cov_id else 6
        for k in range(size):
            def pred(sign):
                dx = np.zeros(N)
                dx[cov + k] = sign * h
                cl = np.array([synth.boxplus_pose(prob.clone_q_p[c], dx[prob.clone_cov_id[c]: prob.clone_cov_id[c] + 6]) for c in range(prob.C)])
                ca = np.array([synth.boxplus_pose(prob.calib_q_p[kk], dx[prob.calib_cov_id[kk]: prob.calib_cov_id[kk] + 6]) for kk in range(prob.K)])
                it = np.array([prob.intrinsics[kk] + dx[prob.intr_cov_id[kk]: prob.intr_cov_id[kk] + 8] for kk in range(prob.K)])
                return np.concatenate([_project(prob, opts, pG, i, cl, ca, it) for i in range(a, b)])
            fd = (pred(+1) - pred(-1)) / (2 * h)
            np.testing.assert_allclose(H_x[:, col_of[cov + k]], fd, rtol=5e-5, atol=5e-4)
            checked += 1
    assert checked > 30


@pytest.mark.parametrize("rep", [capi.REP_GLOBAL_FULL_INVERSE_DEPTH, capi.REP_ANCHORED_3D, capi.REP_ANCHORED_FULL_INVERSE_DEPTH,
                                 capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH])
def test_representation_jacobians_give_same_projected_system(oracle, rep):
    """Every 3-dof representation spans the same feature subspace, so after the nullspace projection the
    information H'^T H', H'^T r' must equal the GLOBAL_3D one (FEJ off so that all are linearised at one point)."""
    prob = synth.make_problem(2, F=6, C=10)
    v = capi.Views(prob)
    o0 = capi.default_options(do_fej=0)
    tri = oracle.triangulate(o0, v)
    f = int(np.nonzero(tri["status"] == 0)[0][0])
    H_f0, H_x0, r0 = oracle.feature_jacobian(o0, v, f, tri["p_FinG"][f], tri["p_FinA"][f], tri["anchor_meas"][f])
    _, Hp0, rp0 = oracle.nullspace_project(H_f0, H_x0, r0)
    o1 = capi.default_options(do_fej=0, feat_rep_msckf=rep)
    H_f1, H_x1, r1 = oracle.feature_jacobian(o1, v, f, tri["p_FinG"][f], tri["p_FinA"][f], tri["anchor_meas"][f])
    _, Hp1, rp1 = oracle.nullspace_project(H_f1, H_x1, r1)
    np.testing.assert_allclose(r1, r0, atol=1e-9)
    np.testing.assert_allclose(Hp1.T @ Hp1, Hp0.T @ Hp0, rtol=1e-7, atol=1e-6)
    np.testing.assert_allclose(Hp1.T @ rp1, Hp0.T @ rp0, rtol=1e-7, atol=1e-6)


# --------------------------------------------------------------------------- whole update
def test_update_posterior_is_consistent(oracle):
    prob, v = _views(2, F=60)
    opts = capi.default_options(chi2_multipler=1.0)
    out = oracle.msckf_update(opts, v, want_compressed=True)
    assert out["stats"]["status"] == 0 and out["stats"]["n_used"] > 40
    P1 = out["P"]
    assert np.array_equal(P1, P1.T)
    w = np.linalg.eigvalsh(P1)
    assert w.min() > -1e-12
    assert np.all(np.diag(P1) <= np.diag(prob.P) + 1e-15)  # information can only shrink the marginals
    # the compressed system reproduces the update through the stand-alone EKF step
    cols = oracle.column_map(opts, v)
    st, P2, dx2 = oracle.ekf_update(prob.P, out["H_comp"], out["r_comp"], cols, 1.0)
    np.testing.assert_allclose(P2, P1, rtol=0, atol=1e-15)
    np.testing.assert_allclose(dx2, out["dx"], rtol=0, atol=1e-15)
    # box-plus keeps unit quaternions with q4 >= 0
    q = out["clone_q_p"][:, :4]
    np.testing.assert_allclose(np.linalg.norm(q, axis=1), 1.0, atol=1e-14)
    assert np.all(q[:, 3] >= 0)


def test_outliers_are_gated(oracle):
    prob, v = _views(2, F=80, outlier_frac=0.25)
    out = oracle.msckf_update(capi.default_options(chi2_multipler=1.0), v)
    n_rej = int(np.sum(out["feat_status"] == capi.FEAT_CHI2_REJECTED))
    assert 5 <= n_rej <= 40


def test_too_few_measurements_are_dropped(oracle):
    prob = synth.make_problem(2, F=10)
    # truncate feature 3 to a single measurement, feature 5 to none
    keep = []
    offs = [0]
    for f in range(prob.F):
        a, b = int(prob.meas_offsets[f]), int(prob.meas_offsets[f + 1])
        if f == 3:
            b = a + 1
        if f == 5:
            b = a
        keep += list(range(a, b))
        offs.append(offs[-1] + (b - a))
    keep = np.asarray(keep)
    prob.meas_offsets = np.asarray(offs, dtype=np.int32)
    prob.uv = prob.uv.reshape(-1, 2)[keep].reshape(-1)
    prob.uvn = prob.uvn.reshape(-1, 2)[keep].reshape(-1)
    prob.clone_idx = prob.clone_idx[keep]
    prob.cam_idx = prob.cam_idx[keep]
    out = oracle.msckf_update(capi.default_options(), capi.Views(prob))
    assert out["feat_status"][3] == capi.FEAT_TOO_FEW_MEAS and out["feat_status"][5] == capi.FEAT_TOO_FEW_MEAS
    assert out["stats"]["status"] == 0


# --------------------------------------------------------------------------- UpdaterSLAM::update (oracle_slam_update)
def test_slam_update_equals_information_form():
    """The SLAM update stacks [H_x | H_f] of landmarks that live in the state and applies one EKF update: the posterior
    must equal (P^-1 + H^T H / sigma^2)^-1 built from the very stack the oracle returns, the gate threshold is the 0.95
    chi-square quantile of 2m dof (UpdaterSLAM.cpp:399-405), and good landmarks move towards the truth."""
    from scipy import stats as sps
    from open_vins_amd import capi, synth
    from oracle import pyoracle
    prob = synth.make_slam_problem(2, L=10)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    o = pyoracle.slam_update(opts, v, want_stack=True)
    assert o["stats"]["status"] == 0 and o["D"] == 6 * prob.C + 14 * prob.K + 3 * 10
    m = np.diff(prob.meas_offsets)
    np.testing.assert_allclose(o["chi2_thresh"], sps.chi2.ppf(0.95, 2 * m), rtol=1e-10)
    used = o["feat_status"] == capi.FEAT_USED
    assert used.sum() >= 8 and o["rows"] == int((2 * m[used]).sum())
    H = np.zeros((o["rows"], prob.N))
    H[:, o["col_cov_id"]] = o["H"]
    Pinf = np.linalg.inv(np.linalg.inv(prob.P) + H.T @ H)
    assert np.linalg.norm(o["P"] - Pinf) / np.linalg.norm(Pinf) < 1e-9
    assert np.linalg.norm(o["dx"] - Pinf @ H.T @ o["r"]) / np.linalg.norm(o["dx"]) < 1e-6
    np.testing.assert_allclose(o["landmarks"], prob.lm_value + o["dx"][prob.lm_cov_id[:, None] + np.arange(3)], rtol=0, atol=1e-15)
    assert np.abs(o["landmarks"] - prob.p_FinG_true)[used].mean() < np.abs(prob.lm_value - prob.p_FinG_true)[used].mean()


def test_slam_update_with_per_feature_noise_equals_weighted_information_form():
    """UpdaterSLAM.cpp:392-409, :444: with per-feature sigmas the posterior is (P^-1 + H^T R^-1 H)^-1 with the diagonal R_big of
    the accepted features; a constant per-feature sigma is the same as the global option."""
    from open_vins_amd import capi, synth
    from oracle import pyoracle
    prob = synth.make_slam_problem(2, L=10)
    v = capi.Views(prob)
    F = v.features.F
    same = pyoracle.slam_update(capi.default_options(chi2_multipler=1.0, sigma_pix=2.0), v)
    const = pyoracle.slam_update(capi.default_options(chi2_multipler=1.0, sigma_pix=1.0), v, feat_sigma=np.full(F, 2.0))
    assert np.array_equal(same["feat_status"], const["feat_status"])
    np.testing.assert_allclose(const["P"], same["P"], rtol=1e-12, atol=1e-18)
    sig = np.where(np.arange(F) % 3 == 0, 2.0, 1.0)
    mult = np.where(np.arange(F) % 3 == 0, 1e6, 1.0)
    opts = capi.default_options(chi2_multipler=1.0)
    o = pyoracle.slam_update(opts, v, want_stack=True, feat_sigma=sig, feat_chi2mult=mult)
    assert np.all(o["feat_status"][::3] == capi.FEAT_USED)  # the wide gate lets every third feature through
    m = np.diff(prob.meas_offsets)
    used = o["feat_status"] == capi.FEAT_USED
    w = np.concatenate([np.full(2 * m[f], 1.0 / sig[f] ** 2) for f in range(F) if used[f]])
    H = np.zeros((o["rows"], prob.N))
    H[:, o["col_cov_id"]] = o["H"]
    Pinf = np.linalg.inv(np.linalg.inv(prob.P) + H.T @ (w[:, None] * H))
    assert np.linalg.norm(o["P"] - Pinf) / np.linalg.norm(Pinf) < 1e-9
    assert np.linalg.norm(o["dx"] - Pinf @ H.T @ (w * o["r"])) / np.linalg.norm(o["dx"]) < 1e-6


def test_prior_whitened_information_form_equals_the_kalman_form():
    """The algebra the device's default route relies on (csrc/k_ekf.h, "EKF update from the Gram matrix"): with G = H^T H,
    g = H^T r, P_DD = U1^T U1, B = U1^-T P(D,:), T = I + U1 G U1^T / s^2 = C^T C, Y2 = C^-T B:
    P' = P - (B^T B - Y2^T Y2) and dx = Y2^T C^-T (U1 g / s^2) reproduce StateHelper::EKFUpdate — also when H is rank deficient
    (the MSCKF stack always is) and when nothing is measured (G = 0 gives P' = P, dx = 0)."""
    from scipy.linalg import cholesky, solve_triangular
    from open_vins_amd import capi, synth
    from oracle import pyoracle
    prob = synth.make_problem(2, F=48, C=10)
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    o = pyoracle.msckf_update(opts, v, want_compressed=True)
    assert o["stats"]["n_used"] > 20
    cols = pyoracle.column_map(opts, v)
    H, r = o["H_comp"], o["r_comp"]          # R^T R = H^T H of the full stack, R^T z = H^T r
    assert np.linalg.matrix_rank(H) < H.shape[1]  # gauge directions: the stack is rank deficient
    s2 = opts.sigma_pix ** 2

    def whitened(G, g):
        P = prob.P
        U1 = cholesky(P[np.ix_(cols, cols)])                    # upper, U1^T U1 = P_DD
        B = solve_triangular(U1, P[cols, :], trans="T")
        T = np.eye(len(cols)) + U1 @ G @ U1.T / s2
        C = cholesky(T)
        Y2 = solve_triangular(C, B, trans="T")
        y2 = solve_triangular(C, U1 @ g / s2, trans="T")
        return P - (B.T @ B - Y2.T @ Y2), Y2.T @ y2, np.linalg.eigvalsh(T)

    P1, dx1, ev = whitened(H.T @ H, H.T @ r)
    assert ev.min() > 1.0 - 1e-9                                # eigenvalues of T are >= 1: no small pivots
    assert np.linalg.norm(P1 - o["P"]) / np.linalg.norm(o["P"]) < 1e-11
    assert np.linalg.norm(dx1 - o["dx"]) / np.linalg.norm(o["dx"]) < 1e-9
    P0, dx0, _ = whitened(np.zeros((len(cols), len(cols))), np.zeros(len(cols)))
    assert np.abs(P0 - prob.P).max() < 1e-15 and not dx0.any()


def test_slam_gate_rejects_a_displaced_landmark():
    from open_vins_amd import capi, synth
    from oracle import pyoracle
    prob = synth.make_slam_problem(2, L=8)
    prob.lm_value[3] += np.array([0.8, -0.6, 0.5])  # far outside its 0.1 m prior
    o = pyoracle.slam_update(capi.default_options(chi2_multipler=1.0), capi.Views(prob))
    assert o["feat_status"][3] == capi.FEAT_CHI2_REJECTED and o["chi2"][3] > o["chi2_thresh"][3]


@pytest.mark.parametrize("rep", [capi.REP_GLOBAL_FULL_INVERSE_DEPTH, capi.REP_ANCHORED_3D, capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH])
def test_slam_update_in_other_representations(rep):
    """Landmarks stored in another representation (Landmark::value() coordinates): the stack the oracle returns must
    reproduce the posterior in information form, and the landmark columns must be the GLOBAL_3D ones chained with
    d xyz / d lambda — checked through finite differences of get_xyz on the residual."""
    from oracle import pyoracle
    prob = synth.make_slam_problem(2, L=8, lm_rep=rep, C=16)
    opts = capi.default_options(chi2_multipler=5.0)
    v = capi.Views(prob)
    o = pyoracle.slam_update(opts, v, want_stack=True)
    assert o["stats"]["status"] == 0
    used = o["feat_status"] == capi.FEAT_USED
    assert used.sum() >= 6
    H = np.zeros((o["rows"], prob.N))
    H[:, o["col_cov_id"]] = o["H"]
    Pinf = np.linalg.inv(np.linalg.inv(prob.P) + H.T @ H)
    assert np.linalg.norm(o["P"] - Pinf) / np.linalg.norm(Pinf) < 1e-8
    # residual(lambda + e) - residual(lambda) ~ -H_lambda e   (do_fej = 0 so the Jacobian is evaluated at the estimate)
    opts0 = capi.default_options(chi2_multipler=1e9, do_fej=0)
    base = pyoracle.slam_update(opts0, v, want_stack=True)
    f = 2
    rows = slice(int(2 * prob.meas_offsets[f]), int(2 * prob.meas_offsets[f + 1]))
    cols = [int(np.where(base["col_cov_id"] == prob.lm_cov_id[f] + i)[0][0]) for i in range(3)]
    for i in range(3):
        # central differences; the residual goes through a float32 round trip (CamBase::distort_d), hence the large step
        eps = 1e-3 * max(abs(prob.lm_value[f, i]), 0.05)
        rr = []
        for sgn in (+1, -1):
            p2 = synth.make_slam_problem(2, L=8, lm_rep=rep, C=16)
            p2.lm_value[f, i] += sgn * eps
            rr.append(pyoracle.slam_update(opts0, capi.Views(p2), want_stack=True)["r"][rows])
        fd = -(rr[0] - rr[1]) / (2 * eps)
        assert np.abs(fd - base["H"][rows, cols[i]]).max() < 5e-3 * max(1.0, np.abs(fd).max())


def _joint_information_form(prob, opts, f, pG, pA, anchor):
    """Posterior of [state ; landmark] from the prior on the state, NO prior on the landmark and all 2m rows of feature f."""
    from oracle import pyoracle
    v = capi.Views(prob)
    H_f, H_x, res = pyoracle.feature_jacobian(opts, v, f, pG, pA, anchor)
    cols = np.zeros(H_x.shape[1], np.int32)
    D = pyoracle.load().oracle_column_map(C.byref(opts), C.byref(v.state), cols.ctypes.data_as(C.POINTER(C.c_int32)))
    N = prob.N
    J = np.zeros((H_x.shape[0], N + 3))
    J[:, cols[:D]] = H_x
    J[:, N:] = H_f
    info = J.T @ J / opts.sigma_pix ** 2
    info[:N, :N] += np.linalg.inv(prob.P)
    Pj = np.linalg.inv(info)
    return Pj, Pj @ J.T @ res / opts.sigma_pix ** 2


@pytest.mark.parametrize("rep", [capi.REP_GLOBAL_3D, capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH])
def test_delayed_init_of_one_feature_equals_joint_least_squares(rep):
    """StateHelper::initialize = Givens split + invertible initialisation + EKF update with the remaining rows.  For one
    feature that is, exactly, the linear least-squares posterior of [state; landmark] with a flat prior on the landmark."""
    from oracle import pyoracle
    prob = synth.make_problem(2, F=1)
    opts = capi.default_options(chi2_multipler=1e6, feat_rep_msckf=rep)
    v = capi.Views(prob)
    tri = pyoracle.triangulate(opts, v)
    assert tri["status"][0] == capi.FEAT_USED
    o = pyoracle.slam_delayed_init(opts, v, feat_rep=rep, tri=tri)
    assert o["rc"] == 0 and o["N"] == prob.N + 3 and o["lm_cov_id"][0] == prob.N
    Pj, dj = _joint_information_form(prob, opts, 0, tri["p_FinG"][0], tri["p_FinA"][0], int(tri["anchor_meas"][0]))
    assert np.linalg.norm(o["P"] - Pj) / np.linalg.norm(Pj) < 1e-7
    np.testing.assert_allclose(o["P"], o["P"].T, atol=1e-12 * np.abs(o["P"]).max())
    dx = o["dx_seq"][0, : prob.N + 3]
    total = np.concatenate([dx[: prob.N], o["lm_value"][0] - o["lm_fej"][0]])
    assert np.linalg.norm(total - dj) / np.linalg.norm(dj) < 1e-6
    xyz = tri["p_FinA"][0] if rep >= 2 else tri["p_FinG"][0]
    np.testing.assert_allclose(o["lm_fej"][0], synth.landmark_from_xyz(rep, xyz), rtol=1e-13)


def test_delayed_init_chain_bookkeeping():
    """Several features, some gated out: the covariance grows by 3 per accepted feature in feature order, rejected ones
    leave no trace (zero dx row, no id), the posterior stays symmetric positive definite and every accepted landmark ends
    close to the truth; an already-resident landmark receives the corrections through its cross-covariance."""
    from oracle import pyoracle
    prob = synth.make_slam_problem(2, L=3, seed=5)
    prob2 = synth.make_problem(2, F=12, seed=5, outlier_frac=0.25)
    # new tracks on the state that already holds 3 landmarks
    for k in ("meas_offsets", "uv", "uvn", "clone_idx", "cam_idx", "p_FinG_true"):
        setattr(prob, k, getattr(prob2, k))
    opts = capi.default_options(chi2_multipler=1.0)
    v = capi.Views(prob)
    o = pyoracle.slam_delayed_init(opts, v, feat_rep=0)
    acc = o["feat_status"] == capi.FEAT_USED
    assert 3 <= acc.sum() < 12 and (o["feat_status"] == capi.FEAT_CHI2_REJECTED).sum() >= 1
    assert o["N"] == prob.N + 3 * acc.sum()
    np.testing.assert_array_equal(o["lm_cov_id"][acc], prob.N + 3 * np.arange(acc.sum()))
    assert (o["lm_cov_id"][~acc] == -1).all() and not o["dx_seq"][~acc].any()
    assert np.linalg.eigvalsh(0.5 * (o["P"] + o["P"].T)).min() > 0
    err = np.linalg.norm(o["lm_value"][acc] - prob2.p_FinG_true[acc], axis=1)
    assert (err < 0.25 * np.linalg.norm(prob2.p_FinG_true[acc], axis=1)).all()  # depth from a short baseline
    assert np.abs(o["landmarks_existing"] - prob.lm_value).max() > 0  # moved by the updates
    m = np.diff(prob.meas_offsets)
    from scipy import stats as sps
    gated = np.isfinite(o["chi2"])
    np.testing.assert_allclose(o["chi2_thresh"][gated], sps.chi2.ppf(0.95, 2 * m[gated]), rtol=1e-10)


def test_window_bookkeeping_against_dense_algebra():
    """StateHelper::marginalize / clone (+ time-offset Jacobian) / EKFPropagation restated in the oracle vs J P J^T forms."""
    from oracle import pyoracle
    rng = np.random.default_rng(0)
    A = rng.normal(size=(20, 20))
    P = A @ A.T
    keep = np.r_[0:5, 11:20]
    np.testing.assert_array_equal(pyoracle.marginalize(P, 5, 6), P[np.ix_(keep, keep)])
    d = rng.normal(size=6)
    J = np.zeros((26, 20))
    J[:20] = np.eye(20)
    J[20:, 3:9] = np.eye(6)
    np.testing.assert_array_equal(pyoracle.augment_clone(P, 3, 6), J @ P @ J.T)  # pure copies
    J[20:, 15] = d
    np.testing.assert_allclose(pyoracle.augment_clone(P, 3, 6, dt_id=15, dnc_dt=d), J @ P @ J.T, rtol=1e-13, atol=1e-13)
    Phi, Q, ids = rng.normal(size=(6, 9)), np.diag(rng.uniform(0.1, 1, 6)), np.r_[2:8, 12:15]
    rc, Pp = pyoracle.propagate(P, 2, ids, Phi, Q)
    F = np.eye(20)
    F[2:8, :] = 0
    F[2:8, ids] = Phi
    Qf = np.zeros((20, 20))
    Qf[2:8, 2:8] = Q
    assert rc == 0
    np.testing.assert_allclose(Pp, F @ P @ F.T + Qf, rtol=1e-12, atol=1e-12)
    rc, _ = pyoracle.propagate(P, 2, ids, Phi, -1e6 * np.eye(6))  # StateHelper.cpp:101-113
    assert rc == capi.ERR_NEGATIVE_DIAGONAL


@pytest.mark.parametrize("rep", [capi.REP_ANCHORED_3D, capi.REP_ANCHORED_FULL_INVERSE_DEPTH, capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH])
def test_anchor_change_keeps_the_landmark_where_it_is(rep):
    """UpdaterSLAM::perform_anchor_change: the landmark's global position (value and first estimate) does not move, only
    the landmark's rows / columns of P change, P stays symmetric positive definite."""
    from oracle import pyoracle
    prob = synth.make_slam_problem(2, L=6, lm_rep=rep)
    opts = capi.default_options()
    v = capi.Views(prob)
    l, new_clone = 2, prob.C - 1
    cam = int(prob.lm_anchor_cam[l])
    o = pyoracle.anchor_change(opts, v, l, cam, new_clone)
    assert o["rc"] == 0

    def to_global(val, clone, q_p):
        qc, qk = q_p[clone], prob.calib_q_p[cam]
        R_GtoI, R_ItoC = synth.quat_2_rot(qc[:4]), synth.quat_2_rot(qk[:4])
        return R_GtoI.T @ (R_ItoC.T @ (synth.landmark_to_xyz(rep, val) - qk[4:7])) + qc[4:7]

    old_clone = int(prob.lm_anchor_clone[l])
    np.testing.assert_allclose(to_global(o["value"], new_clone, prob.clone_q_p), to_global(prob.lm_value[l], old_clone, prob.clone_q_p), rtol=0, atol=1e-11)
    # the first estimate: carried through the FEJ poses — except that Landmark::get_xyz(true) reads the CURRENT value for the MSCKF
    # inverse depth (Landmark.cpp:47-59 ignores its flag there; found with oracle/_ref in round 4), so that representation's new
    # first estimate is its current value seen through the FEJ poses
    src_fej = prob.lm_value[l] if rep == capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH else prob.lm_fej[l]
    np.testing.assert_allclose(to_global(o["fej"], new_clone, prob.clone_q_p_fej), to_global(src_fej, old_clone, prob.clone_q_p_fej), rtol=0, atol=1e-11)
    idx = int(prob.lm_cov_id[l]) + np.arange(3)
    rest = np.setdiff1d(np.arange(prob.N), idx)
    np.testing.assert_array_equal(o["P"][np.ix_(rest, rest)], prob.P[np.ix_(rest, rest)])
    assert np.abs(o["P"][np.ix_(idx, idx)] - prob.P[np.ix_(idx, idx)]).max() > 0
    assert np.linalg.eigvalsh(0.5 * (o["P"] + o["P"].T)).min() > 0


def test_single_depth_landmark_is_the_marginal_of_the_inverse_depth_one():
    """ANCHORED_INVERSE_DEPTH_SINGLE projects the bearing out of every system (UpdaterSLAM.cpp:181-196, :371-379) = a flat
    prior on the two bearing coordinates of ANCHORED_MSCKF_INVERSE_DEPTH, marginalised.  Delayed initialisation of one
    feature must therefore give the 3-dof result restricted to (state, rho); the SLAM update must reproduce its own stack in
    information form with 2m - 2 rows per landmark and the 0.95 quantile of 2m - 2 dof as threshold."""
    from scipy import stats as sps
    from oracle import pyoracle
    p1 = synth.make_problem(2, F=1)
    o1 = capi.default_options(chi2_multipler=1e6)
    v1 = capi.Views(p1)
    tri = pyoracle.triangulate(o1, v1)
    d5 = pyoracle.slam_delayed_init(o1, v1, feat_rep=capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE, tri=tri)
    d4 = pyoracle.slam_delayed_init(o1, v1, feat_rep=capi.REP_ANCHORED_MSCKF_INVERSE_DEPTH, tri=tri)
    N = p1.N
    keep = np.r_[0:N, N + 2]
    assert d5["N"] == N + 1 and d4["N"] == N + 3
    np.testing.assert_allclose(d5["P"], d4["P"][np.ix_(keep, keep)], rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(d5["lm_value"][:, 2], d4["lm_value"][:, 2], rtol=1e-12)  # rho; the bearing stays what the triangulation gave
    np.testing.assert_array_equal(d5["lm_value"][:, :2], d5["lm_fej"][:, :2])
    np.testing.assert_allclose(d5["dx_seq"][0, :N], d4["dx_seq"][0, :N], rtol=1e-10, atol=1e-14)
    m = int(np.diff(p1.meas_offsets)[0])
    np.testing.assert_allclose(d5["chi2_thresh"][0] / 1e6, sps.chi2.ppf(0.95, 2 * m - 2), rtol=1e-10)
    prob = synth.make_slam_problem(2, L=6, lm_rep=capi.REP_ANCHORED_INVERSE_DEPTH_SINGLE)
    opts = capi.default_options(chi2_multipler=5.0)
    o = pyoracle.slam_update(opts, capi.Views(prob), want_stack=True)
    used = o["feat_status"] == capi.FEAT_USED
    mm = np.diff(prob.meas_offsets)
    assert used.sum() >= 4 and o["rows"] == int((2 * mm[used] - 2).sum()) and o["D"] == 6 * prob.C + 14 * prob.K + 6
    np.testing.assert_allclose(o["chi2_thresh"][used], 5.0 * sps.chi2.ppf(0.95, 2 * mm[used] - 2), rtol=1e-10)
    H = np.zeros((o["rows"], prob.N))
    H[:, o["col_cov_id"]] = o["H"]
    Pinf = np.linalg.inv(np.linalg.inv(prob.P) + H.T @ H)
    assert np.linalg.norm(o["P"] - Pinf) / np.linalg.norm(Pinf) < 1e-9
    np.testing.assert_array_equal(o["landmarks"][:, :2], prob.lm_value[:, :2])  # the bearing is a constant of the landmark


# ---------------------------------------------------------------------------------------------------
# retriangulation of the active tracks (oracle/retri_oracle.py, VioManagerHelper.cpp:190-387)
# ---------------------------------------------------------------------------------------------------
def _run_retri(scene, **kw):
    from oracle import retri_oracle
    at = retri_oracle.ActiveTracks(**kw)
    hist = []
    for t in range(scene.T):
        pos, uvd = at.frame(scene.R_GtoI[t], scene.p_IinG[t], scene.cams(), scene.obs[t], 752, 480)
        hist.append((dict(pos), dict(uvd), dict(at.count)))
    return hist


def test_retriangulation_recovers_noise_free_truth():
    from tests.retri_scene import Scene
    sc = Scene(T=9, n_pts=30, noise=0.0, p_see=(1.0, 1.0))
    hist = _run_retri(sc)
    assert not hist[2][0]                                     # fewer than four observations: nothing yet (:275)
    pos, uvd, count = hist[-1]
    assert len(pos) >= 25 and max(count.values()) == sc.T      # one observation per FRAME is kept for a known track (:268-272)
    for fid, p in pos.items():
        assert np.abs(p - sc.pts[fid - 100]).max() < 1e-5      # float32 normalised coordinates
    for fid, d in uvd.items():
        pc = sc.R_ItoC[0] @ (sc.R_GtoI[-1] @ (pos[fid] - sc.p_IinG[-1])) + sc.p_IinC[0]
        assert abs(d[2] - pc[2]) < 1e-12 and 0 <= d[0] < 752 and 0 <= d[1] < 480


def test_retriangulation_drops_tracks_that_miss_a_frame():
    from tests.retri_scene import Scene
    sc = Scene(T=8, n_pts=20, noise=0.0, p_see=(1.0, 1.0))
    fid = sc.obs[5][0][0][0]
    for k in (0, 1):
        sc.obs[5][k] = [o for o in sc.obs[5][k] if o[0] != fid]  # the track is lost in frame 5 and re-detected in frame 6
    hist = _run_retri(sc)
    assert fid in hist[4][0] and fid not in hist[5][2]
    assert hist[6][2][fid] == 1 and hist[7][2][fid] == 2 and fid not in hist[7][0]


def test_retriangulation_multi_camera_bookkeeping():
    """:264-272: a known track adds each camera's observation to the OLD system (the last camera's survives), a new track keeps
    the first camera's."""
    from oracle import retri_oracle
    from tests.retri_scene import Scene
    sc = Scene(T=3, n_pts=5, noise=0.0, p_see=(1.0, 1.0))
    at = retri_oracle.ActiveTracks()
    at.frame(sc.R_GtoI[0], sc.p_IinG[0], sc.cams(), sc.obs[0], 752, 480)
    fid, _, pn0 = sc.obs[0][0][0]

    def Ai(t, k, pn):
        R_GtoC = sc.R_ItoC[k] @ sc.R_GtoI[t]
        b = R_GtoC.T @ np.array([float(pn[0]), float(pn[1]), 1.0])
        B = retri_oracle.skew_x(b / np.linalg.norm(b))
        return B.T @ B
    np.testing.assert_allclose(at.A[fid], Ai(0, 0, pn0), atol=1e-15)    # new track: camera 0's only
    A_old = at.A[fid].copy()
    at.frame(sc.R_GtoI[1], sc.p_IinG[1], sc.cams(), sc.obs[1], 752, 480)
    pn1 = [o for o in sc.obs[1][1] if o[0] == fid][0][2]
    np.testing.assert_allclose(at.A[fid], A_old + Ai(1, 1, pn1), atol=1e-15)  # known track: old + camera 1's
    assert at.count[fid] == 2


# --------------------------------------------------------------------------- the forced gate verdicts the parity tests fall back on
def test_forced_gate_verdicts(oracle):
    """ORACLE_FORCE_ACCEPT / _REJECT (ov_oracle.h): the verdict of a borderline feature can be imposed; chi2 is still reported."""
    prob, v = _views(2, F=40, outlier_frac=0.3)
    opts = capi.default_options(chi2_multipler=1.0)
    tri = oracle.triangulate(opts, v)
    ref = oracle.msckf_update(opts, v, given=tri)
    used = np.flatnonzero(ref["feat_status"] == capi.FEAT_USED)
    rej = np.flatnonzero(ref["feat_status"] == capi.FEAT_CHI2_REJECTED)
    assert len(used) > 2 and len(rej) > 0
    st = np.array(tri["status"], dtype=np.int32)
    st[used[0]], st[rej[0]] = -2, -1
    out = oracle.msckf_update(opts, v, given=dict(tri, status=st))
    assert out["feat_status"][used[0]] == capi.FEAT_CHI2_REJECTED and out["feat_status"][rej[0]] == capi.FEAT_USED
    assert out["stats"]["n_used"] == ref["stats"]["n_used"]
    np.testing.assert_array_equal(out["chi2"], ref["chi2"])
    assert not np.allclose(out["dx"], ref["dx"])
    # untouched verdicts reproduce the plain run bit for bit
    out2 = oracle.msckf_update(opts, v, given=dict(tri, status=np.array(tri["status"], dtype=np.int32)))
    np.testing.assert_array_equal(out2["dx"], ref["dx"])


def test_tight_window_snapshot_is_consistent():
    """synth.tight_window_problem (bench.py's `tight_window`, the GPU test of the gate's residual bound): errors and prior scaled alike,
    so the gate accepts as it does on the unscaled snapshot — and the accepted statistics sit where chi2(2m - 3) puts them (mean ~ dof),
    which is what lets the bound |r'|^2 / sigma^2 decide most features there."""
    from oracle import pyoracle
    opts = capi.default_options(chi2_multipler=1.0)
    frac, ratio = {}, {}
    for name, prob in (("survey", synth.make_problem(3, F=200)), ("tight", synth.tight_window_problem(3, 0.05, F=200))):
        o = pyoracle.msckf_update(opts, capi.Views(prob))
        reached = np.isfinite(o["chi2"])
        used = o["feat_status"] == capi.FEAT_USED
        assert reached.sum() > 150
        frac[name] = used.sum() / reached.sum()
        dof = 2 * np.diff(prob.meas_offsets)[used] - 3
        ratio[name] = float(np.mean(o["chi2"][used] / dof))
    assert frac["tight"] > 0.9 and frac["survey"] > 0.9
    assert 0.7 < ratio["tight"] < 1.2 and 0.7 < ratio["survey"] < 1.2, ratio
    # the tight window really is tighter: its prior's clone block is 0.05^2 of the survey snapshot's
    a, b = synth.make_problem(3, F=10), synth.tight_window_problem(3, 0.05, F=10)
    i = int(a.clone_cov_id[0])
    np.testing.assert_allclose(b.P[i:i + 6, i:i + 6], 0.0025 * a.P[i:i + 6, i:i + 6], rtol=1e-12)
