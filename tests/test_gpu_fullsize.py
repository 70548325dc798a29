"""GPU parity at BASELINE.json's full sizes, against the CPU ORACLE (never GPU against GPU), the prior-conditioning
sweep of the default (Gram / prior-whitened) route, and the camera models called directly.

  * configs[2] (2000 features), one rank's shard of configs[3] (4 cameras, 1250 features), north_star's 10 000-feature
    batch on the 30-clone state, configs[4] geometry (50 clones, 4 cameras, D = 356) at F >= 200: accept sets, chi2, dx
    (rel 1e-8), P (rel-Frobenius 1e-9) with the oracle's triangulation injected on both sides.
  * conditioning: priors built like a live window (synth.realistic_prior), cond(P_DD) 1e4 ... 3e16, each compared with
    an extended-precision (numpy longdouble, 64-bit mantissa) evaluation of the same update.  The oracle — the reference's
    own S = H P H^T + R form — loses accuracy with the conditioning too; the GPU must stay within 1e-9 (P) / 1e-8 (dx) or
    within a small factor of the oracle's own error, whichever is larger.
"""
import numpy as np
import pytest

from open_vins_amd import capi, synth
from parity_util import assert_chi2, oracle_with_the_same_gate_verdicts

pytestmark = pytest.mark.gpu
LD = np.longdouble


@pytest.fixture(scope="module")
def Updater():
    import torch
    assert torch.cuda.is_available(), "these tests need a GPU"
    from open_vins_amd.updater import UpdaterMSCKF
    return UpdaterMSCKF


def _rel(a, b):
    return float(np.linalg.norm(np.asarray(a - b, dtype=np.float64)) / max(np.linalg.norm(np.asarray(b, dtype=np.float64)), 1e-300))


_ORACLE_RUNS = {}  # key -> (triangulation, update) of the oracle: the large batches are used by more than one test


def _parity(Updater, oracle, prob, opts, tol_dx=1e-8, tol_p=1e-9, key=None, **fields):
    v = capi.Views(prob)
    if key is not None and key in _ORACLE_RUNS:
        tri, ref = _ORACLE_RUNS[key]
    else:
        tri = oracle.triangulate(opts, v)
        ref = oracle.msckf_update(opts, v, given=tri)
        if key is not None:
            _ORACLE_RUNS[key] = (tri, ref)
    up = Updater(opts)
    up.set_problem(prob)
    up.set_triangulation(tri["p_FinG"], tri["p_FinA"], tri["anchor_meas"], tri["status"])
    out = up.update()
    out["stack_f32"] = up.debug_option("stack_is_f32")
    up.close()
    ref = oracle_with_the_same_gate_verdicts(oracle, opts, v, tri, ref, out)
    assert_chi2(out, ref, 1e-8, strict=bool(opts.gate_always_factor))
    assert ref["stats"]["n_used"] > 0.8 * prob.F
    assert np.array_equal(out["feat_status"], ref["feat_status"])
    assert _rel(out["dx"], ref["dx"]) < tol_dx, _rel(out["dx"], ref["dx"])
    assert _rel(out["P"], ref["P"]) < tol_p, _rel(out["P"], ref["P"])
    assert np.array_equal(out["P"], out["P"].T)
    # the posterior clone table: dx's tolerance times the size of the correction (|dx| ~ 0.1 - 0.3 on these snapshots)
    assert np.abs(out["clone_q_p"] - ref["clone_q_p"]).max() < (1e-9 if tol_dx <= 1e-8 else 0.3 * tol_dx)
    return out, ref


def _parity_device_triangulation(Updater, oracle, prob, opts, tol_dx=1e-8, tol_p=1e-9, key=None):
    """END TO END at full size with the DEVICE's triangulation (no set_triangulation): triangulate -> per-feature systems -> gate ->
    compression -> update in one ovgpu_msckf_update.  Two comparisons against the oracle:
      1. the triangulation itself (verdicts identical, positions to 1e-6 m: FeatureInitializer's float32 cost path decides the
         Levenberg-Marquardt steps, so a feature near a step boundary may end one iteration apart; tests/test_gpu_parity.py holds the
         tight bound on batches without such features);
      2. everything downstream against the oracle run ON THE DEVICE'S positions (accept sets, chi2, dx, P') at the full-size tolerances."""
    v = capi.Views(prob)
    if key is not None and key in _ORACLE_RUNS:
        tri, _ = _ORACLE_RUNS[key]
    else:
        tri = oracle.triangulate(opts, v)
    up = Updater(opts)
    up.set_problem(prob)
    out = up.update()
    got = up.get_triangulation()
    up.close()
    tri_failed = (out["feat_status"] != capi.FEAT_USED) & (out["feat_status"] != capi.FEAT_CHI2_REJECTED)
    assert np.array_equal(tri_failed, tri["status"] != capi.FEAT_USED), "the triangulation verdicts differ"
    assert np.array_equal(out["feat_status"][tri_failed], tri["status"][tri_failed])
    ok = ~tri_failed
    assert ok.sum() > 0.9 * prob.F
    assert np.abs(got["p_FinG"][ok] - tri["p_FinG"][ok]).max() < 1e-6
    status = np.where(tri_failed, out["feat_status"], capi.FEAT_USED).astype(np.int32)
    given = dict(p_FinG=got["p_FinG"], p_FinA=got["p_FinA"], anchor_meas=got["anchor_meas"], status=status)
    ref = oracle.msckf_update(opts, v, given=given)
    ref = oracle_with_the_same_gate_verdicts(oracle, opts, v, given, ref, out)
    assert_chi2(out, ref, 1e-8, strict=bool(opts.gate_always_factor))
    assert np.array_equal(out["feat_status"], ref["feat_status"])
    assert ref["stats"]["n_used"] > 0.8 * prob.F
    assert _rel(out["dx"], ref["dx"]) < tol_dx, _rel(out["dx"], ref["dx"])
    assert _rel(out["P"], ref["P"]) < tol_p, _rel(out["P"], ref["P"])
    assert np.abs(out["clone_q_p"] - ref["clone_q_p"]).max() < 1e-9
    return out, ref


def test_cfg3_full_size_end_to_end_with_the_device_triangulation(Updater, oracle):
    """BASELINE configs[2] (2000 features, 100 k measurements) on SURVEY 8(d)'s N = 224 state with nothing injected (bench.py's
    `survey_8d_state` extra; the batch bench.py's HEADLINE times is the next test's)."""
    prob = synth.make_problem(3)
    assert prob.N == 224
    _parity_device_triangulation(Updater, oracle, prob, capi.default_options(chi2_multipler=1.0), key=("cfg3", 2000))


def test_headline_config_end_to_end_with_the_device_triangulation(Updater, oracle):
    """THE configuration bench.py times (bench.py: synth.make_problem(3, imu_intrinsics=True)): BASELINE configs[2] as written — 2000 features,
    30 clones, stereo, online camera AND IMU calibration: the 24 IMU-intrinsic variables of State.cpp:65-88 are in the state (N = 248) and
    get no Jacobian columns (UpdaterHelper.cpp:201-261), their coupling enters through P (UpdaterMSCKF.cpp:209-234, StateHelper.cpp:116-197).
    Nothing injected: device triangulation -> gate -> compression -> update against the oracle."""
    prob = synth.make_problem(3, imu_intrinsics=True)
    assert prob.F == 2000 and prob.N == 248
    _parity_device_triangulation(Updater, oracle, prob, capi.default_options(chi2_multipler=1.0), key=("cfg3_imu", 2000))


@pytest.mark.parametrize("full_gate", [0, 1])
def test_headline_config_against_oracle(Updater, oracle, full_gate):
    """The same N = 248 batch with the oracle's triangulation injected on both sides (positions then agree exactly: every number downstream is
    held to the full-size tolerances), with the default gate and with every gate matrix factored."""
    prob = synth.make_problem(3, imu_intrinsics=True)
    assert prob.F == 2000 and prob.N == 248
    out, ref = _parity(Updater, oracle, prob, capi.default_options(chi2_multipler=1.0, gate_always_factor=full_gate), key=("cfg3_imu", 2000))
    # the IMU-intrinsic block moves through its cross-covariance only: its correction is non-zero and its posterior block the oracle's
    imu_i = slice(15, 39)
    assert np.abs(ref["dx"][imu_i]).max() > 0 and _rel(out["P"][imu_i, imu_i], ref["P"][imu_i, imu_i]) < 1e-9


def test_cfg4_shard_end_to_end_with_the_device_triangulation(Updater, oracle):
    """One of 8 ranks' share of BASELINE configs[3] (4 cameras, 1250 features of ~100 observations: k_feat_y<8, 17>) with nothing injected."""
    prob = synth.make_problem(4, F=1250)
    _parity_device_triangulation(Updater, oracle, prob, capi.default_options(chi2_multipler=1.0), key=("cfg4", 1250))


@pytest.mark.parametrize("full_gate", [0, 1])
def test_cfg3_full_size_against_oracle(Updater, oracle, full_gate):
    """BASELINE configs[2]: 30 clones + online calibration, 2000 features (93 k measurements).  full_gate = 0: the library's
    default, features under the residual bound skip their gate matrix (most of this batch); 1: every gate matrix factored."""
    prob = synth.make_problem(3)
    assert prob.F == 2000 and prob.N == 224
    out, _ = _parity(Updater, oracle, prob, capi.default_options(chi2_multipler=1.0, gate_always_factor=full_gate), key=("cfg3", 2000))
    assert full_gate == 0 or out["stats"]["n_gate_bound"] == 0  # (on THIS snapshot the bound decides nothing either way: 0.57 deg / 5 cm per clone)


def test_gate_bound_decides_a_tight_window(Updater, oracle):
    """The same 2000-feature batch on a window 0.05 x as uncertain (synth.tight_window_problem: a running filter's regime): most
    features pass by the residual bound — no gate matrix formed — and accept sets, dx, P' are still the oracle's."""
    prob = synth.tight_window_problem(3, 0.05)
    out, ref = _parity(Updater, oracle, prob, capi.default_options(chi2_multipler=1.0))
    assert out["stats"]["n_gate_bound"] > 0.8 * out["stats"]["n_used"], out["stats"]
    full, _ = _parity(Updater, oracle, prob, capi.default_options(chi2_multipler=1.0, gate_always_factor=1))
    assert full["stats"]["n_gate_bound"] == 0 and np.array_equal(full["feat_status"], out["feat_status"])
    assert _rel(full["dx"], out["dx"]) < 1e-12 and _rel(full["P"], out["P"]) < 1e-12


def test_cfg4_shard_against_oracle(Updater, oracle):
    """One of 8 ranks' share of BASELINE configs[3]: 4 cameras, N = 252, D = 236, 1250 features of ~100 observations."""
    prob = synth.make_problem(4, F=1250)
    assert prob.K == 4 and prob.N == 252
    _parity(Updater, oracle, prob, capi.default_options(chi2_multipler=1.0), key=("cfg4", 1250))


def test_10k_features_against_oracle(Updater, oracle):
    """north_star's target batch on one GPU: 10 000 MSCKF features on the 30-clone / 15-dof-calibration state."""
    prob = synth.make_problem(2, F=10000)
    assert prob.F == 10000
    _parity(Updater, oracle, prob, capi.default_options(chi2_multipler=1.0))


@pytest.mark.parametrize("F,full_gate", [(240, 0), (240, 1), (800, 0), (800, 1)])
def test_cfg5_geometry_against_oracle(Updater, oracle, F, full_gate):
    """BASELINE configs[4] geometry: 50 clones, 4 cameras, N = 372, D = 356 columns (23 column tiles), tracks of up to 200
    observations (gate matrices of 25 tile rows: k_featy_big.h); F = 800 is a third of one rank's share of the 20 000-feature job (the full
    share, 2500 features, was this test's size until round 5: the oracle needs 80 s for it — the suite's budget — and holds nothing the third does not:
    ~150 k measurements, every CU busy several times over)."""
    prob = synth.make_problem(5, F=F)
    assert prob.C == 50 and prob.K == 4 and prob.N == 372 and np.diff(prob.meas_offsets).max() == 200
    out, _ = _parity(Updater, oracle, prob, capi.default_options(chi2_multipler=1.0, gate_always_factor=full_gate), key=("cfg5", F))
    assert out["route"] == capi.COMPRESS_GRAM  # 23 tile columns: the block variant of the Gram kernel (k_gram_blk), f64


@pytest.mark.parametrize("cfg,F", [(5, 500), (5, 800), (2, 300)])
def test_fp32_gram_variant(Updater, oracle, cfg, F):
    """BASELINE configs[4]'s "fp32 compressed-QR": with options.gram_fp32 the prior-whitened stack leaves the per-feature kernel as
    FLOATS and its Gram matrix is accumulated on v_mfma_f32_32x32x2_f32 (csrc/k_gram32.h: two-level f32 sums inside a workgroup's
    <= 1536 rows, f64 across workgroups).  Everything else stays f64 (gate, whitening, both factorisations), so the accept sets and chi2 are those
    of the f64 path; dx within 1e-4 and P within 1e-3 of the f64 oracle (measured: see the printed line)."""
    prob = synth.make_problem(cfg, F=F)
    opts = capi.default_options(chi2_multipler=1.0, gram_fp32=1)
    key = (f"cfg{cfg}", F)  # the oracle's result does not depend on gram_fp32
    out, ref = _parity(Updater, oracle, prob, opts, tol_dx=1e-4, tol_p=1e-3, key=key)
    assert out["route"] == capi.COMPRESS_GRAM
    assert out["stack_f32"] == 1  # the per-feature kernel stored floats and k_gram_f32 (csrc/k_gram32.h) read them
    print(f"fp32 Gram, cfg {cfg}, {F} features: |ddx|/|dx| {_rel(out['dx'], ref['dx']):.1e}, |dP|/|P| {_rel(out['P'], ref['P']):.1e}")
    f64 = _parity(Updater, oracle, prob, capi.default_options(chi2_multipler=1.0), key=key)[0]
    assert _rel(out["P"], f64["P"]) > 1e-12  # it really is another arithmetic


# --------------------------------------------------------------------------- conditioning of the prior
def _chol_upper_ld(A):
    A = A.astype(LD)
    n = A.shape[0]
    U = np.zeros_like(A)
    for k in range(n):
        d = A[k, k] - (U[:k, k] ** 2).sum()
        U[k, k] = np.sqrt(d)
        U[k, k + 1:] = (A[k, k + 1:] - U[:k, k] @ U[:k, k + 1:]) / U[k, k]
    return U


def _fsub_ld(U, B):  # U^-T B
    B = B.astype(LD)
    Y = np.zeros_like(B)
    for i in range(U.shape[0]):
        Y[i] = (B[i] - U[:i, i] @ Y[:i]) / U[i, i]
    return Y


def extended_precision_update(P, cols, R, rc, sigma2):
    """StateHelper::EKFUpdate for the compressed system (R, rc) evaluated with 64-bit mantissas (x86 long double), in
    the prior-whitened form (every factorisation is of a well-conditioned matrix; residual error ~1e-19 cond)."""
    D = len(cols)
    Pl = P.astype(LD)
    U1 = _chol_upper_ld(Pl[np.ix_(cols, cols)])
    Z = R.astype(LD) @ U1.T
    T = np.eye(D, dtype=LD) + (Z.T @ Z) / LD(sigma2)
    B = _fsub_ld(U1, Pl[cols, :])
    Ct = _chol_upper_ld(T)
    Y2 = _fsub_ld(Ct, B)
    y2 = _fsub_ld(Ct, ((Z.T @ rc.astype(LD)) / LD(sigma2))[:, None])[:, 0]
    return Pl - (B.T @ B - Y2.T @ Y2), Y2.T @ y2


SWEEP = [  # sigma of the global position / yaw, per-clone process noise (rad, m): cond(P_DD) 1e6 ... 3e16
    dict(sigma_g_p=0.05, sigma_g_th=0.01, q_th=1e-3, q_p=5e-3),
    dict(sigma_g_p=0.3, sigma_g_th=0.02, q_th=2e-4, q_p=1e-4),
    dict(sigma_g_p=1.0, sigma_g_th=0.05, q_th=5e-5, q_p=1e-5),
    dict(sigma_g_p=3.0, sigma_g_th=0.1, q_th=5e-5, q_p=1e-5),
    dict(sigma_g_p=10.0, sigma_g_th=0.2, q_th=5e-5, q_p=1e-5),
    dict(sigma_g_p=10.0, sigma_g_th=0.2, q_th=5e-6, q_p=1e-6),
]


# The snapshot's clone estimates are 2 cm / 0.5 deg off the truth, far outside what these tight priors allow: the gate would
# reject every feature.  The sweep is about the linear algebra, so the gate is opened (chi2_multipler 1e12: every feature is
# stacked) and the chi2 VALUES are compared instead.
GATE_OPEN = 1e12


@pytest.mark.parametrize("kw", SWEEP)
def test_prior_conditioning_sweep(Updater, oracle, kw):
    """Every prior of the sweep through (i) the default options — the Gram route while the prior block's Cholesky pivots stay above
    1e-13 of their diagonal entries, the Householder route beyond — and (ii) the Gram route FORCED (pivot tolerance 1e-300), which is
    how the switch-over point was measured: whitened before the Gram accumulation, the route holds the tolerance over the whole
    sweep, cond(P_DD) = 3e16 included, so the switch is a safety margin and not an accuracy cliff."""
    prob = synth.make_problem(2, F=300)
    prob.P = synth.realistic_prior(prob, **kw)
    opts = capi.default_options(chi2_multipler=GATE_OPEN)
    v = capi.Views(prob)
    tri = oracle.triangulate(opts, v)
    ref = oracle.msckf_update(opts, v, want_compressed=True, given=tri)
    cols = oracle.column_map(opts, v)
    PDD = prob.P[np.ix_(cols, cols)]
    ev = np.linalg.eigvalsh(PDD)
    cond = ev[-1] / max(ev[0], 1e-300)
    P_true, dx_true = extended_precision_update(prob.P, cols, ref["H_comp"], ref["r_comp"], opts.sigma_pix ** 2)
    e_ref = (_rel(ref["P"].astype(LD), P_true), _rel(ref["dx"].astype(LD), dx_true))
    assert ref["stats"]["n_used"] > 250
    gate = np.isfinite(ref["chi2"])
    for forced in (False, True):
        up = Updater(capi.default_options(chi2_multipler=GATE_OPEN, prior_pivot_tol=1e-300 if forced else 0.0))
        up.set_problem(prob)
        up.set_triangulation(tri["p_FinG"], tri["p_FinA"], tri["anchor_meas"], tri["status"])
        # mode A first (it leaves the resident state alone): the pivoted factor of the whitened Gram matrix, un-whitened, through the
        # STOCK EKFUpdate — held to the same bounds as the update on the device; where the prior's pivot test fails it is the
        # Householder triangle that comes back
        cmp = up.compress()
        route_a = up.lib.ovgpu_last_update_route(up._ctx)
        st, Pa, dxa = oracle.ekf_update(prob.P, cmp["H"], cmp["r"], cmp["col_cov_id"], opts.sigma_pix ** 2)
        e_a = (_rel(Pa.astype(LD), P_true), _rel(dxa.astype(LD), dx_true))
        print(f"cond(P_DD) {cond:.1e} mode A route {'pivoted gram factor' if route_a == capi.COMPRESS_PCHOLQR else 'tsqr'}{' (forced)' if forced else ''}, "
              f"{cmp['rows']} rows: |dP|/|P| {e_a[0]:.1e} (Householder triangle through the same EKFUpdate {e_ref[0]:.1e}); |ddx|/|dx| {e_a[1]:.1e} ({e_ref[1]:.1e})")
        assert st == 0 and (route_a == capi.COMPRESS_PCHOLQR or not forced)
        assert e_a[0] < max(1e-9, 5 * e_ref[0]) and e_a[1] < max(1e-8, 10 * e_ref[1]), (cond, forced, e_a, e_ref)
        out = up.update()
        up.close()
        assert out["stats"]["status"] == 0
        assert np.array_equal(out["feat_status"], ref["feat_status"])
        assert_chi2(out, ref, 1e-7)
        e_gpu = (_rel(out["P"].astype(LD), P_true), _rel(out["dx"].astype(LD), dx_true))
        print(f"cond(P_DD) {cond:.1e} route {'gram' if out['route'] == capi.COMPRESS_GRAM else 'tsqr'}{' (forced)' if forced else ''}: "
              f"|dP|/|P| gpu {e_gpu[0]:.1e} oracle {e_ref[0]:.1e}; |ddx|/|dx| gpu {e_gpu[1]:.1e} oracle {e_ref[1]:.1e}; "
              f"gpu vs oracle P {_rel(out['P'], ref['P']):.1e} dx {_rel(out['dx'], ref['dx']):.1e}")
        if forced:
            assert out["route"] == capi.COMPRESS_GRAM
        assert e_gpu[0] < max(1e-9, 5 * e_ref[0]), (cond, forced, e_gpu, e_ref)
        assert e_gpu[1] < max(1e-8, 10 * e_ref[1]), (cond, forced, e_gpu, e_ref)
        assert np.array_equal(out["P"], out["P"].T) and np.linalg.eigvalsh(out["P"]).min() > -1e-9 * np.abs(np.diag(out["P"])).max()


def test_round1_route_loses_the_gauge(Updater, oracle):
    """Why the stack is whitened BEFORE its Gram matrix is formed: options.gram_no_whiten = 1 (Gram matrix of the raw stack,
    U1 G U1^T afterwards — round 1's form) carries eps |H|^2 sigma_g^2 into the unobservable directions."""
    prob = synth.make_problem(2, F=300)
    prob.P = synth.realistic_prior(prob, sigma_g_p=10.0, sigma_g_th=0.2)
    opts = capi.default_options(chi2_multipler=GATE_OPEN)
    v = capi.Views(prob)
    tri = oracle.triangulate(opts, v)
    ref = oracle.msckf_update(opts, v, given=tri)
    assert ref["stats"]["n_used"] > 250
    errs = {}
    for nw in (0, 1):
        up = Updater(capi.default_options(chi2_multipler=GATE_OPEN, gram_no_whiten=nw, prior_pivot_tol=1e-300))
        up.set_problem(prob)
        up.set_triangulation(tri["p_FinG"], tri["p_FinA"], tri["anchor_meas"], tri["status"])
        out = up.update()
        up.close()
        assert out["route"] == capi.COMPRESS_GRAM
        errs[nw] = _rel(out["P"], ref["P"])
    print("gram then whiten", errs[1], "whiten then gram", errs[0])
    assert errs[0] < 1e-9 and errs[1] > 10 * errs[0]


# --------------------------------------------------------------------------- camera models, called directly
@pytest.mark.parametrize("fish", [0, 1])
def test_camera_models_bit_exact(Updater, oracle, fish):
    """CamRadtan / CamEqui::distort_d with the float round trip of CamBase.h:130-135 and compute_distort_jacobian
    (CamRadtan.h:127-198, CamEqui.h:136-230): the distorted pixel is bit-exact, the Jacobians agree to round-off."""
    up = Updater(capi.default_options())
    lib = oracle.load()
    dp = capi.c_double_p
    rng = np.random.default_rng(fish)
    cam = np.array((synth._INTRINSICS_EQUI if fish else synth._INTRINSICS)[0])
    n = 20000
    uvn = rng.uniform(-0.6, 0.6, (n, 2))
    uvn[:4] = [[0, 0], [1e-12, 0], [0, -1e-9], [0.6, -0.6]]
    uv, a, b = np.zeros((n, 2)), np.zeros((n, 4)), np.zeros((n, 16))
    capi.check(up.lib.ovgpu_cam_distort(up._ctx, fish, cam.ctypes.data_as(dp), n, uvn.ctypes.data_as(dp), uv.ctypes.data_as(dp),
                                        a.ctypes.data_as(dp), b.ctypes.data_as(dp)), "ovgpu_cam_distort")
    uv2, a2, b2 = np.zeros((n, 2)), np.zeros((n, 4)), np.zeros((n, 16))
    for i in range(n):
        lib.oracle_cam_distort(cam.ctypes.data_as(dp), fish, uvn[i].ctypes.data_as(dp), uv2[i].ctypes.data_as(dp), a2[i].ctypes.data_as(dp),
                               b2[i].ctypes.data_as(dp))
    up.close()
    assert np.array_equal(uv, uv2), f"{(uv != uv2).sum()} of {uv.size} pixels differ"
    assert np.abs(a - a2).max() <= 1e-12 * np.abs(a2).max()
    assert np.abs(b - b2).max() <= 1e-12 * max(np.abs(b2).max(), 1.0)


# --------------------------------------------------------------------------- native multi-GPU entry points
def test_native_sharded_update_world_of_one(Updater, oracle):
    """ovgpu_comm_init_rank + ovgpu_msckf_update_sharded with a communicator of ONE rank (this box has one GPU: RCCL refuses two
    ranks on a device): the native local stage -> exchange -> update chain on one stream must give the plain update bit for bit,
    for the Gram exchange and for the triangle exchange of the Householder route."""
    prob = synth.make_problem(2, F=300)
    for route in (capi.COMPRESS_GRAM, capi.COMPRESS_TSQR):
        opts = capi.default_options(chi2_multipler=1.0, compress_route=route)
        a = Updater(opts)
        a.set_problem(prob)
        ref = a.update()
        a.close()
        b = Updater(opts)
        b.set_problem(prob)
        b.comm_init_single()
        out = b.update_sharded()
        b.reset_state()
        b.update_sharded_async()
        b.synchronize()
        again = b.get_state()
        b.close()
        assert np.array_equal(out["feat_status"], ref["feat_status"])
        assert np.array_equal(out["dx"], ref["dx"]) and np.array_equal(out["P"], ref["P"])
        assert np.array_equal(again["P"], ref["P"])


def test_single_process_multi_device_api(oracle):
    """ovgpu_multi_*: the one-process host interface (feature dealing, per-feature outputs back in the caller's order) on a set of
    one device — the only set this box offers — against the oracle."""
    from open_vins_amd.updater import MultiUpdater
    prob = synth.make_problem(2, F=200, outlier_frac=0.2)
    opts = capi.default_options(chi2_multipler=1.0)
    ref = oracle.msckf_update(opts, capi.Views(prob))
    m = MultiUpdater(opts, devices=[0])
    m.set_problem(prob)
    out = m.update()
    m.close()
    assert np.array_equal(out["feat_status"], ref["feat_status"])
    assert_chi2(out, ref, 1e-7)
    ok = ref["feat_status"] == capi.FEAT_USED
    assert np.abs(out["p_FinG"][ok] - ref["p_FinG"][ok]).max() < 1e-9
    assert _rel(out["dx"], ref["dx"]) < 1e-7 and _rel(out["P"], ref["P"]) < 1e-8
    assert out["stats"]["n_used"] == ref["stats"]["n_used"]


@pytest.mark.parametrize("G,route,F", [(2, "gram", 300), (4, "gram", 301), (2, "tsqr", 300), (3, "tsqr", 200), (2, "gram", 1), (4, "tsqr", 2)])
def test_sharded_update_with_more_than_one_rank_on_hardware(oracle, G, route, F):
    """The G > 1 branches of the feature-sharded update, executed on the GPU although this pool leases one device at a time: G ranks
    share device 0 and the collective is the library's loop-back double (ovgpu_multi_create with a repeated device: the sum in rank
    order / the concatenation is delivered to every rank's stream; RCCL itself refuses two ranks on one GPU).  Everything around the
    transport is the production path: features dealt by track length, local stages on G contexts, the empty shard's zero
    contribution (F < G), Gram all-reduce -> G identical prior-whitened updates, or triangle all-gather -> merge tree -> update.
    Against the oracle, and every rank against rank 0 bit for bit."""
    from open_vins_amd.updater import MultiUpdater
    prob = synth.make_problem(2, F=F, outlier_frac=0.2 if F > 10 else 0.0)
    opts = capi.default_options(chi2_multipler=1.0, compress_route=capi.COMPRESS_TSQR if route == "tsqr" else capi.COMPRESS_GRAM)
    ref = oracle.msckf_update(opts, capi.Views(prob))
    m = MultiUpdater(opts, devices=[0] * G)
    m.set_problem(prob)
    out = m.update()
    assert np.array_equal(out["feat_status"], ref["feat_status"])
    assert_chi2(out, ref, 1e-7)
    assert out["stats"]["n_used"] == ref["stats"]["n_used"]
    if ref["stats"]["n_used"] > 0:
        assert _rel(out["dx"], ref["dx"]) < 1e-7 and _rel(out["P"], ref["P"]) < 1e-8
    # the replicated update: every rank holds the same posterior, bit for bit
    import ctypes as C
    states = []
    for g in range(G):
        ctx = m.lib.ovgpu_multi_ctx(m._m, g)
        P = np.zeros((prob.N, prob.N))
        q = np.zeros((prob.C, 7))
        capi.check(m.lib.ovgpu_get_state(C.c_void_p(ctx), P.ctypes.data_as(capi.c_double_p), q.ctypes.data_as(capi.c_double_p), None, None), "ovgpu_get_state")
        states.append((P, q))
    for P, q in states[1:]:
        np.testing.assert_array_equal(P, states[0][0])
        np.testing.assert_array_equal(q, states[0][1])
    np.testing.assert_array_equal(states[0][0], out["P"])
    m.close()
