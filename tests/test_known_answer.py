"""The known-answer fixtures of tests/golden/known_answer_msckf_*.json.gz pin the oracle — and, on the GPU, the library.

The fixtures come from tools/make_known_answer.py: an evaluation of the reference's formulas (cited there) in 50-digit arithmetic
with the reference's float32 operations emulated, written WITHOUT looking at oracle/ov_oracle.cpp or at the kernels.  A misreading
of the reference shared by the oracle and the kernels would pass every oracle-vs-GPU test; it does not pass these.

Tolerances.  The oracle and the GPU compute in float64; the fixture is exact to 25 digits.  What is asserted is therefore the
float64 round-off of each stage on this snapshot (measured, then fixed with a margin of ~10x):
  * triangulation + Levenberg loop          p_FinA / p_FinG     1e-11 m (cond(A) of the 3 x 3 system is 2e3 .. 6e3; measured 1.4e-12)
  * predicted pixels (float32 quantised)    bit-exact
  * H_f, H_x blocks, residual               1e-12 relative to the block's largest entry
  * chi2                                    1e-12 relative (measured 2e-14), thresholds 1e-11, accept sets identical
  * stack: H^T H, H^T r                     1e-13 relative, Frobenius (measured 8e-15)
  * dx                                      1e-11 relative (measured 1e-13);  P' 1e-13 relative, Frobenius (1e-15);  poses 1e-13
    (with the fixture's positions injected; end to end, through the oracle's own loop A, 100x these)
The GPU is held to the parity tolerances of tests/test_gpu_parity.py (1e-9 m / 1e-8 / 1e-8 / 1e-9) against the SAME file.
"""
import gzip
import json
import os

import numpy as np
import pytest

from open_vins_amd import capi, synth

GOLDEN = os.path.join(os.path.dirname(__file__), "golden")
CASES = ["radtan_fej", "equi_nofej_invdepth"]
STATUS = dict(USED=capi.FEAT_USED, TRI_FAILED=capi.FEAT_TRI_FAILED, GN_FAILED=capi.FEAT_GN_FAILED, CHI2_REJECTED=capi.FEAT_CHI2_REJECTED)


def _arr(x):
    """nested lists of decimal strings -> float64 array (each string is rounded correctly to the nearest double)."""
    if isinstance(x, str):
        return float(x)
    return np.array([_arr(v) for v in x], dtype=np.float64)


def _upper_to_full(rows):
    n = len(rows)
    A = np.zeros((n, n))
    for i, r in enumerate(rows):
        A[i, i:] = [float(v) for v in r]
    return A + np.triu(A, 1).T


def load_case(name):
    with gzip.open(os.path.join(GOLDEN, f"known_answer_msckf_{name}.json.gz"), "rb") as fh:
        doc = json.loads(fh.read().decode())
    i = doc["inputs"]
    F = len(i["meas_offsets"]) - 1
    prob = synth.Problem(
        cfg=0, seed=0, N=i["N"], C=i["C"], K=i["K"], P=np.array(i["P"], dtype=np.float64),
        clone_q_p=np.array(i["clone_q_p"]), clone_q_p_fej=np.array(i["clone_q_p_fej"]), clone_q_p_true=np.array(i["clone_q_p"]),
        clone_cov_id=np.array(i["clone_cov_id"], dtype=np.int32), calib_q_p=np.array(i["calib_q_p"]), calib_q_p_true=np.array(i["calib_q_p"]),
        intrinsics=np.array(i["intrinsics"]), cam_is_fisheye=np.array(i["cam_is_fisheye"], dtype=np.uint8),
        calib_cov_id=np.array(i["calib_cov_id"], dtype=np.int32), intr_cov_id=np.array(i["intr_cov_id"], dtype=np.int32),
        meas_offsets=np.array(i["meas_offsets"], dtype=np.int32), uv=np.array(i["uv"], dtype=np.float32), uvn=np.array(i["uvn"], dtype=np.float32),
        clone_idx=np.array(i["clone_idx"], dtype=np.int32), cam_idx=np.array(i["cam_idx"], dtype=np.int32), p_FinG_true=np.zeros((F, 3)))
    o = doc["options"]
    opts = capi.default_options(**{k: (int(v) if k in ("triangulate_1d", "refine_features", "max_runs", "do_fej", "do_calib_camera_pose",
                                                       "do_calib_camera_intrinsics", "feat_rep_msckf") else float(v)) for k, v in o.items()})
    return doc, prob, opts


def _rel(a, b):
    return np.linalg.norm(np.asarray(a) - np.asarray(b)) / max(np.linalg.norm(b), 1e-300)


def _dense_H_x(ans_feat, prob, f, cols, opts):
    a, b = int(prob.meas_offsets[f]), int(prob.meas_offsets[f + 1])
    H = np.zeros((2 * (b - a), len(cols)))
    cols = list(cols)
    for c, i in enumerate(range(a, b)):
        blk = ans_feat["H_x_blocks"][c]
        j, k = int(prob.clone_idx[i]), int(prob.cam_idx[i])
        c0 = cols.index(int(prob.clone_cov_id[j]))
        H[2 * c:2 * c + 2, c0:c0 + 6] = _arr(blk["clone"])
        if opts.do_calib_camera_pose:
            c0 = cols.index(int(prob.calib_cov_id[k]))
            H[2 * c:2 * c + 2, c0:c0 + 6] = _arr(blk["extrinsics"])
        if opts.do_calib_camera_intrinsics:
            c0 = cols.index(int(prob.intr_cov_id[k]))
            H[2 * c:2 * c + 2, c0:c0 + 8] = _arr(blk["intrinsics"])
    return H


def test_fixture_decisions_are_not_razor_edge():
    """Every branch the fixture took (anchor, Levenberg accept / reject, gate, ...) is at least 1e-9 (relative) from flipping: a
    float64 evaluation (round-off 1e-15) takes the same path."""
    for name in CASES:
        doc, _, _ = load_case(name)
        assert min(doc["decision_margins"].values()) > 1e-9, doc["decision_margins"]
        st = [f["status"] for f in doc["answer"]["features"]]
        assert "USED" in st
    doc, _, _ = load_case("radtan_fej")
    st = [f["status"] for f in doc["answer"]["features"]]
    assert "CHI2_REJECTED" in st and "GN_FAILED" in st  # the gate and the baseline check both reject something in this snapshot


@pytest.mark.parametrize("name", CASES)
def test_oracle_against_the_known_answer(oracle, name):
    doc, prob, opts = load_case(name)
    ans = doc["answer"]
    v = capi.Views(prob)
    cols = oracle.column_map(opts, v)
    assert list(cols) == ans["col_cov_id"]

    # ---- loop A: anchor rule, linear triangulation, Levenberg loop on float32 costs (FeatureInitializer.cpp:30-375)
    tri = oracle.triangulate(opts, v)
    worst = dict(tri=0.0, Hf=0.0, Hx=0.0, res=0.0, chi2=0.0)
    for f, af in enumerate(ans["features"]):
        assert tri["status"][f] == (STATUS[af["status"]] if af["status"] in ("TRI_FAILED", "GN_FAILED") else capi.FEAT_USED), (f, af["status"])
        assert tri["anchor_meas"][f] == af["anchor_meas"]
        if af["status"] in ("TRI_FAILED", "GN_FAILED"):
            continue
        e = max(np.abs(tri["p_FinA"][f] - _arr(af["p_FinA"])).max(), np.abs(tri["p_FinG"][f] - _arr(af["p_FinG"])).max())
        worst["tri"] = max(worst["tri"], e)
        assert e < 1e-11, (f, e)  # cond(A) ~ 2e3 .. 6e3 amplifies the 1e-16 of float64: ~1e-12 m

    # ---- Jacobians on the FIXTURE's positions (so that loop A's round-off does not enter): UpdaterHelper.cpp:192-424
    for f, af in enumerate(ans["features"]):
        if "H_f" not in af:
            continue
        H_f, H_x, res = oracle.feature_jacobian(opts, v, f, _arr(af["p_FinG"]), _arr(af["p_FinA"]), af["anchor_meas"])
        ref_Hf, ref_Hx, ref_res = _arr(af["H_f"]), _dense_H_x(af, prob, f, cols, opts), _arr(af["res"])
        # the predicted pixel is float32-quantised (CamBase.h:130-135): the residual is an exact difference of two floats
        a = int(prob.meas_offsets[f])
        uvm = prob.uv.reshape(-1, 2)[a:a + len(ref_res) // 2].astype(np.float64).reshape(-1)
        assert np.array_equal((uvm - res).astype(np.float32), np.asarray(af["pixels_predicted"], dtype=np.float32).reshape(-1)), f
        assert np.array_equal(res, ref_res), f
        worst["Hf"] = max(worst["Hf"], np.abs(H_f - ref_Hf).max() / np.abs(ref_Hf).max())
        worst["Hx"] = max(worst["Hx"], np.abs(H_x - ref_Hx).max() / np.abs(ref_Hx).max())
        assert np.abs(H_f - ref_Hf).max() < 1e-12 * np.abs(ref_Hf).max()
        assert np.abs(H_x - ref_Hx).max() < 1e-12 * np.abs(ref_Hx).max()
        assert np.array_equal(H_x == 0.0, ref_Hx == 0.0)  # the sparsity pattern

    # ---- the complete update with the fixture's triangulation injected: gate, stack, compression, EKF (UpdaterMSCKF.cpp:144-285)
    F = prob.F
    given = dict(p_FinG=np.array([_arr(af["p_FinG"]) if "p_FinG" in af else np.zeros(3) for af in ans["features"]]),
                 p_FinA=np.array([_arr(af["p_FinA"]) if "p_FinA" in af else np.zeros(3) for af in ans["features"]]),
                 anchor_meas=np.array([af["anchor_meas"] for af in ans["features"]], dtype=np.int32),
                 status=np.array([STATUS[af["status"]] if af["status"] in ("TRI_FAILED", "GN_FAILED") else capi.FEAT_USED for af in ans["features"]],
                                 dtype=np.int32))
    ref = oracle.msckf_update(opts, v, want_compressed=True, given=given)
    _check_update(ref, ans, prob, tol=dict(chi2=1e-12, gram=1e-13, dx=1e-11, P=1e-13, pose=1e-13), worst=worst)
    # ---- and end to end (the oracle's own loop A)
    ref = oracle.msckf_update(opts, v, want_compressed=True)
    _check_update(ref, ans, prob, tol=dict(chi2=1e-10, gram=1e-11, dx=1e-9, P=1e-11, pose=1e-11))
    print(f"\n[{name}] oracle vs known answer: " + ", ".join(f"{k} {x:.1e}" for k, x in worst.items()))


def _check_update(out, ans, prob, tol, worst=None):
    want = np.array([STATUS[af["status"]] for af in ans["features"]], dtype=np.int32)
    assert np.array_equal(out["feat_status"], want)
    for f, af in enumerate(ans["features"]):
        if "chi2" in af:
            assert out["chi2"][f] == pytest.approx(float(af["chi2"]), rel=tol["chi2"]), f
            assert out["chi2_thresh"][f] == pytest.approx(float(af["chi2_thresh"]), rel=1e-11), f
            if worst is not None:
                worst["chi2"] = max(worst["chi2"], abs(out["chi2"][f] / float(af["chi2"]) - 1))
    G, g, rr = _upper_to_full(ans["Gram_stack_upper"]), _arr(ans["gvec_stack"]), float(ans["rr_stack"])
    if "H_comp" in out:
        Hc, rc = out["H_comp"], out["r_comp"]
        assert Hc.shape[0] == ans["rows_compressed"]
        eG, eg = _rel(Hc.T @ Hc, G), _rel(Hc.T @ rc, g)
        assert eG < tol["gram"] and eg < tol["gram"], (eG, eg)
        if worst is not None:
            worst["gram"], worst["gvec"] = eG, eg
        if ans["rows_stacked"] > ans["D"]:
            assert np.abs(np.tril(Hc, -1)).max() <= 1e-12 * np.abs(Hc).max()  # measurement_compress_inplace leaves a triangle
    if "stats" in out and out["stats"].get("n_rows") is not None:
        assert out["stats"]["n_rows"] == ans["rows_stacked"]
    e_dx, e_P = _rel(out["dx"], _arr(ans["dx"])), _rel(out["P"], _upper_to_full(ans["P_post_upper"]))
    assert e_dx < tol["dx"] and e_P < tol["P"], (e_dx, e_P)
    e_pose = max(np.abs(out["clone_q_p"] - _arr(ans["clone_q_p_post"])).max(), np.abs(out["calib_q_p"] - _arr(ans["calib_q_p_post"])).max())
    assert e_pose < tol["pose"], e_pose
    assert np.abs(out["intrinsics"] - _arr(ans["intrinsics_post"])).max() < 1e3 * tol["pose"]  # focal lengths are ~460
    if worst is not None:
        worst.update(dx=e_dx, P=e_P, pose=e_pose)


# ------------------------------------------------------------------------------------------------ the HIP library against the same file
@pytest.mark.gpu
@pytest.mark.parametrize("name", CASES)
@pytest.mark.parametrize("route", ["gram", "tsqr"])
def test_gpu_against_the_known_answer(name, route):
    import torch
    assert torch.cuda.is_available(), "needs a GPU"
    from open_vins_amd.updater import UpdaterMSCKF
    doc, prob, opts = load_case(name)
    opts.compress_route = capi.COMPRESS_TSQR if route == "tsqr" else capi.COMPRESS_GRAM
    opts.gate_always_factor = 1  # every chi2 is the reference's statistic (the default's residual bound: the leg at the end)
    ans = doc["answer"]
    up = UpdaterMSCKF(opts)
    up.set_problem(prob)
    tri = up.triangulate()
    for f, af in enumerate(ans["features"]):
        assert tri["anchor_meas"][f] == af["anchor_meas"]
        if af["status"] in ("TRI_FAILED", "GN_FAILED"):
            assert tri["status"][f] == STATUS[af["status"]]
            continue
        assert tri["status"][f] == capi.FEAT_USED
        assert np.abs(tri["p_FinA"][f] - _arr(af["p_FinA"])).max() < 1e-9
        assert np.abs(tri["p_FinG"][f] - _arr(af["p_FinG"])).max() < 1e-9
    out = up.update()
    _check_update(out, ans, prob, tol=dict(chi2=1e-8, gram=1e-9, dx=1e-8, P=1e-9, pose=1e-9))
    assert np.array_equal(out["P"], out["P"].T)
    # mode A: the compressed system handed to the stock StateHelper::EKFUpdate carries the fixture's Gram matrices
    up.reset_state()
    comp = up.compress()
    G, g = _upper_to_full(ans["Gram_stack_upper"]), _arr(ans["gvec_stack"])
    Hc, rc = comp["H"], comp["r"]
    assert list(comp["col_cov_id"]) == ans["col_cov_id"]
    assert _rel(Hc.T @ Hc, G) < 1e-9 and _rel(Hc.T @ rc, g) < 1e-9
    up.close()
    # the library's default: features whose residual bound is under the threshold skip their gate matrix — same verdicts, same
    # update, and a reported statistic between the known answer's chi2 and the threshold
    opts.gate_always_factor = 0
    up = UpdaterMSCKF(opts)
    up.set_problem(prob)
    out0 = up.update()
    up.close()
    assert np.array_equal(out0["feat_status"], out["feat_status"])
    assert _rel(out0["dx"], out["dx"]) < 1e-12 and _rel(out0["P"], out["P"]) < 1e-12
    n_bound = 0
    for f, af in enumerate(ans["features"]):
        if "chi2" in af and out0["chi2"][f] != out["chi2"][f]:
            n_bound += 1
            assert float(af["chi2"]) * (1 - 1e-9) <= out0["chi2"][f] <= out0["chi2_thresh"][f] and out0["feat_status"][f] == capi.FEAT_USED
    assert n_bound <= out0["stats"]["n_gate_bound"]
