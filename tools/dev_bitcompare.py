"""Two builds of libovgpu.so must give the SAME BYTES: `dump <out.npz>` runs a list of MSCKF updates that reach the fused per-feature
kernel of the headline shape (feat::k_feat_y<4, 11, 2>, both stack precisions) and the other per-feature kernels through the library
in the tree and stores every output; `compare <a.npz> <b.npz>` compares two such files bit for bit.  Used when a build changes
nothing but instruction ORDER (a scheduler strategy, ovgpu_featy_tu.hip): no tolerance applies, and no oracle time is spent
(tools/gpu_bitcompare.sh swaps the library files on one box)."""
import sys

import numpy as np

sys.path.insert(0, __file__.rsplit("/tools/", 1)[0])


def shapes():
    from open_vins_amd import capi
    R = capi
    yield "cfg2_400", dict(cfg=2, F=400), {}
    yield "cfg3_headline", dict(cfg=3), {}
    yield "cfg3_every_gate_factored", dict(cfg=3, F=600), dict(gate_always_factor=1)
    yield "cfg3_outliers", dict(cfg=3, F=500, outlier_frac=0.15), {}
    yield "cfg3_ragged", dict(cfg=3, F=500, track="ragged"), {}
    yield "cfg3_fisheye", dict(cfg=3, F=300, fisheye=True), {}
    yield "cfg3_fp32_stack", dict(cfg=3, F=500), dict(gram_fp32=1)
    yield "cfg3_fp32_stack_outliers", dict(cfg=3, F=300, outlier_frac=0.2, track="ragged"), dict(gram_fp32=1, gate_always_factor=1)
    yield "mono_12_clones", dict(cfg=2, F=120, C=12, K=1), {}
    yield "few_clones", dict(cfg=2, F=60, C=5, K=2), {}
    yield "full_inverse_depth", dict(cfg=2, F=200), dict(feat_rep_msckf=R.REP_GLOBAL_FULL_INVERSE_DEPTH)
    yield "no_calibration_no_fej", dict(cfg=2, F=200), dict(do_calib_camera_pose=0, do_calib_camera_intrinsics=0, do_fej=0)
    yield "imu_intrinsics_state", dict(cfg=3, F=300, imu_intrinsics=True), {}
    yield "tight_chi2", dict(cfg=3, F=400), dict(chi2_multipler=0.7)
    yield "one_feature", dict(cfg=2, F=1), {}
    # the 8-wavefront and block-row shapes (translation unit 1: unchanged code, here as the control)
    yield "cfg4_long_tracks", dict(cfg=4, F=300), {}
    yield "cfg5_block_rows", dict(cfg=5, F=60), {}


def prepare(path):
    """The problems are generated once (CPU work: here, not on the GPU box's clock) and travel as a pickle."""
    import pickle
    from open_vins_amd import synth
    probs = {}
    for name, pk, _ in shapes():
        pk = dict(pk)
        probs[name] = synth.make_problem(pk.pop("cfg"), **pk)
    with open(path, "wb") as f:
        pickle.dump(probs, f)


def dump(path, problems=None):
    import pickle
    from open_vins_amd import capi, synth
    from open_vins_amd.updater import UpdaterMSCKF
    probs = pickle.load(open(problems, "rb")) if problems else {}
    out = {}
    for name, pk, ok in shapes():
        pk = dict(pk)
        prob = probs[name] if name in probs else synth.make_problem(pk.pop("cfg"), **pk)
        opts = capi.default_options(**{"chi2_multipler": 1.0, **ok})
        up = UpdaterMSCKF(opts, device=0)
        up.set_problem(prob)
        res = up.update()
        for k in ("feat_status", "chi2", "chi2_thresh", "p_FinG", "dx", "P", "clone_q_p", "calib_q_p", "intrinsics"):
            if k in res:
                out[f"{name}.{k}"] = np.ascontiguousarray(res[k])
        out[f"{name}.route"] = np.array([up.lib.ovgpu_last_update_route(up._ctx), res["stats"]["n_used"], res["stats"]["n_rows"], res["stats"].get("n_gate_bound", 0)])
        # mode A on the same upload: the compressed system the stock EKFUpdate would get
        up.reset_state()
        ca = up.compress()
        for k in ("H", "r", "col_cov_id"):
            if k in ca:
                out[f"{name}.modeA.{k}"] = np.ascontiguousarray(ca[k])
        up.close()
        print(name, "F", prob.F, "used", res["stats"]["n_used"], "route", out[f"{name}.route"][0], flush=True)
    np.savez(path, **out)


def compare(a, b):
    A, B = np.load(a), np.load(b)
    assert sorted(A.files) == sorted(B.files), "different output sets"
    bad = []
    for k in A.files:
        x, y = A[k], B[k]
        if x.shape != y.shape or x.dtype != y.dtype or x.tobytes() != y.tobytes():
            d = float(np.max(np.abs(x.astype(np.float64) - y.astype(np.float64)))) if x.shape == y.shape else float("nan")
            bad.append((k, d))
    print(f"{len(A.files)} arrays compared, {len(bad)} differ")
    for k, d in bad[:40]:
        print("  DIFFERS", k, "max |a - b| =", d)
    return 1 if bad else 0


if __name__ == "__main__":
    if sys.argv[1] == "prepare":
        prepare(sys.argv[2])
    elif sys.argv[1] == "dump":
        dump(sys.argv[2], sys.argv[3] if len(sys.argv) > 3 else None)
    else:
        sys.exit(compare(sys.argv[2], sys.argv[3]))
