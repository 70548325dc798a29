"""Run by tests/test_dropin_build.py in a SUBPROCESS (a C++ exception of the shim crossing the C driver would take the caller down):
`python dropin_probe.py <a|b>` runs the reference's VioManager-side calls — UpdaterMSCKF::update, UpdaterSLAM::update / delayed_init /
change_anchors on a reference `State` built by oracle/ref/ref_driver.cpp — once through oracle/_ref/libov_ref.so (the reference's own
updaters) and once through oracle/_ref/libov_dropin_<mode>.so (the SAME driver and reference classes with open_vins_amd/shim's
translation units in place of the reference's updaters, on the GPU through libovgpu), and prints one JSON line per case with the
deviations.  Modes a_cpu / b_cpu run the same drop-in build linked against tests/fake_ovgpu (the C ABI served by the CPU oracle): the shim's
C++ end to end without a GPU.  TEST INFRASTRUCTURE."""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)

from open_vins_amd import capi, synth  # noqa: E402
from oracle import pyref  # noqa: E402


def rel(a, b):
    m = np.isfinite(a) & np.isfinite(b)
    return float(np.linalg.norm(a[m] - b[m]) / max(np.linalg.norm(b[m]), 1e-300)) if m.any() else 0.0


def emit(case, **kw):
    print(json.dumps(dict(case=case, **kw)), flush=True)


def state_dev(a, b):
    return float(max(np.abs(a[k] - b[k]).max() for k in ("clone_q_p", "calib_q_p", "intrinsics")))


def msckf(mode, seed):
    from test_ref_build import _msckf_case
    prob, opts = _msckf_case(seed)
    v = capi.Views(prob)
    ref = pyref.msckf_update(opts, v)
    with pyref.using(pyref.dropin_path(mode)):
        got = pyref.msckf_update(opts, capi.Views(prob))
    tri = (ref["feat_status"] == capi.FEAT_USED) | (ref["feat_status"] == capi.FEAT_CHI2_REJECTED)
    used = int((ref["feat_status"] == capi.FEAT_USED).sum())
    emit(f"msckf:{seed}", F=int(prob.F), C=int(prob.C), K=int(prob.K), rep=int(opts.feat_rep_msckf), used=used,
         status_equal=bool(np.array_equal(got["feat_status"], ref["feat_status"])),
         pos=float(np.abs(got["p_FinG"] - ref["p_FinG"])[tri].max()) if tri.any() else 0.0,
         dx=rel(got["dx"], ref["dx"]) if used else 0.0, P=rel(got["P"], ref["P"]), state=state_dev(got, ref))


def slam(mode, rep):
    prob = synth.make_slam_problem(2, L=10, lm_rep=rep)
    opts = capi.default_options(chi2_multipler=1.0)
    ref = pyref.slam_update(opts, capi.Views(prob))
    with pyref.using(pyref.dropin_path(mode)):
        got = pyref.slam_update(opts, capi.Views(prob))
    emit(f"slam:{rep}", used=int((ref["feat_status"] == capi.FEAT_USED).sum()), status_equal=bool(np.array_equal(got["feat_status"], ref["feat_status"])),
         dx=rel(got["dx"], ref["dx"]), P=rel(got["P"], ref["P"]), landmarks=float(np.abs(got["landmarks"] - ref["landmarks"]).max()), state=state_dev(got, ref))


def _aruco(F):
    """UpdaterSLAM's second option set (UpdaterSLAM.cpp:226-232, :392-409): 40 % of the features are ArUco corners with their own sigma / multiplier."""
    tag = np.random.default_rng(5).random(F) < 0.4
    return np.where(tag, 2.5, 1.0), np.where(tag, 3.0, 1.0)


def slam_aruco(mode, rep):
    if mode == "a_cpu":  # (tests/fake_ovgpu does not model the row scaling of ovgpu_slam_compress under per-feature sigma)
        return
    prob = synth.make_slam_problem(2, L=10, lm_rep=rep)
    opts = capi.default_options(chi2_multipler=1.0)
    sig, mult = _aruco(prob.F)
    ref = pyref.slam_update(opts, capi.Views(prob), feat_sigma=sig, feat_chi2mult=mult)
    with pyref.using(pyref.dropin_path(mode)):
        got = pyref.slam_update(opts, capi.Views(prob), feat_sigma=sig, feat_chi2mult=mult)
    emit(f"slam:aruco{rep}", used=int((ref["feat_status"] == capi.FEAT_USED).sum()), status_equal=bool(np.array_equal(got["feat_status"], ref["feat_status"])),
         dx=rel(got["dx"], ref["dx"]), P=rel(got["P"], ref["P"]), landmarks=float(np.abs(got["landmarks"] - ref["landmarks"]).max()), state=state_dev(got, ref))


_MIXES = [[4, 0], [0, 3], [5, 0], [2, 5], [0, 1, 2, 3, 4, 5]]  # feat_rep_slam next to feat_rep_aruco; all six at once


def slam_mixed(mode, k):
    """Round 5 (ABI 7): SLAM landmarks and ArUco corners kept in DIFFERENT representations in one UpdaterSLAM::update — the reference stacks them
    into one Hx_big (UpdaterSLAM.cpp:427-447); the shim hands every landmark over with its own representation and makes ONE library call."""
    each = np.array([_MIXES[k][l % len(_MIXES[k])] for l in range(12)], np.int32)
    prob = synth.make_slam_problem(2, L=12, lm_rep=each)
    opts = capi.default_options(chi2_multipler=1.0)
    tag = each == _MIXES[k][-1]
    for name, kw in (("", {}), ("aruco", dict(feat_sigma=np.where(tag, 2.5, 1.0), feat_chi2mult=np.where(tag, 3.0, 1.0)))):
        if name and mode == "a_cpu":  # (tests/fake_ovgpu does not model the row scaling of ovgpu_slam_compress under per-feature sigma)
            continue
        ref = pyref.slam_update(opts, capi.Views(prob), **kw)
        with pyref.using(pyref.dropin_path(mode)):
            got = pyref.slam_update(opts, capi.Views(prob), **kw)
        emit(f"slam:mixed{name}{k}", used=int((ref["feat_status"] == capi.FEAT_USED).sum()), status_equal=bool(np.array_equal(got["feat_status"], ref["feat_status"])),
             dx=rel(got["dx"], ref["dx"]), P=rel(got["P"], ref["P"]), landmarks=float(np.abs(got["landmarks"] - ref["landmarks"]).max()), state=state_dev(got, ref))


def delayed_mixed(mode, k):
    """UpdaterSLAM::delayed_init with feat_rep_aruco != feat_rep_slam (UpdaterSLAM.cpp:160-166): the corners of the batch are initialised in one
    representation, the other features in another, in one chain; next to resident landmarks of a third (k = 1)."""
    rep_slam, rep_aruco = ((4, 0), (0, 5), (5, 2))[k]
    tag = np.random.default_rng(5).random(16) < 0.4
    prob = synth.make_problem(2, F=16, outlier_frac=0.2) if k != 1 else synth.make_slam_problem(2, L=16, lm_rep=np.array([4, 2] * 8, np.int32), outlier_frac=0.2)
    opts = capi.default_options(chi2_multipler=1.0)
    for name, kw in (("", dict(feat_is_aruco=tag)), ("aruco", dict(feat_sigma=np.where(tag, 2.5, 1.0), feat_chi2mult=np.where(tag, 3.0, 1.0)))):
        ref = pyref.slam_delayed_init(opts, capi.Views(prob), feat_rep=rep_slam, feat_rep_aruco=rep_aruco, **kw)
        with pyref.using(pyref.dropin_path(mode)):
            got = pyref.slam_delayed_init(opts, capi.Views(prob), feat_rep=rep_slam, feat_rep_aruco=rep_aruco, **kw)
        acc = ref["lm_cov_id"] >= 0
        same = bool(np.array_equal(got["feat_status"], ref["feat_status"]) and got["N"] == ref["N"] and np.array_equal(got["lm_cov_id"], ref["lm_cov_id"]))
        old = float(np.abs(got["landmarks_existing"] - ref["landmarks_existing"]).max()) if ref["landmarks_existing"].size else 0.0
        emit(f"delayed:mixed{name}{k}", accepted=int(acc.sum()), status_equal=same,
             value=max(float(np.abs(got["lm_value"][acc] - ref["lm_value"][acc]).max()), old) if same and acc.any() else -1.0,
             P=rel(got["P"], ref["P"]) if same else -1.0, state=state_dev(got, ref))


def anchors_mixed(mode, _):
    each = np.array([4, 0, 5, 2, 1, 3, 4, 0, 5, 2], np.int32)
    prob = synth.make_slam_problem(2, L=10, lm_rep=each)
    opts = capi.default_options(chi2_multipler=1.0)
    ref = pyref.change_anchors(opts, capi.Views(prob))
    with pyref.using(pyref.dropin_path(mode)):
        got = pyref.change_anchors(opts, capi.Views(prob))
    emit("anchors:mixed", moved=int(((prob.lm_anchor_clone == 0) & (each >= 2)).sum()), status_equal=bool(np.array_equal(got["anchor_clone"], ref["anchor_clone"])),
         P=rel(got["P"], ref["P"]), value=float(np.abs(got["value"] - ref["value"]).max()), fej=float(np.abs(got["fej"] - ref["fej"]).max()))


def delayed(mode, rep, aruco=False):
    prob = synth.make_problem(2, F=16, outlier_frac=0.2)
    opts = capi.default_options(chi2_multipler=1.0)
    kw = dict(zip(("feat_sigma", "feat_chi2mult"), _aruco(16))) if aruco else {}
    ref = pyref.slam_delayed_init(opts, capi.Views(prob), feat_rep=rep, **kw)
    with pyref.using(pyref.dropin_path(mode)):
        got = pyref.slam_delayed_init(opts, capi.Views(prob), feat_rep=rep, **kw)
    acc = ref["lm_cov_id"] >= 0
    same = bool(np.array_equal(got["feat_status"], ref["feat_status"]) and got["N"] == ref["N"] and np.array_equal(got["lm_cov_id"], ref["lm_cov_id"]))
    emit(f"delayed:{'aruco' if aruco else ''}{rep}", accepted=int(acc.sum()), status_equal=same,
         value=float(np.abs(got["lm_value"][acc] - ref["lm_value"][acc]).max()) if same and acc.any() else -1.0,
         P=rel(got["P"], ref["P"]) if same else -1.0, state=state_dev(got, ref))


def delayed_aruco(mode, rep):
    delayed(mode, rep, aruco=True)


def zupt(mode, moving):
    """UpdaterZeroVelocity::try_update (UpdaterZeroVelocity.cpp:64-332) with INTEGRATION.md's patch applied to the reference's own file at build time
    (oracle/ref/patch_zupt.py: the chi2 on the marginal covariance, the bias random walk and the update through shim/ovgpu_zupt.h; mode B builds)
    against the unpatched function: an IMU at rest (accepted: bias propagation + EKF update, the state time moves) and one that moves (rejected)."""
    prob = synth.make_problem(2, F=1, C=8)
    opts = capi.default_options(chi2_multipler=1.0)
    rng = np.random.default_rng(3)
    t_state = 10.0 + 0.1 * (prob.C - 1)  # the driver's clone times are 10.0, 10.1, ...; the state sits at the newest clone
    t_update = t_state + 0.1
    t = np.arange(t_state - 0.02, t_update + 0.02, 1.0 / 400.0)
    bg, ba = np.array([0.002, -0.001, 0.0015]), np.array([0.02, -0.01, 0.015])
    imu = np.concatenate([[0, 0, 0, 1], [0.3, -0.2, 1.1], [0.4, 0, 0] if moving else [0.001, -0.002, 0.0005], bg, ba])
    wm = bg + 1.6968e-04 * np.sqrt(400.0) * rng.standard_normal((len(t), 3)) + ([0.2, 0, 0] if moving else 0)
    am = np.array([0, 0, 9.81]) + ba + 2.0e-3 * np.sqrt(400.0) * rng.standard_normal((len(t), 3))
    args = (opts, None, imu, t, wm, am, t_state, t_update)
    ref = pyref.zupt_try_update(args[0], capi.Views(prob), *args[2:])
    with pyref.using(pyref.dropin_path(mode)):
        got = pyref.zupt_try_update(args[0], capi.Views(prob), *args[2:])
    m = np.isfinite(ref["dx"])
    emit(f"zupt:{'moving' if moving else 'rest'}", accepted=bool(ref["accepted"]), status_equal=bool(got["accepted"] == ref["accepted"] and got["timestamp"] == ref["timestamp"]),
         dx=rel(got["dx"][m], ref["dx"][m]) if ref["accepted"] else float(np.abs(got["dx"][m] - ref["dx"][m]).max()), P=rel(got["P"], ref["P"]),
         state=float(np.abs(got["imu"] - ref["imu"]).max()))


def anchors(mode, rep):
    prob = synth.make_slam_problem(2, L=10, lm_rep=rep)
    opts = capi.default_options(chi2_multipler=1.0)
    ref = pyref.change_anchors(opts, capi.Views(prob))
    with pyref.using(pyref.dropin_path(mode)):
        got = pyref.change_anchors(opts, capi.Views(prob))
    emit(f"anchors:{rep}", moved=int((prob.lm_anchor_clone == 0).sum()), status_equal=bool(np.array_equal(got["anchor_clone"], ref["anchor_clone"])),
         P=rel(got["P"], ref["P"]), value=float(np.abs(got["value"] - ref["value"]).max()), fej=float(np.abs(got["fej"] - ref["fej"]).max()))


def loop_slam(mode, seconds):
    """The same loop with SLAM landmarks in the filter (config/rpng_sim/estimator_config.yaml ships max_slam: 50; here 25, delay 2 s): VioManager's
    landmark handling (VioManager.cpp:430-491, :529-547, :585) around UpdaterSLAM::update / delayed_init / change_anchors — the shim's, against the
    reference's."""
    loop(mode, seconds, name="loop_slam", max_slam_features=25, dt_slam_delay=2.0)


def loop_stereo(mode, seconds):
    """Two cameras, anchored representations (MSCKF features and SLAM landmarks as ANCHORED_MSCKF_INVERSE_DEPTH): camera groups inside a track, the
    anchor rule on tied counts, and UpdaterSLAM::change_anchors moving live landmarks every time their anchor clone is about to be marginalised."""
    loop(mode, seconds, name="loop_stereo", num_cameras=2, max_slam_features=15, dt_slam_delay=2.0, feat_rep_slam=4, feat_rep_msckf=4)


def loop_wide(mode, seconds):
    """Two seconds of the loop with a stereo rig and 3 000 points in view: > 256 tracks per update, where the shim splits its clean + flatten walk
    over threads (shim/ovgpu_shim_common.h: clean_flatten_batch) and compacts the accepted tracks in one pass — the same decisions, in the same
    order, as the reference's erase-as-you-go loops."""
    loop(mode, 2.0, name="loop_wide", num_cameras=2, num_pts=3000, max_msckf_in_update=4000)


def loop(mode, seconds, name="loop", **cfg):
    """The rpng_sim closed loop (BASELINE configs[0]; tests/test_rpng_sim_loop.py) with the DROP-IN as the filter's updater: the reference's
    Simulator, Propagator, FeatureDatabase and State around open_vins_amd/shim/UpdaterMSCKF.cpp, against the same loop around the reference's
    UpdaterMSCKF.cpp; the control = two runs of the reference 1e-13 m apart at the start."""
    from test_rpng_sim_loop import _ate, run_filter, separation
    ref = run_filter("reference", seconds=seconds, **cfg)
    ctl = run_filter("reference", seconds=seconds, perturb=1e-13, **cfg)
    traffic = None
    with pyref.using(pyref.dropin_path(mode)) as lib:
        if mode.startswith("c"):  # the resident-covariance builds count the N x N copies they make
            import ctypes
            before = (ctypes.c_long(0), ctypes.c_long(0))
            lib.ovgpu_shim_resident_cov_traffic(ctypes.byref(before[0]), ctypes.byref(before[1]))
        got = run_filter("reference", seconds=seconds, **cfg)  # ("reference" = the library's own updaters: here the shim's)
        if mode.startswith("c"):
            after = (ctypes.c_long(0), ctypes.c_long(0))
            lib.ovgpu_shim_resident_cov_traffic(ctypes.byref(after[0]), ctypes.byref(after[1]))
            traffic = dict(cov_uploads=after[0].value - before[0].value, cov_downloads=after[1].value - before[1].value)
    n = min(len(got["used"]), len(ref["used"]))
    same_first = all(np.array_equal(got["used"][k], ref["used"][k]) for k in range(min(n, 100)))
    n_dec = sum(len(u) for u in ref["used"][:n])
    n_diff = sum(int((u != v).sum()) if len(u) == len(v) else len(v) for u, v in zip(got["used"][:n], ref["used"][:n]))
    d = separation(got, ref)
    a, b = _ate(got), _ate(ref)
    emit(f"{name}:{seconds}", updates=len(got["used"]), state_dim_max=int(got["N"].max()), state_dim_max_reference=int(ref["N"].max()), updates_reference=len(ref["used"]), status_equal=bool(same_first and len(got["used"]) == len(ref["used"])),
         decisions=int(n_dec), differing=int(n_diff), sep_first_ten=float(d[:10].max()), sep=float(d.max()), control=float(separation(ctl, ref).max()),
         ate_deg=a[0], ate_m=a[1], ate_deg_reference=b[0], ate_m_reference=b[1], **(traffic or {}))


_LAPS = ("snapshot", "clean_flatten", "state_handover", "track_handover", "call", "triangulation_readback", "track_side_effects", "state_writeback")


def _laps(lib, reset=False):
    """UpdaterMSCKF::update's own stopwatch (test builds: -DOVGPU_SHIM_TIMING), ms per update since the last reset."""
    import ctypes
    try:
        fn = lib.ovgpu_shim_update_laps
    except AttributeError:
        return None
    out = (ctypes.c_double * 9)()
    fn(out, 1 if reset else 0)
    n = max(out[8], 1.0)
    return {k: round(out[i] / n, 4) for i, k in enumerate(_LAPS)}


def time_loop(mode, seconds=6.0, **cfg):
    """The drop-in's cost INSIDE a running filter at a large shape: the reference's simulator / propagator / database around the shim with a stereo
    rig, a 30-clone window and as many tracks per frame as the simulator is asked for — wall time per frame inside Propagator::propagate_and_clone
    (StateHelper::EKFPropagation + augment_clone), UpdaterMSCKF::update and StateHelper::marginalize_old_clone, second half of the run."""
    from oracle import refsim
    kw = dict(num_cameras=2, max_clones=30, num_pts=1100, max_msckf_in_update=4000)
    kw.update(cfg)
    with pyref.using(pyref.dropin_path(mode)) as lib:
        sim = refsim.RefSim(refsim.rpng_sim_config(**kw))
        t0 = None
        half = None
        frames = 0
        while sim.advance():
            prob = sim.pending(with_cov=False)
            sim.update_reference(prob.F)
            sim.finish()
            e, g, extra, ok = sim.state()
            t0 = e[0] if t0 is None else t0
            frames += 1
            if half is None and e[0] - t0 >= seconds / 2:
                half = (frames, sim.times())
                _laps(lib, reset=True)
            if e[0] - t0 >= seconds:
                break
        end = sim.times()
        laps = _laps(lib)
        sim.close()
        traffic = {}
        if mode.startswith("c"):
            import ctypes
            u, d = ctypes.c_long(0), ctypes.c_long(0)
            lib.ovgpu_shim_resident_cov_traffic(ctypes.byref(u), ctypes.byref(d))
            traffic = dict(cov_uploads=u.value, cov_downloads=d.value)
    n = frames - half[0]
    h = half[1]
    emit(f"time:{mode}", frames=n, features_per_update=(end["features"] - h["features"]) / n, observations_per_update=(end["observations"] - h["observations"]) / n,
         state_dim=int(extra[1]), update_ms=1e3 * (end["update_s"] - h["update_s"]) / n, propagate_and_clone_ms=1e3 * (end["propagate_s"] - h["propagate_s"]) / n,
         marginalize_ms=1e3 * (end["marginalize_s"] - h["marginalize_s"]) / n, **traffic, **({"update_laps_ms": laps} if laps else {}))


def sweep(mode):
    """Every seeded shape of the GPU parity suite through the drop-in: the 40 MSCKF updates (3-40 clones, 1-4 cameras, 1-89 features, ragged / full
    tracks, both lens models, six representations, FEJ / calibration flags, outliers), the 12 random SLAM updates and the 12 random delayed-init chains of
    tests/test_ref_build.py.  One line: the worst deviations and the cases whose accept / reject sets (or new landmark ids / anchors) differ."""
    from test_ref_build import _msckf_case
    bad, w = [], dict(msckf_dx=0.0, msckf_P=0.0, msckf_pos=0.0, slam_dx=0.0, slam_P=0.0, slam_lm=0.0, delayed_P=0.0, delayed_value=0.0)
    path = pyref.dropin_path(mode)
    for seed in range(40):
        prob, opts = _msckf_case(seed)
        ref = pyref.msckf_update(opts, capi.Views(prob))
        with pyref.using(path):
            got = pyref.msckf_update(opts, capi.Views(prob))
        if not np.array_equal(got["feat_status"], ref["feat_status"]):
            bad.append(f"msckf:{seed}")
            continue  # (another accept set is another update: its numbers say nothing)
        tri = (ref["feat_status"] == capi.FEAT_USED) | (ref["feat_status"] == capi.FEAT_CHI2_REJECTED)
        if (ref["feat_status"] == capi.FEAT_USED).any():
            w["msckf_dx"] = max(w["msckf_dx"], rel(got["dx"], ref["dx"]))
        w["msckf_P"] = max(w["msckf_P"], rel(got["P"], ref["P"]))
        if tri.any():
            w["msckf_pos"] = max(w["msckf_pos"], float(np.abs(got["p_FinG"] - ref["p_FinG"])[tri].max()))
    for seed in range(12):
        rng = np.random.default_rng(2000 + seed)
        kw = dict(C=int(rng.integers(6, 31)), K=int(rng.integers(1, 4)), track=("full", "ragged")[int(rng.integers(2))], fisheye=bool(rng.integers(2)),
                  seed=int(rng.integers(1 << 20)))
        prob = synth.make_slam_problem(2, L=int(rng.integers(1, 13)), lm_rep=int(rng.integers(0, 6)), **kw)
        opts = capi.default_options(chi2_multipler=float(rng.choice([1.0, 5.0])), do_fej=int(rng.integers(2)), do_calib_camera_pose=int(rng.integers(2)),
                                    do_calib_camera_intrinsics=int(rng.integers(2)))
        ref = pyref.slam_update(opts, capi.Views(prob))
        with pyref.using(path):
            got = pyref.slam_update(opts, capi.Views(prob))
        if not np.array_equal(got["feat_status"], ref["feat_status"]):
            bad.append(f"slam:{seed}")
            continue
        if (ref["feat_status"] == capi.FEAT_USED).any():
            w["slam_dx"], w["slam_P"] = max(w["slam_dx"], rel(got["dx"], ref["dx"])), max(w["slam_P"], rel(got["P"], ref["P"]))
        w["slam_lm"] = max(w["slam_lm"], float(np.abs(got["landmarks"] - ref["landmarks"]).max()))
    for seed in range(12):
        rng = np.random.default_rng(3000 + seed)
        kw = dict(C=int(rng.integers(6, 31)), K=int(rng.integers(1, 4)), F=int(rng.integers(1, 21)), track=("full", "ragged")[int(rng.integers(2))],
                  fisheye=bool(rng.integers(2)), seed=int(rng.integers(1 << 20)), outlier_frac=float(rng.choice([0.0, 0.3])))
        rep = int(rng.integers(0, 6))
        prob = synth.make_problem(2, **kw)
        opts = capi.default_options(chi2_multipler=float(rng.choice([1.0, 5.0])), do_fej=int(rng.integers(2)), do_calib_camera_pose=int(rng.integers(2)),
                                    do_calib_camera_intrinsics=int(rng.integers(2)))
        ref = pyref.slam_delayed_init(opts, capi.Views(prob), feat_rep=rep)
        with pyref.using(path):
            got = pyref.slam_delayed_init(opts, capi.Views(prob), feat_rep=rep)
        acc = ref["lm_cov_id"] >= 0
        anch = acc & (ref["anchor_cam"] >= 0)
        if not (np.array_equal(got["feat_status"], ref["feat_status"]) and got["N"] == ref["N"] and np.array_equal(got["lm_cov_id"], ref["lm_cov_id"])
                and np.array_equal(got["anchor_cam"][anch], ref["anchor_cam"][anch]) and np.array_equal(got["anchor_clone"][anch], ref["anchor_clone"][anch])):
            bad.append(f"delayed:{seed}")
            continue
        w["delayed_P"] = max(w["delayed_P"], rel(got["P"], ref["P"]))
        if acc.any():
            w["delayed_value"] = max(w["delayed_value"], float(np.abs(got["lm_value"][acc] - ref["lm_value"][acc]).max()))
    emit("sweep", differing=bad, **w)


CASES = [("msckf", 0), ("msckf", 3), ("msckf", 7), ("msckf", 11), ("msckf", 19), ("slam", 0), ("slam", 4), ("slam_aruco", 0), ("slam_aruco", 5), ("delayed", 0), ("delayed", 4), ("delayed", 5), ("delayed_aruco", 0), ("anchors", 2), ("anchors", 4), ("slam_mixed", 0), ("slam_mixed", 2), ("slam_mixed", 3), ("slam_mixed", 4), ("delayed_mixed", 0), ("delayed_mixed", 1),
         ("delayed_mixed", 2), ("anchors_mixed", 0), ("zupt", 0), ("zupt", 1), ("loop", 60.0), ("loop_slam", 60.0), ("loop_stereo", 60.0), ("loop_wide", 2.0)]

if __name__ == "__main__":
    mode = sys.argv[1]  # a | b (libovgpu: needs the GPU) or a_cpu | b_cpu (tests/fake_ovgpu, the oracle-backed double of the C ABI: runs anywhere)
    seconds = float(sys.argv[2]) if len(sys.argv) > 2 and sys.argv[2] not in ("sweep", "time") else 60.0
    pyref.load()
    if len(sys.argv) > 2 and sys.argv[2] == "time":
        time_loop(mode, float(sys.argv[3]) if len(sys.argv) > 3 else 6.0, **({"num_pts": int(sys.argv[4])} if len(sys.argv) > 4 else {}))
        emit("done")
        sys.stdout.flush()
        os._exit(0)
    if len(sys.argv) > 2 and sys.argv[2] == "sweep":
        sweep(mode)
        emit("done")
        sys.stdout.flush()
        os._exit(0)
    for kind, arg in CASES:
        globals()[kind](mode, seconds if kind.startswith("loop") else arg)
    emit("done")
    sys.stdout.flush()
    os._exit(0)  # (the shims keep their contexts in a function-local static: nothing to learn from the order of static destructors at exit)
