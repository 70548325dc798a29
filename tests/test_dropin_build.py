"""The drop-in translation units of open_vins_amd/shim INSIDE the reference tree (round 4; VERDICT r3 row (b): "real-tree build
unverified").

Eigen, Boost and OpenCV are not on this machine, but oracle/ref/standin restates what the reference's update path needs of them — enough to
compile the reference's OWN sources (oracle/_ref/libov_ref.so, tests/test_ref_build.py).  The same stand-ins let the shim units be
compiled against the reference's own HEADERS (not the hand-written stand-ins of tests/shim_mock) and LINKED with the reference's own
State / StateHelper / Propagator / types objects in place of ov_msckf/src/update/UpdaterMSCKF.cpp, ov_core/src/feat/FeatureInitializer.cpp
and the three member functions of UpdaterSLAM.cpp the shim redefines: oracle/_ref/libov_dropin_a.so (mode A) and _b.so (mode B), behind
the SAME C driver as the reference build.

CPU legs: every unit compiles in the tree in both modes (the first such build found a real defect: shim/FeatureInitializer.cpp used
ov_core::Feature through FeatureInitializer.h's forward declaration only); the two mode-B-only units need the friend line; the libraries
link, load, bind the C-ABI entry points of their mode, carry the SHIM's definitions of the replaced functions, and run the reference's
driver up to the shim's own early returns; and — linked against tests/fake_ovgpu, a test double of the C ABI served by the CPU oracle
(oracle/_ref/libov_dropin_{a,b}_cpu.so) — the shim's C++ runs END TO END here: UpdaterMSCKF::update, UpdaterSLAM::update / delayed_init /
change_anchors and the rpng_sim closed loop with the shim as the reference filter's updater, against the reference's own updaters
(tests/dropin_probe.py in a subprocess).  GPU legs: the same probe through the libraries linked against libovgpu."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from open_vins_amd import capi, synth
from oracle import pyref

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "open_vins_amd", "shim")
REF = pyref.REFERENCE
UNITS = {"UpdaterMSCKF": ("A", "B"), "UpdaterSLAM_update": ("A", "B"), "FeatureInitializer": ("A", "B"), "UpdaterSLAM_delayed_init": ("B",),
         "UpdaterSLAM_change_anchors": ("B",)}
in_tree = pytest.mark.skipif(not pyref.can_build(), reason="/root/reference is not on this machine")


def _compile(unit, mode, friend):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-w", f"-I{ROOT}/oracle/ref/standin", f"-I{REF}/ov_core/src", f"-I{REF}/ov_msckf/src",
           f"-I{REF}/ov_msckf/src/update", f"-I{REF}/ov_core/src/feat", f"-I{ROOT}/include", f"-I{SHIM}", os.path.join(SHIM, unit + ".cpp")]
    if mode == "B":
        cmd.insert(1, "-DOVGPU_SHIM_MODE_B")
    if friend:  # INTEGRATION.md's `friend class ovgpu_shim::StateAccess;` in State.h, emulated: the tree is not ours to patch
        cmd[1:1] = ["-Dprivate=public", "-Dprotected=public"]
    return subprocess.run(cmd, capture_output=True, text=True)


@in_tree
@pytest.mark.parametrize("unit,mode", [(u, m) for u, ms in UNITS.items() for m in ms])
def test_unit_compiles_against_the_reference_headers(unit, mode):
    r = _compile(unit, mode, friend=(mode == "B"))
    assert r.returncode == 0, r.stderr[-3000:]


@in_tree
@pytest.mark.parametrize("unit", [u for u, ms in UNITS.items() if "A" not in ms])
def test_mode_b_only_units_need_the_friend_line_in_the_real_state_h(unit):
    r = _compile(unit, "B", friend=False)
    assert r.returncode != 0 and "private within this context" in r.stderr and "_Cov" in r.stderr


@in_tree
def test_helper_headers_compile_against_the_reference_headers():
    """ovgpu_zupt.h / ovgpu_retri.h (called FROM UpdaterZeroVelocity::try_update / VioManager::retriangulate_active_tracks), instantiated
    by tests/shim_mock/probe_helpers.cpp the way those callers would — here against the reference's own State / Type / options headers."""
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-w", "-DOVGPU_SHIM_MODE_B", "-Dprivate=public", "-Dprotected=public", f"-I{ROOT}/oracle/ref/standin",
           f"-I{REF}/ov_core/src", f"-I{REF}/ov_msckf/src", f"-I{REF}/ov_msckf/src/update", f"-I{REF}/ov_core/src/feat", f"-I{ROOT}/include", f"-I{SHIM}",
           os.path.join(ROOT, "tests", "shim_mock", "probe_helpers.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def _nm(path, *flags):
    return subprocess.check_output(["nm", "-C", *flags, path], text=True)


@pytest.fixture(scope="module")
def dropin_libs():
    if pyref.can_build():
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "open_vins_amd", "csrc")])
        pyref.build()
        pyref.build_dropin()
    paths = {m: pyref.dropin_path(m) for m in ("a", "b")}
    if not all(os.path.exists(p) for p in paths.values()):
        pytest.skip("oracle/_ref/libov_dropin_*.so are not here and cannot be built (no /root/reference)")
    return paths


def test_dropin_libraries_link_the_shim_into_the_reference_objects(dropin_libs):
    for mode, path in dropin_libs.items():
        und = _nm(path, "-D", "--undefined-only")
        want = ("ovgpu_msckf_compress", "ovgpu_slam_compress") if mode == "a" else ("ovgpu_msckf_update", "ovgpu_slam_update")
        other = ("ovgpu_msckf_update", "ovgpu_slam_update") if mode == "a" else ("ovgpu_msckf_compress", "ovgpu_slam_compress")
        for s in want + ("ovgpu_set_state", "ovgpu_set_features", "ovgpu_get_triangulation", "ovgpu_slam_delayed_init", "ovgpu_slam_change_anchors"):
            assert f"U {s}\n" in und, (mode, s)
        for s in other:
            assert f"U {s}\n" not in und, (mode, s)
        # UpdaterZeroVelocity::try_update: mode B carries the reference's own file with INTEGRATION.md's patch applied at build time
        # (oracle/ref/patch_zupt.py: its covariance work goes through shim/ovgpu_zupt.h), mode A the unpatched object
        for s in ("ovgpu_state_marginal_covariance", "ovgpu_state_propagate", "ovgpu_ekf_update"):
            assert (f"U {s}\n" in und) == (mode == "b"), (mode, s)
        syms = _nm(path)
        # the replaced member functions are the SHIM's (its persistent flattening buffer is a function-local static of update()) ...
        assert "guard variable for ov_msckf::UpdaterMSCKF::update(" in syms or "ov_msckf::UpdaterMSCKF::update(std::shared_ptr<ov_msckf::State>, std::vector<std::shared_ptr<ov_core::Feature>" in syms
        assert "ovgpu_shim::" in syms
        # ... and the reference's own classes around them are still there
        for s in ("ov_msckf::StateHelper::EKFUpdate(", "ov_msckf::StateHelper::marginalize(", "ov_msckf::UpdaterSLAM::perform_anchor_change(",
                  "ov_msckf::UpdaterSLAM::UpdaterSLAM(", "ov_msckf::Propagator::", "ov_core::FeatureDatabase::update_feature("):
            assert s in syms, (mode, s)
    ref_syms = _nm(pyref.LIB_PATH)
    assert "ovgpu_shim::" not in ref_syms and "ovgpu_" not in _nm(pyref.LIB_PATH, "-D", "--undefined-only")


def test_dropin_library_runs_the_reference_driver_up_to_the_shims_early_return(dropin_libs):
    """No GPU here: a batch whose tracks all fall under two observations never reaches the library (UpdaterMSCKF.cpp:88-92, the shim's
    clean + flatten loop erases them and returns), so the reference's State construction, the shim's State snapshot, its track flattening
    and its side effects on the Feature objects run on the CPU — and must leave the state exactly as the reference's own updater leaves it."""
    prob = synth.make_problem(2, F=6, C=8)
    keep = np.zeros(prob.M, bool)
    keep[prob.meas_offsets[:-1]] = True  # ONE observation per track
    short = prob.subset(np.arange(prob.F))
    offs = np.arange(prob.F + 1, dtype=np.int32)
    short.meas_offsets = offs
    short.uv, short.uvn = prob.uv.reshape(-1, 2)[keep].reshape(-1).copy(), prob.uvn.reshape(-1, 2)[keep].reshape(-1).copy()
    short.clone_idx, short.cam_idx = prob.clone_idx[keep].copy(), prob.cam_idx[keep].copy()
    opts = capi.default_options(chi2_multipler=1.0)
    ref = pyref.msckf_update(opts, capi.Views(short))
    for mode, path in dropin_libs.items():
        with pyref.using(path):
            got = pyref.msckf_update(opts, capi.Views(short))
        assert np.array_equal(got["feat_status"], ref["feat_status"]) and (got["feat_status"] != capi.FEAT_USED).all(), mode
        assert np.array_equal(got["P"], ref["P"]) and np.array_equal(got["clone_q_p"], ref["clone_q_p"]) and np.array_equal(got["dx"], ref["dx"]), mode


# what the drop-in library has to reproduce of the reference's own updaters (tests/dropin_probe.py's keys): the tolerances of the
# GPU-against-reference fixtures (tests/test_ref_fixtures.py)
LIMITS = {"msckf": dict(pos=1e-8, dx=1e-7, P=1e-8, state=1e-9), "slam": dict(dx=1e-7, P=1e-8, landmarks=1e-9, state=1e-9),
          "delayed": dict(value=1e-8, P=1e-7, state=1e-9), "anchors": dict(P=1e-11, value=1e-11, fej=1e-11), "zupt": dict(dx=1e-7, P=1e-8, state=1e-9)}


def _run_probe(mode, seconds):
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_probe.py"), mode, str(seconds)], capture_output=True, text=True, timeout=900)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    seen = [l["case"] for l in lines]
    assert p.returncode == 0 and seen and seen[-1] == "done", (p.returncode, seen, p.stderr[-2000:])
    return lines[:-1]


@pytest.fixture(scope="module")
def cpu_probes(dropin_libs):
    """The oracle-backed builds run side by side (they share nothing): module-scoped, so the CPU suite pays the longest of them once."""
    if pyref.can_build():
        pyref.build_dropin("dropin_cpu")
    modes = [m for m in ("a_cpu", "b_cpu", "c_cpu") if os.path.exists(pyref.dropin_path(m))]
    procs = {m: subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dropin_probe.py"), m, "20"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True)
             for m in modes}
    for m in modes:  # ... and, beside them, the sweep over every seeded shape of the parity suite (modes A and B)
        procs["sweep_" + m] = subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "dropin_probe.py"), m, "sweep"], stdout=subprocess.PIPE,
                                               stderr=subprocess.PIPE, text=True)
    out = {}
    for m, pr in procs.items():
        so, se = pr.communicate(timeout=900)
        lines = [json.loads(l) for l in so.splitlines() if l.startswith("{")]
        out[m] = (pr.returncode, lines, se)
    return out


def _cpu_lines(cpu_probes, mode):
    if mode not in cpu_probes:
        pytest.skip(f"oracle/_ref/libov_dropin_{mode}.so is not here and cannot be built (no /root/reference)")
    rc, lines, err = cpu_probes[mode]
    seen = [l["case"] for l in lines]
    assert rc == 0 and seen and seen[-1] == "done", (rc, seen, err[-2000:])
    return lines[:-1]


def _judge(lines, limits, min_updates, loop_only=False):
    bad = []
    for l in lines:
        kind = l["case"].split(":")[0]
        if not l["status_equal"]:
            bad.append((l["case"], "accept / reject sets differ"))
        if kind.startswith("loop"):  # "loop": the MSCKF-only filter of BASELINE configs[0]; "loop_slam": the same with SLAM landmarks in the state
            # the yardsticks of tests/test_rpng_sim_loop.py: same decisions over the first hundred updates (status_equal) and on >= 99 % of all,
            # estimates close over the first ten updates and within 5 x the reference's own control run (two reference runs 1e-13 m apart at
            # the start), the same ATE
            print("drop-in closed loop:", l)
            if kind == "loop_wide":  # two seconds, but hundreds of tracks per update: the threaded flattening and the one-pass compaction of the shim
                if not (l["updates"] == l["updates_reference"] >= 20 and l["decisions"] >= 256 * l["updates"] and l["differing"] == 0 and l["sep"] < 5 * l["control"]):
                    bad.append(l)
                continue
            if kind in ("loop_slam", "loop_stereo") and not (l["state_dim_max"] == l["state_dim_max_reference"] >= 126 + 3 * 10):
                bad.append((l["case"], "landmarks in the state", l["state_dim_max"], l["state_dim_max_reference"]))
            ok = (l["updates"] >= min_updates and l["updates"] == l["updates_reference"] and l["differing"] <= 0.01 * l["decisions"]
                  and l["sep_first_ten"] < limits["loop_first_ten"] and l["sep"] < 5 * l["control"]
                  and abs(l["ate_deg"] - l["ate_deg_reference"]) < 1e-4 and abs(l["ate_m"] - l["ate_m_reference"]) < 1e-5)
            if not ok:
                bad.append(l)
            continue
        for k, lim in limits[kind].items():
            if not (0.0 <= l[k] < lim):
                bad.append((l["case"], k, l[k], lim))
    assert not bad, bad
    for kind in ("loop:", "loop_slam:", "loop_stereo:"):
        assert any(l["case"].startswith(kind) for l in lines), kind
    if not loop_only:
        assert any(l["case"].startswith("loop_wide:") for l in lines)
        assert sum(l.get("used", 0) for l in lines) > 50 and any(l["case"].startswith("delayed") and l["accepted"] >= 4 for l in lines)


# the C ABI served by the CPU oracle: what differs from the reference is the oracle's arithmetic (tests/test_ref_build.py: 1e-12)
LIMITS_CPU = {"msckf": dict(pos=1e-10, dx=1e-10, P=1e-11, state=1e-10), "slam": dict(dx=1e-10, P=1e-11, landmarks=1e-10, state=1e-10),
              "delayed": dict(value=1e-10, P=1e-10, state=1e-10), "anchors": dict(P=1e-13, value=1e-13, fej=1e-13), "zupt": dict(dx=1e-10, P=1e-11, state=1e-11), "loop_first_ten": 1e-10}


@pytest.mark.parametrize("mode", ["a_cpu", "b_cpu"])
def test_dropin_library_equals_the_reference_updaters_with_the_oracle_behind_the_abi(cpu_probes, mode):
    """The shim's C++ END TO END on this machine: the drop-in library linked against tests/fake_ovgpu (include/ovgpu.h's entry points served by
    the CPU oracle) instead of libovgpu.  UpdaterMSCKF::update on five seeded batches (six representations, calibration / FEJ flags, outliers),
    UpdaterSLAM::update, delayed_init chains, change_anchors, and 20 s of the rpng_sim closed loop with the shim as the reference filter's
    updater (MSCKF-only as BASELINE configs[0], and with SLAM landmarks in the state: VioManager's landmark handling around the shim's UpdaterSLAM) — each against the reference's own updaters on identical reference `State`s: identical accept / reject sets with the rejecting
    stage (which also runs the FeatureInitializer shim against the reference's), dx / P' / landmarks at the oracle's agreement with the
    reference, the closed loop inside the reference's own control run."""
    _judge(_cpu_lines(cpu_probes, mode), LIMITS_CPU, 190)


@pytest.mark.parametrize("mode", ["a_cpu", "b_cpu"])
def test_dropin_library_on_every_seeded_shape_of_the_parity_suite(cpu_probes, mode):
    """The 40 MSCKF updates, 12 SLAM updates and 12 delayed-initialisation chains of tests/test_ref_build.py (3-40 clones, 1-4 cameras, both lens
    models, six representations, every flag combination, outliers) through the shim: the same accept / reject sets, landmark ids and anchors as the
    reference's updaters on every one, the numbers at the oracle's agreement with the reference."""
    (l,) = _cpu_lines(cpu_probes, "sweep_" + mode)
    assert l["case"] == "sweep" and l["differing"] == [], l
    assert l["msckf_dx"] < 1e-10 and l["msckf_P"] < 1e-11 and l["msckf_pos"] < 1e-9 and l["slam_dx"] < 1e-10 and l["slam_P"] < 1e-11 and l["slam_lm"] < 1e-10
    assert l["delayed_P"] < 1e-10 and l["delayed_value"] < 1e-9


def _judge_resident_covariance(lines):
    """What the mode is for: between two MSCKF updates nothing moves the N x N covariance across the bus.  The MSCKF-only loop uploads it once
    (the filter's initial covariance) and never reads it back: propagation, cloning, marginalisation and the update itself all happen on the
    device's copy (shim/StateHelper_resident.cpp).  With SLAM landmarks the SLAM units still work on the host's copy (their mode-B write-backs
    mark it newer), so those loops copy about once per frame each way — counted, not hidden."""
    (l,) = [l for l in lines if l["case"].startswith("loop:")]
    assert l["cov_uploads"] <= 2 and l["cov_downloads"] == 0, l
    for l in lines:
        if l["case"].startswith(("loop_slam:", "loop_stereo:")):
            assert 0 < l["cov_uploads"] <= l["updates"] + 2 and l["cov_downloads"] <= l["updates"] + 2, l


def test_resident_covariance_mode_of_the_dropin_equals_the_reference_with_the_oracle_behind_the_abi(cpu_probes):
    """-DOVGPU_SHIM_RESIDENT_COV (shim/ovgpu_resident_cov.h + shim/StateHelper_resident.cpp): State::_Cov lives in the
    library's context between frames.  The reference's StateHelper.cpp is compiled under another class name (-DStateHelper=StateHelperHost, the
    file is not edited) and the shim's StateHelper stands in its place: EKFPropagation, augment_clone, marginalize and get_marginal_covariance
    act on the device's copy through ovgpu_state_*, everything else brings the covariance back first and runs the reference's own code.
    Every per-call case and the closed loops, against the reference's updaters, exactly as for mode B."""
    lines = _cpu_lines(cpu_probes, "c_cpu")
    _judge(lines, LIMITS_CPU, 190)
    _judge_resident_covariance(lines)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["a", "b", "c"])
def test_dropin_library_equals_the_reference_updaters_on_the_gpu(dropin_libs, mode):
    if not os.path.exists(pyref.dropin_path(mode)):
        pytest.skip("drop-in library of this mode is not here")
    lines = _run_probe(mode, 6.0)  # (6 s of each closed loop = 60 updates; the CPU legs run 20 s, the GPU suite has a time budget)
    _judge(lines, dict(LIMITS, loop_first_ten=1e-8), 50)
    if mode == "c":  # the covariance stays on the device from frame to frame
        _judge_resident_covariance(lines)


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["a", "b"])
def test_dropin_library_on_every_seeded_shape_of_the_parity_suite_on_the_gpu(dropin_libs, mode):
    """The same sweep (40 MSCKF updates, 12 SLAM updates, 12 delayed-initialisation chains) through the drop-in library linked against libovgpu."""
    p = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "dropin_probe.py"), mode, "sweep"], capture_output=True, text=True, timeout=900)
    lines = [json.loads(l) for l in p.stdout.splitlines() if l.startswith("{")]
    assert p.returncode == 0 and len(lines) == 2 and lines[-1]["case"] == "done", (p.returncode, p.stdout[-500:], p.stderr[-2000:])
    l = lines[0]
    print("drop-in sweep on the GPU:", l)
    # a gate decision within round-off of its threshold may differ between two float64 implementations (tests/parity_util.py): at most one such case
    assert len(l["differing"]) <= 1, l
    assert l["msckf_dx"] < 1e-7 and l["msckf_P"] < 1e-8 and l["msckf_pos"] < 1e-8 and l["slam_dx"] < 1e-7 and l["slam_P"] < 1e-8 and l["slam_lm"] < 1e-9
    assert l["delayed_P"] < 1e-7 and l["delayed_value"] < 1e-8
