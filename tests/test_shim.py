"""CPU tests of the C++ side of the boundary (open_vins_amd/shim).

Every drop-in translation unit goes through `g++ -std=c++17 -Wall -Werror -fsyntax-only` against tests/shim_mock, stand-ins for
Eigen and the reference's headers written from the reference's declarations (same member names, signatures and access
specifiers as the files cited in them).  Eigen and the reference tree are not on this machine, so this is the strongest check
available here: missing includes, wrong member names, const-ness and signature drift all fail it."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SHIM = os.path.join(ROOT, "open_vins_amd", "shim")
MOCK = os.path.join(ROOT, "tests", "shim_mock")

# translation unit -> modes it must compile in ("A" = unpatched reference, "B" = with the StateAccess friend line)
UNITS = {
    "UpdaterMSCKF.cpp": ("A", "B"),
    "UpdaterSLAM_update.cpp": ("A", "B"),
    "UpdaterSLAM_delayed_init.cpp": ("B",),
    "UpdaterSLAM_change_anchors.cpp": ("B",),
    "FeatureInitializer.cpp": ("A", "B"),
}


def _compile(unit, mode):
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", f"-I{MOCK}", f"-I{MOCK}/update", f"-I{MOCK}/feat",
           f"-I{ROOT}/include", f"-I{SHIM}", os.path.join(SHIM, unit)]
    if mode == "B":
        cmd.insert(1, "-DOVGPU_SHIM_MODE_B")
    return subprocess.run(cmd, capture_output=True, text=True)


@pytest.mark.parametrize("unit,mode", [(u, m) for u, ms in UNITS.items() for m in ms])
def test_dropin_unit_compiles_against_the_reference_declarations(unit, mode):
    r = _compile(unit, mode)
    assert r.returncode == 0, r.stderr[-3000:]


@pytest.mark.parametrize("unit", [u for u, ms in UNITS.items() if "A" not in ms])
def test_mode_b_only_units_need_the_friend_line(unit):
    """State::_Cov / _variables are private (State.h:182-192): the units that write them must NOT compile against the
    unpatched declaration — which also shows the stand-in State keeps the reference's access specifiers."""
    r = _compile(unit, "A")
    assert r.returncode != 0 and "private" in r.stderr


def test_helper_headers_compile_in_a_caller():
    """ovgpu_zupt.h / ovgpu_retri.h are header-only helpers called FROM reference code (UpdaterZeroVelocity::try_update,
    VioManager::retriangulate_active_tracks); tests/shim_mock/probe_helpers.cpp instantiates them the way those callers would."""
    cmd = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-DOVGPU_SHIM_MODE_B", f"-I{MOCK}", f"-I{MOCK}/update", f"-I{MOCK}/feat",
           f"-I{ROOT}/include", f"-I{SHIM}", os.path.join(MOCK, "probe_helpers.cpp")]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]


def test_resident_covariance_mode_of_the_msckf_unit_compiles():
    """-DOVGPU_SHIM_RESIDENT_COV (ovgpu_resident_cov.h): the covariance stays in the library's context; mode B only."""
    base = ["g++", "-std=c++17", "-fsyntax-only", "-Wall", "-Werror", "-DOVGPU_SHIM_RESIDENT_COV", f"-I{MOCK}", f"-I{MOCK}/update", f"-I{MOCK}/feat",
            f"-I{ROOT}/include", f"-I{SHIM}", os.path.join(SHIM, "UpdaterMSCKF.cpp")]
    r = subprocess.run(base[:1] + ["-DOVGPU_SHIM_MODE_B"] + base[1:], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    r = subprocess.run(base, capture_output=True, text=True)
    assert r.returncode != 0 and "needs OVGPU_SHIM_MODE_B" in r.stderr


def test_every_shim_source_is_covered():
    # StateHelper_resident.cpp implements the reference's WHOLE StateHelper interface (the stand-in header here declares the three functions the
    # other units call): it is compiled against the reference's own headers and run in tests/test_dropin_build.py (libov_dropin_c / _rc)
    assert sorted(f for f in os.listdir(SHIM) if f.endswith(".cpp") and f not in ("selftest.cpp", "StateHelper_resident.cpp")) == sorted(UNITS)


def test_shim_selftest_builds_and_passes():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "open_vins_amd", "csrc")])
    out = subprocess.check_output(["make", "-s", "-C", SHIM], text=True)
    assert "shim selftest ok" in out


def _src(name):
    return open(os.path.join(SHIM, name)).read()


def test_dropin_units_keep_the_reference_signatures():
    # ov_msckf/src/update/UpdaterMSCKF.h:60,68; UpdaterSLAM.h: update / delayed_init / change_anchors
    m = _src("UpdaterMSCKF.cpp")
    assert "UpdaterMSCKF::UpdaterMSCKF(UpdaterOptions &options, FeatureInitializerOptions &feat_init_options)" in m
    assert "void UpdaterMSCKF::update(std::shared_ptr<State> state, std::vector<std::shared_ptr<Feature>> &feature_vec)" in m
    assert "void UpdaterSLAM::update(std::shared_ptr<State> state, std::vector<std::shared_ptr<Feature>> &feature_vec)" in _src("UpdaterSLAM_update.cpp")
    assert "void UpdaterSLAM::delayed_init(std::shared_ptr<State> state, std::vector<std::shared_ptr<Feature>> &feature_vec)" in _src("UpdaterSLAM_delayed_init.cpp")
    assert "void UpdaterSLAM::change_anchors(std::shared_ptr<State> state)" in _src("UpdaterSLAM_change_anchors.cpp")
    f = _src("FeatureInitializer.cpp")
    for name in ("single_triangulation", "single_triangulation_1d", "single_gaussnewton"):  # FeatureInitializer.h:100-122
        assert f"bool FeatureInitializer::{name}(std::shared_ptr<Feature> feat, std::unordered_map<size_t, std::unordered_map<double, ClonePose>> &clonesCAM)" in f


def test_no_shim_source_touches_the_oracle_or_the_environment():
    for name in list(UNITS) + ["ovgpu_shim_common.h", "ovgpu_flatten.h", "ovgpu_state_access.h", "ovgpu_zupt.h", "ovgpu_retri.h", "ovgpu_resident_cov.h", "StateHelper_resident.cpp"]:
        s = _src(name)
        assert "oracle" not in s and "getenv" not in s, name


def test_mode_a_triangulates_once_and_resyncs_options():
    """The compress call triangulates on the device; the Feature side effects come from ovgpu_get_triangulation, not from a second
    ovgpu_triangulate pass.  Options are re-read on every call (the context cache is keyed by their values)."""
    m = _src("UpdaterMSCKF.cpp")
    assert "ovgpu_get_triangulation(" in m and "ovgpu_triangulate(" not in m
    c = _src("ovgpu_shim_common.h")
    assert "context_for(" in c and "memcmp" in c
    for name in ("UpdaterMSCKF.cpp", "UpdaterSLAM_update.cpp", "UpdaterSLAM_delayed_init.cpp", "UpdaterSLAM_change_anchors.cpp"):
        assert "context_for(ovgpu_shim::make_options(" in _src(name), name
    assert '#include "cam/CamEqui.h"' in c


def test_mode_b_switch_of_the_update_shims():
    """-DOVGPU_SHIM_MODE_B: the device applies the update (ovgpu_msckf_update / ovgpu_slam_update) and the shim writes dx, P'
    back through StateAccess (the tail of StateHelper::EKFUpdate, StateHelper.cpp:166-196); without it the stock
    StateHelper::EKFUpdate runs on the compressed system."""
    acc = _src("ovgpu_state_access.h")
    assert "s._Cov" in acc and "var->update(" in acc and "_cam_intrinsics_cameras" in acc and "do_calib_camera_intrinsics" in acc
    for name, call in (("UpdaterMSCKF.cpp", "ovgpu_msckf_update("), ("UpdaterSLAM_update.cpp", "ovgpu_slam_update(")):
        src = _src(name)
        assert call in src and "StateAccess::apply_update(*state" in src and "ekf_update_with(" in src
    assert "StateHelper::EKFUpdate(state, Hx_order_big, Hx_big, res_big, R_big)" in _src("ovgpu_shim_common.h")  # mode A


def test_delayed_init_refreshes_the_camera_objects_and_reports_the_anchor():
    s = _src("UpdaterSLAM_delayed_init.cpp")
    assert "StateAccess::refresh_cameras(*state)" in s  # StateHelper.cpp:191-196
    assert "landmark->_unique_camera_id = (*it)->anchor_cam_id" in s and "write_triangulation(" in s  # UpdaterSLAM.cpp:214


@pytest.mark.gpu
def test_shim_drives_an_update_from_cpp():
    """The C++ side of the boundary (ovgpu_flatten.h + the C ABI, no Python in between) on the GPU."""
    subprocess.check_call(["make", "-s", "-C", SHIM, "selftest"])
    out = subprocess.check_output([os.path.join(SHIM, "selftest"), "--gpu"], text=True)
    assert "shim gpu selftest ok" in out, out


def test_slam_update_required_meas_rule(tmp_path):
    """UpdaterSLAM.cpp:283-295: a landmark without measurements is flagged and erased; an ANCHORED_INVERSE_DEPTH_SINGLE landmark
    with exactly ONE measurement is erased from this update WITHOUT to_delete (FeatureDatabase::cleanup must keep the measurement,
    the next frame brings the second one).  The shim decides that before the library sees the track — so the drop-in translation
    unit can be RUN against the stand-in headers (tests/shim_mock/run_required_meas.cpp): three such tracks, two passes (the
    second representation), and the library is never reached."""
    exe = str(tmp_path / "run_required_meas")
    cmd = ["g++", "-std=c++17", "-O1", "-Wall", "-Werror", f"-I{MOCK}", f"-I{MOCK}/update", f"-I{MOCK}/feat", f"-I{ROOT}/include", f"-I{SHIM}",
           os.path.join(MOCK, "run_required_meas.cpp"), os.path.join(SHIM, "UpdaterSLAM_update.cpp"), "-o", exe,
           f"-L{ROOT}/open_vins_amd/csrc", "-lovgpu", f"-Wl,-rpath,{ROOT}/open_vins_amd/csrc"]
    r = subprocess.run(cmd, capture_output=True, text=True)
    assert r.returncode == 0, r.stderr[-3000:]
    p = subprocess.run([exe], capture_output=True, text=True, timeout=60)
    assert p.returncode == 0, p.stdout + p.stderr
    lines = p.stdout.split("\n")
    assert "left in feature_vec: 0" in lines                      # all three left the update ...
    assert "feature 10 to_delete 1" in lines                      # ... no measurement: flagged
    assert "feature 11 to_delete 0" in lines                      # ... single depth with ONE measurement: kept for the next frame
    assert "feature 12 to_delete 1" in lines                      # ... the second pass (another representation) applies the same rule
    assert "EKFUpdate reached" not in p.stdout


@pytest.mark.gpu
def test_shim_timing_mode_reports_the_drop_in_cost():
    """`selftest --time`: reference-shaped tracks -> flatten -> upload -> mode A / mode B, host to host, as one JSON line (what bench.py
    embeds as `shim`)."""
    import json
    import subprocess
    exe = os.path.join(ROOT, "open_vins_amd", "shim", "selftest")
    p = subprocess.run([exe, "--time", "300", "3"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    d = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert d["features"] == 300 and d["clones"] == 30 and d["cameras"] == 2 and d["features_used"] > 200
    assert d["measurements"] < d["observations_incl_stale"]  # the stale observation of every track was cleaned away
    for k in ("flatten_ms", "upload_ms", "mode_a_call_ms", "mode_b_call_ms", "shim_mode_a_ms", "shim_mode_b_ms"):
        assert d[k] > 0
    assert abs(d["shim_mode_a_ms"] - (d["flatten_ms"] + d["upload_ms"] + d["mode_a_call_ms"])) < 1e-3
    # ... and the resident form: tracks appended frame by frame, the batch assembled on the device
    p = subprocess.run([exe, "--time-resident", "300", "2"], capture_output=True, text=True, timeout=300)
    assert p.returncode == 0, p.stdout + p.stderr
    r = json.loads([l for l in p.stdout.splitlines() if l.startswith("{")][-1])
    assert r["features"] == 300 and r["features_used"] == d["features_used"]  # the device-assembled batch gates like the flattened one
    assert r["observations_newest_frame"] > 300 and r["resident_mode_b_ms"] > 0
