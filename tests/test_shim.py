"""CPU test of the C++ marshalling layer of the drop-in shim (open_vins_amd/shim): builds and runs its self-test."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_shim_selftest_builds_and_passes():
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "open_vins_amd", "csrc")])
    out = subprocess.check_output(["make", "-s", "-C", os.path.join(ROOT, "open_vins_amd", "shim")], text=True)
    assert "shim selftest ok" in out


def test_dropin_translation_unit_keeps_the_reference_signatures():
    src = open(os.path.join(ROOT, "open_vins_amd", "shim", "UpdaterMSCKF.cpp")).read()
    # ov_msckf/src/update/UpdaterMSCKF.h:60,68
    assert "UpdaterMSCKF::UpdaterMSCKF(UpdaterOptions &options, FeatureInitializerOptions &feat_init_options)" in src
    assert "void UpdaterMSCKF::update(std::shared_ptr<State> state, std::vector<std::shared_ptr<Feature>> &feature_vec)" in src
    assert "StateHelper::EKFUpdate(state, Hx_order_big, Hx_big, res_big, R_big)" in src
    assert "oracle" not in src


def test_feature_initializer_shim_keeps_the_class_api():
    src = open(os.path.join(ROOT, "open_vins_amd", "shim", "FeatureInitializer.cpp")).read()
    for name in ("single_triangulation", "single_triangulation_1d", "single_gaussnewton"):  # FeatureInitializer.h:100-122
        assert f"bool FeatureInitializer::{name}(std::shared_ptr<Feature> feat, std::unordered_map<size_t, std::unordered_map<double, ClonePose>> &clonesCAM)" in src
    assert "ovgpu_set_camera_poses" in src and "oracle" not in src


def test_slam_shims_keep_the_reference_signatures():
    d = os.path.join(ROOT, "open_vins_amd", "shim")
    upd = open(os.path.join(d, "UpdaterSLAM_update.cpp")).read()
    ini = open(os.path.join(d, "UpdaterSLAM_delayed_init.cpp")).read()
    # ov_msckf/src/update/UpdaterSLAM.h: update / delayed_init
    assert "void UpdaterSLAM::update(std::shared_ptr<State> state, std::vector<std::shared_ptr<Feature>> &feature_vec)" in upd
    assert "void UpdaterSLAM::delayed_init(std::shared_ptr<State> state, std::vector<std::shared_ptr<Feature>> &feature_vec)" in ini
    assert "ovgpu_slam_compress" in upd and "ovgpu_slam_delayed_init" in ini and "lv.feat_rep" in upd and "lv.feat_rep" in ini
    assert "oracle" not in upd and "oracle" not in ini


def test_mode_b_switch_of_the_update_shims():
    """-DOVGPU_SHIM_MODE_B: the device applies the update (ovgpu_msckf_update / ovgpu_slam_update) and the shim writes dx, P'
    back through StateAccess (the tail of StateHelper::EKFUpdate, StateHelper.cpp:166-196); without it the stock
    StateHelper::EKFUpdate runs on the compressed system."""
    d = os.path.join(ROOT, "open_vins_amd", "shim")
    acc = open(os.path.join(d, "ovgpu_state_access.h")).read()
    assert "s._Cov" in acc and "var->update(" in acc and "_cam_intrinsics_cameras" in acc and "do_calib_camera_intrinsics" in acc
    for name, call in (("UpdaterMSCKF.cpp", "ovgpu_msckf_update("), ("UpdaterSLAM_update.cpp", "ovgpu_slam_update(")):
        src = open(os.path.join(d, name)).read()
        assert src.count("#ifdef OVGPU_SHIM_MODE_B") == 3 and call in src and "StateAccess::apply_update(*state" in src
        assert "StateHelper::EKFUpdate(state, Hx_order_big, Hx_big, res_big, R_big)" in src  # mode A stays the default


@pytest.mark.gpu
def test_shim_drives_an_update_from_cpp():
    """The C++ side of the boundary (ovgpu_flatten.h + the C ABI, no Python in between) on the GPU."""
    subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "open_vins_amd", "shim"), "selftest"])
    out = subprocess.check_output([os.path.join(ROOT, "open_vins_amd", "shim", "selftest"), "--gpu"], text=True)
    assert "shim gpu selftest ok" in out, out
