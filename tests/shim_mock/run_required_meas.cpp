// Runs the drop-in UpdaterSLAM::update (open_vins_amd/shim/UpdaterSLAM_update.cpp, compiled against the stand-in headers of this
// directory) on tracks that must leave the update BEFORE the library is reached (UpdaterSLAM.cpp:283-295), and prints what happened
// to them.  TEST INFRASTRUCTURE (tests/test_shim.py::test_slam_update_required_meas_rule): no GPU, no call into libovgpu.so.
#include <cstdio>
#include <memory>
#include <vector>

#include "UpdaterSLAM.h"
#include "cam/CamRadtan.h"
#include "feat/Feature.h"
#include "state/State.h"
#include "state/StateHelper.h"

using namespace ov_msckf;
using namespace ov_type;
using namespace ov_core;

// the pieces of the reference this translation unit would link against
UpdaterSLAM::UpdaterSLAM(UpdaterOptions &a, UpdaterOptions &b, FeatureInitializerOptions &) : _options_slam(a), _options_aruco(b) {}
Eigen::MatrixXd StateHelper::get_full_covariance(std::shared_ptr<State>) { return Eigen::MatrixXd(18, 18); }
void StateHelper::EKFUpdate(std::shared_ptr<State>, const std::vector<std::shared_ptr<Type>> &, const Eigen::MatrixXd &, const Eigen::VectorXd &, const Eigen::MatrixXd &) {
  std::printf("EKFUpdate reached\n");
}

struct TestCam : CamRadtan { // (the stand-in camera classes carry no model)
  Eigen::Vector2f undistort_f(const Eigen::Vector2f &uv) override { return uv; }
};

static std::shared_ptr<Feature> track(size_t id, int n_obs, double t0) {
  auto f = std::make_shared<Feature>();
  f->featid = id;
  for (int i = 0; i < n_obs; i++) {
    Eigen::VectorXf uv(2);
    uv(0) = 100.f, uv(1) = 120.f;
    f->uvs[0].push_back(uv), f->uvs_norm[0].push_back(uv), f->timestamps[0].push_back(t0 + 0.1 * i);
  }
  return f;
}

int main() {
  auto state = std::make_shared<State>();
  for (int i = 0; i < 2; i++) {
    auto c = std::make_shared<PoseJPL>();
    c->set_local_id(6 * i);
    state->_clones_IMU[1.0 + 0.1 * i] = c;
  }
  auto calib = std::make_shared<PoseJPL>();
  calib->set_local_id(12);
  state->_calib_IMUtoCAM[0] = calib;
  auto intr = std::make_shared<Vec>(8);
  intr->set_value(Eigen::MatrixXd(8, 1));
  state->_cam_intrinsics[0] = intr;
  state->_cam_intrinsics_cameras[0] = std::make_shared<TestCam>();
  state->_options.feat_rep_slam = LandmarkRepresentation::Representation::ANCHORED_INVERSE_DEPTH_SINGLE;
  state->_options.feat_rep_aruco = LandmarkRepresentation::Representation::GLOBAL_3D;
  state->_options.max_aruco_features = 0;
  auto lm = [&](size_t id, LandmarkRepresentation::Representation rep) {
    auto l = std::make_shared<Landmark>(rep == LandmarkRepresentation::Representation::ANCHORED_INVERSE_DEPTH_SINGLE ? 1 : 3);
    l->_featid = id, l->_feat_representation = rep;
    state->_features_SLAM[id] = l;
  };
  lm(10, LandmarkRepresentation::Representation::ANCHORED_INVERSE_DEPTH_SINGLE); // no measurement at all
  lm(11, LandmarkRepresentation::Representation::ANCHORED_INVERSE_DEPTH_SINGLE); // exactly one: needs two
  lm(12, LandmarkRepresentation::Representation::GLOBAL_3D);                     // another representation (second pass), no measurement
  std::vector<std::shared_ptr<Feature>> vec = {track(10, 0, 1.0), track(11, 1, 1.0), track(12, 0, 1.0)};
  const std::vector<std::shared_ptr<Feature>> all = vec;
  UpdaterOptions os, oa;
  FeatureInitializerOptions fo;
  UpdaterSLAM upd(os, oa, fo);
  upd.update(state, vec);
  std::printf("left in feature_vec: %zu\n", vec.size());
  for (const auto &f : all) std::printf("feature %zu to_delete %d\n", f->featid, f->to_delete ? 1 : 0);
  return 0;
}
