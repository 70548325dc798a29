// Drop-in replacement for the body of ov_msckf::UpdaterSLAM::update (ov_msckf/src/update/UpdaterSLAM.cpp:253-479,
// rpng/open_vins v2.7).  Delete that definition from UpdaterSLAM.cpp and compile this file next to it (the class
// declaration, the constructor, delayed_init, change_anchors and perform_anchor_change stay the reference's).
// Mode A, like the MSCKF shim: the GPU builds, gates, stacks and compresses the system, the stock
// StateHelper::EKFUpdate applies it.  Landmark representations: all six (LandmarkRepresentation.h:38-46), every landmark of a
// call in the same one (StateOptions::feat_rep_slam); ArUco tags with their own options stay on the reference's CPU path.
#include "UpdaterSLAM.h"

#include "feat/Feature.h"
#include "state/State.h"
#include "state/StateHelper.h"
#include "types/Landmark.h"
#include "types/LandmarkRepresentation.h"

#include "ovgpu.h"
#include "ovgpu_flatten.h"
#ifdef OVGPU_SHIM_MODE_B
#include "ovgpu_state_access.h" // needs `friend struct ovgpu_shim::StateAccess;` in State.h
#endif

using namespace ov_core;
using namespace ov_type;
using namespace ov_msckf;

namespace {
std::unique_ptr<ovgpu_shim::Context> g_slam_ctx;
}

void UpdaterSLAM::update(std::shared_ptr<State> state, std::vector<std::shared_ptr<Feature>> &feature_vec) {
  if (feature_vec.empty()) return; // :256-257

  // ---- state snapshot (same flattening as the MSCKF shim)
  ovgpu_shim::FlatState fs;
  std::vector<std::shared_ptr<Type>> var_of_cov;
  for (const auto &c : state->_clones_IMU) {
    const Eigen::Vector4d q = c.second->quat(), qf = c.second->quat_fej();
    const Eigen::Vector3d p = c.second->pos(), pf = c.second->pos_fej();
    fs.add_clone(c.first, q.data(), p.data(), qf.data(), pf.data(), c.second->id());
    var_of_cov.push_back(c.second);
  }
  std::vector<size_t> cam_ids;
  for (const auto &c : state->_calib_IMUtoCAM) cam_ids.push_back(c.first);
  std::sort(cam_ids.begin(), cam_ids.end());
  std::unordered_map<size_t, int> cam_index;
  for (size_t k = 0; k < cam_ids.size(); k++) {
    const size_t id = cam_ids[k];
    cam_index[id] = (int)k;
    const auto &pose = state->_calib_IMUtoCAM.at(id);
    const Eigen::Vector4d q = pose->quat();
    const Eigen::Vector3d p = pose->pos();
    const Eigen::Matrix<double, 8, 1> intr = state->_cam_intrinsics.at(id)->value();
    const bool fisheye = std::dynamic_pointer_cast<CamEqui>(state->_cam_intrinsics_cameras.at(id)) != nullptr;
    fs.add_camera(q.data(), p.data(), intr.data(), fisheye, state->_options.do_calib_camera_pose ? pose->id() : -1,
                  state->_options.do_calib_camera_intrinsics ? state->_cam_intrinsics.at(id)->id() : -1);
    var_of_cov.push_back(pose), var_of_cov.push_back(state->_cam_intrinsics.at(id));
  }
  const Eigen::MatrixXd P = StateHelper::get_full_covariance(state);
  fs.N = (int32_t)P.rows();
  fs.P.assign(P.data(), P.data() + P.size());

  // ---- 1. clean the tracks (UpdaterSLAM.cpp:266-296), flatten them and their landmarks
  const ovgpu_shim::CloneIndex clones(fs.clone_times);
  ovgpu_shim::FlatFeatures ff;
  std::vector<double> lm_value, lm_fej;
  std::vector<int32_t> lm_cov, lm_index, lm_anchor_cam, lm_anchor_clone;
  std::vector<double> f_sigma, f_mult; // per-feature options: ArUco corners use _options_aruco (:392-394, :408-409)
  bool any_aruco = false;
  const auto rep = state->_options.feat_rep_slam;
  const bool single = rep == LandmarkRepresentation::Representation::ANCHORED_INVERSE_DEPTH_SINGLE;
  auto it0 = feature_vec.begin();
  while (it0 != feature_vec.end()) {
    (*it0)->clean_old_measurements(fs.clone_times);
    int ct_meas = 0;
    for (const auto &pair : (*it0)->timestamps) ct_meas += (int)pair.second.size();
    std::shared_ptr<Landmark> landmark = state->_features_SLAM.at((*it0)->featid);
    if (landmark->_feat_representation != rep)
      throw std::runtime_error("ovgpu SLAM shim: every landmark of a call must use StateOptions::feat_rep_slam");
    if (ct_meas < 1) { // :289-291
      (*it0)->to_delete = true;
      it0 = feature_vec.erase(it0);
      continue;
    }
    Feature &f = **it0;
    for (const auto &pair : f.timestamps) {
      const auto &uvs = f.uvs.at(pair.first), &uvn = f.uvs_norm.at(pair.first);
      ff.add_camera(cam_index.at(pair.first), pair.second, [&](size_t i, float &a, float &b) { a = uvs[i](0), b = uvs[i](1); },
                    [&](size_t i, float &a, float &b) { a = uvn[i](0), b = uvn[i](1); }, clones);
    }
    ff.end_feature();
    // the library holds Landmark::value() / fej() — representation coordinates — and applies get_xyz itself (:345-353)
    Eigen::Vector3d v, vf;
    if (single) { // 1-dof landmark: constant bearing + inverse depth (Landmark.cpp:57-60, :124-140)
      v << landmark->uv_norm_zero(0), landmark->uv_norm_zero(1), landmark->value()(0);
      vf << landmark->uv_norm_zero_fej(0), landmark->uv_norm_zero_fej(1), landmark->fej()(0);
    } else {
      v = landmark->value(), vf = landmark->fej();
    }
    const bool relative = LandmarkRepresentation::is_relative_representation(rep);
    lm_anchor_cam.push_back(relative ? cam_index.at(landmark->_anchor_cam_id) : -1);
    lm_anchor_clone.push_back(relative ? clones.find(landmark->_anchor_clone_timestamp) : -1);
    lm_index.push_back((int32_t)lm_cov.size());
    lm_cov.push_back(landmark->id());
    lm_value.insert(lm_value.end(), v.data(), v.data() + 3), lm_fej.insert(lm_fej.end(), vf.data(), vf.data() + 3);
    var_of_cov.push_back(landmark);
    const bool is_aruco = (int)f.featid < state->_options.max_aruco_features; // :392
    any_aruco |= is_aruco;
    f_sigma.push_back(is_aruco ? _options_aruco.sigma_pix : _options_slam.sigma_pix);
    f_mult.push_back(is_aruco ? _options_aruco.chi2_multipler : _options_slam.chi2_multipler);
    it0++;
  }
  if (feature_vec.empty()) return;

  // ---- 2..4 on the GPU
  if (!g_slam_ctx) {
    ovgpu_options o;
    ovgpu_default_options(&o);
    o.chi2_multipler = _options_slam.chi2_multipler, o.sigma_pix = _options_slam.sigma_pix; // ArUco corners: per-feature options below
    o.do_fej = state->_options.do_fej, o.do_calib_camera_pose = state->_options.do_calib_camera_pose;
    o.do_calib_camera_intrinsics = state->_options.do_calib_camera_intrinsics, o.feat_rep_msckf = OVGPU_REP_GLOBAL_3D;
    g_slam_ctx.reset(new ovgpu_shim::Context(o));
  }
  const ovgpu_state_view sv = fs.view();
  const ovgpu_features_view fv = ff.view();
  ovgpu_landmarks_view lv;
  lv.L = (int32_t)lm_cov.size(), lv.feat_rep = (int32_t)rep; // ovgpu_feat_rep follows the enum order of LandmarkRepresentation.h:38-46
  lv.p_value = lm_value.data(), lv.p_fej = lm_fej.data(), lv.cov_id = lm_cov.data();
  lv.anchor_cam = lm_anchor_cam.data(), lv.anchor_clone = lm_anchor_clone.data();
  g_slam_ctx->check(ovgpu_set_state(g_slam_ctx->get(), &sv), "ovgpu_set_state");
  g_slam_ctx->check(ovgpu_set_landmarks(g_slam_ctx->get(), &lv), "ovgpu_set_landmarks");
  g_slam_ctx->check(ovgpu_set_features(g_slam_ctx->get(), &fv), "ovgpu_set_features");
  if (any_aruco) // rows come back scaled to _options_slam.sigma_pix, so R_big below stays isotropic
    g_slam_ctx->check(ovgpu_set_feature_options(g_slam_ctx->get(), f_sigma.data(), f_mult.data()), "ovgpu_set_feature_options");
  const int F = fv.F, Dmax = 6 * sv.C + 14 * sv.K + 3 * lv.L; // upper bound (a single-depth landmark has one column)
  std::vector<int32_t> status(F), col_cov(Dmax);
  std::vector<double> H((size_t)Dmax * Dmax), r(Dmax);
  int32_t D = 0, rows = 0;
  ovgpu_update_stats stats;
#ifdef OVGPU_SHIM_MODE_B
  // mode B: the update is applied on the device (ovgpu_slam_update); dx and P' are written back through StateAccess.  Landmark
  // values follow from dx like every other variable (Landmark::update = Vec::update), so nothing else has to come back.
  std::vector<double> dx_dev((size_t)sv.N), P_dev((size_t)sv.N * sv.N);
  g_slam_ctx->check(ovgpu_slam_update(g_slam_ctx->get(), lm_index.data(), status.data(), nullptr, nullptr, dx_dev.data(), P_dev.data(), nullptr, &stats),
                    "ovgpu_slam_update");
  rows = stats.n_rows;
#else
  g_slam_ctx->check(ovgpu_slam_compress(g_slam_ctx->get(), lm_index.data(), status.data(), nullptr, nullptr, &D, &rows, col_cov.data(), H.data(),
                                        r.data(), &stats),
                    "ovgpu_slam_compress");
#endif

  // ---- side effects (UpdaterSLAM.cpp:410-420, :452-454): rejected tracks erased and flagged, fail count bumped; used tracks flagged
  size_t f = 0;
  auto it1 = feature_vec.begin();
  while (it1 != feature_vec.end()) {
    (*it1)->to_delete = true;
    if (status[f] == OVGPU_FEAT_CHI2_REJECTED) {
      state->_features_SLAM.at((*it1)->featid)->update_fail_count++;
      it1 = feature_vec.erase(it1);
    } else {
      it1++;
    }
    f++;
  }
  if (rows < 1) return; // :456-458

#ifdef OVGPU_SHIM_MODE_B
  ovgpu_shim::StateAccess::apply_update(*state, P_dev.data(), dx_dev.data(), sv.N); // StateHelper.cpp:166-196
  return;
#endif
  // ---- 5. the stock EKF update on the compressed system (:470)
  std::vector<std::shared_ptr<Type>> Hx_order_big;
  for (int c = 0; c < D;) {
    std::shared_ptr<Type> v;
    for (const auto &cand : var_of_cov)
      if (cand->id() == col_cov[c]) v = cand;
    if (!v) throw std::runtime_error("ovgpu: Jacobian column without a state variable");
    Hx_order_big.push_back(v);
    c += v->size();
  }
  Eigen::MatrixXd Hx_big = Eigen::Map<const Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>>(H.data(), rows, D);
  Eigen::VectorXd res_big = Eigen::Map<const Eigen::VectorXd>(r.data(), rows);
  Eigen::MatrixXd R_big = _options_slam.sigma_pix_sq * Eigen::MatrixXd::Identity(rows, rows);
  StateHelper::EKFUpdate(state, Hx_order_big, Hx_big, res_big, R_big);
}
