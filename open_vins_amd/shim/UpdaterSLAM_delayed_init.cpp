// Mode-B replacement for the body of ov_msckf::UpdaterSLAM::delayed_init (ov_msckf/src/update/UpdaterSLAM.cpp:61-251,
// rpng/open_vins v2.7): triangulation, the per-feature StateHelper::initialize chain and every EKF update of it run on
// the GPU (ovgpu_slam_delayed_init); the host writes the posterior back.
//
// Unlike the mode-A shims this one has to WRITE State::_Cov and State::_variables, which are private with
// `friend class StateHelper` (State.h:182-192).  It therefore needs ONE line added to the reference, inside class State:
//
//     friend struct ovgpu_shim::StateAccess;      // State.h, next to `friend class StateHelper;`
//
// Without that patch keep the reference's delayed_init: it keeps working on the GPU triangulation through the
// FeatureInitializer shim (FeatureInitializer.cpp), with the StateHelper::initialize chain on the CPU.
// Not covered here (falls to the reference code): ArUco features (own options / representation, :166-168, :227-232).
#include "UpdaterSLAM.h"

#include "feat/Feature.h"
#include "state/State.h"
#include "state/StateHelper.h"
#include "types/Landmark.h"
#include "types/LandmarkRepresentation.h"

#include "ovgpu.h"
#include "ovgpu_flatten.h"
#include "ovgpu_state_access.h"

using namespace ov_core;
using namespace ov_type;
using namespace ov_msckf;


namespace {
std::unique_ptr<ovgpu_shim::Context> g_init_ctx;
}

void UpdaterSLAM::delayed_init(std::shared_ptr<State> state, std::vector<std::shared_ptr<Feature>> &feature_vec) {
  if (feature_vec.empty()) return; // :64-65
  const auto rep = state->_options.feat_rep_slam;
  const bool single = rep == LandmarkRepresentation::Representation::ANCHORED_INVERSE_DEPTH_SINGLE;
  const int lsz = single ? 1 : 3; // landmark_size, :199

  // ---- state snapshot (same flattening as the other shims)
  ovgpu_shim::FlatState fs;
  std::vector<std::shared_ptr<PoseJPL>> clone_vars;
  for (const auto &c : state->_clones_IMU) {
    const Eigen::Vector4d q = c.second->quat(), qf = c.second->quat_fej();
    const Eigen::Vector3d p = c.second->pos(), pf = c.second->pos_fej();
    fs.add_clone(c.first, q.data(), p.data(), qf.data(), pf.data(), c.second->id());
    clone_vars.push_back(c.second);
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
  }
  const Eigen::MatrixXd P = StateHelper::get_full_covariance(state);
  const int N0 = (int)P.rows();
  fs.N = N0;
  fs.P.assign(P.data(), P.data() + P.size());

  // ---- landmarks already in the state: resident so that every update of the chain corrects them
  std::vector<std::shared_ptr<Landmark>> old_lm;
  std::vector<double> lm_value, lm_fej;
  std::vector<int32_t> lm_cov, lm_acam, lm_aclone;
  const ovgpu_shim::CloneIndex clones(fs.clone_times);
  const bool relative = LandmarkRepresentation::is_relative_representation(rep);
  for (const auto &kv : state->_features_SLAM) {
    const auto &lm = kv.second;
    if (lm->_feat_representation != rep) continue; // e.g. ArUco tags in another representation: corrected below through dx_seq
    Eigen::Vector3d v, vf;
    if (single) {
      v << lm->uv_norm_zero(0), lm->uv_norm_zero(1), lm->value()(0);
      vf << lm->uv_norm_zero_fej(0), lm->uv_norm_zero_fej(1), lm->fej()(0);
    } else {
      v = lm->value(), vf = lm->fej();
    }
    lm_value.insert(lm_value.end(), v.data(), v.data() + 3), lm_fej.insert(lm_fej.end(), vf.data(), vf.data() + 3);
    lm_cov.push_back(lm->id());
    lm_acam.push_back(relative ? cam_index.at(lm->_anchor_cam_id) : -1);
    lm_aclone.push_back(relative ? clones.find(lm->_anchor_clone_timestamp) : -1);
    old_lm.push_back(lm);
  }

  // ---- 1. clean the tracks (:75-96) and flatten them
  ovgpu_shim::FlatFeatures ff;
  auto it0 = feature_vec.begin();
  std::vector<double> f_sigma, f_mult; // ArUco corners use _options_aruco (:226-232)
  bool any_aruco = false;
  while (it0 != feature_vec.end()) {
    (*it0)->clean_old_measurements(fs.clone_times);
    int ct_meas = 0;
    for (const auto &pair : (*it0)->timestamps) ct_meas += (int)pair.second.size();
    if (ct_meas < 2) { // :91-93
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
    const bool is_aruco = (int)f.featid < state->_options.max_aruco_features; // :226-232
    any_aruco |= is_aruco;
    f_sigma.push_back(is_aruco ? _options_aruco.sigma_pix : _options_slam.sigma_pix);
    f_mult.push_back(is_aruco ? _options_aruco.chi2_multipler : _options_slam.chi2_multipler);
    it0++;
  }
  if (feature_vec.empty()) return;

  // ---- 2..4 on the GPU
  if (!g_init_ctx) {
    ovgpu_options o;
    ovgpu_default_options(&o);
    const FeatureInitializerOptions &fo = initializer_feat->config(); // FeatureInitializerOptions.h:33-69
    o.triangulate_1d = fo.triangulate_1d, o.refine_features = fo.refine_features, o.max_runs = fo.max_runs;
    o.init_lamda = fo.init_lamda, o.max_lamda = fo.max_lamda, o.min_dx = fo.min_dx, o.min_dcost = fo.min_dcost, o.lam_mult = fo.lam_mult;
    o.min_dist = fo.min_dist, o.max_dist = fo.max_dist, o.max_baseline = fo.max_baseline, o.max_cond_number = fo.max_cond_number;
    o.chi2_multipler = _options_slam.chi2_multipler, o.sigma_pix = _options_slam.sigma_pix;
    o.do_fej = state->_options.do_fej, o.do_calib_camera_pose = state->_options.do_calib_camera_pose;
    o.do_calib_camera_intrinsics = state->_options.do_calib_camera_intrinsics, o.feat_rep_msckf = (int32_t)rep;
    g_init_ctx.reset(new ovgpu_shim::Context(o));
  }
  ovgpu_ctx *ctx = g_init_ctx->get();
  const ovgpu_state_view sv = fs.view();
  const ovgpu_features_view fv = ff.view();
  ovgpu_landmarks_view lv;
  lv.L = (int32_t)lm_cov.size(), lv.feat_rep = (int32_t)rep, lv.p_value = lm_value.data(), lv.p_fej = lm_fej.data(), lv.cov_id = lm_cov.data();
  lv.anchor_cam = lm_acam.data(), lv.anchor_clone = lm_aclone.data();
  g_init_ctx->check(ovgpu_set_state(ctx, &sv), "ovgpu_set_state");
  g_init_ctx->check(ovgpu_set_landmarks(ctx, &lv), "ovgpu_set_landmarks");
  g_init_ctx->check(ovgpu_set_features(ctx, &fv), "ovgpu_set_features");
  if (any_aruco) g_init_ctx->check(ovgpu_set_feature_options(ctx, f_sigma.data(), f_mult.data()), "ovgpu_set_feature_options");
  const int F = fv.F, Nmax = N0 + lsz * F;
  std::vector<int32_t> status(F), new_cov(F), acam(F), aclone(F);
  std::vector<double> new_val(3 * (size_t)F), new_fej(3 * (size_t)F), dx_seq((size_t)F * Nmax), Pout((size_t)Nmax * Nmax);
  int32_t N1 = 0;
  g_init_ctx->check(ovgpu_slam_delayed_init(ctx, (int32_t)rep, status.data(), nullptr, nullptr, new_cov.data(), new_val.data(), new_fej.data(),
                                            acam.data(), aclone.data(), dx_seq.data(), &N1, Pout.data(), nullptr),
                    "ovgpu_slam_delayed_init");

  // ---- write the posterior back
  // (a) variables the library does not hold (IMU, time offset, IMU intrinsics, landmarks of another representation):
  //     the corrections of the chain, in order (StateHelper.cpp:185-187)
  std::vector<char> held(N0, 0);
  auto mark = [&](const std::shared_ptr<Type> &v) {
    for (int i = 0; i < v->size(); i++) held[v->id() + i] = 1;
  };
  for (const auto &c : clone_vars) mark(c);
  for (size_t id : cam_ids) {
    if (state->_options.do_calib_camera_pose) mark(state->_calib_IMUtoCAM.at(id));
    if (state->_options.do_calib_camera_intrinsics) mark(state->_cam_intrinsics.at(id));
  }
  for (const auto &lm : old_lm) mark(lm);
  for (int f = 0; f < F; f++) {
    if (new_cov[f] < 0) continue;
    const double *dx = dx_seq.data() + (size_t)f * Nmax;
    for (const auto &var : ovgpu_shim::StateAccess::variables(*state)) {
      if (held[var->id()]) continue;
      var->update(Eigen::Map<const Eigen::VectorXd>(dx + var->id(), var->size()));
    }
  }
  // (b) clones, calibration, resident landmarks: their values after the chain
  {
    std::vector<double> cq(7 * (size_t)sv.C), kq(7 * (size_t)sv.K), iq(8 * (size_t)sv.K);
    g_init_ctx->check(ovgpu_get_state(ctx, nullptr, cq.data(), kq.data(), iq.data()), "ovgpu_get_state");
    for (size_t i = 0; i < clone_vars.size(); i++) clone_vars[i]->set_value(Eigen::Map<const Eigen::Matrix<double, 7, 1>>(cq.data() + 7 * i));
    for (size_t k = 0; k < cam_ids.size(); k++) {
      if (state->_options.do_calib_camera_pose)
        state->_calib_IMUtoCAM.at(cam_ids[k])->set_value(Eigen::Map<const Eigen::Matrix<double, 7, 1>>(kq.data() + 7 * k));
      if (state->_options.do_calib_camera_intrinsics)
        state->_cam_intrinsics.at(cam_ids[k])->set_value(Eigen::Map<const Eigen::Matrix<double, 8, 1>>(iq.data() + 8 * k));
    }
    int32_t L1 = 0;
    g_init_ctx->check(ovgpu_get_landmarks(ctx, &L1, nullptr, nullptr, nullptr, nullptr, nullptr), "ovgpu_get_landmarks");
    std::vector<double> lv1(3 * (size_t)L1);
    g_init_ctx->check(ovgpu_get_landmarks(ctx, &L1, lv1.data(), nullptr, nullptr, nullptr, nullptr), "ovgpu_get_landmarks");
    for (size_t l = 0; l < old_lm.size(); l++) {
      if (single) old_lm[l]->set_value(Eigen::Matrix<double, 1, 1>(lv1[3 * l + 2]));
      else old_lm[l]->set_value(Eigen::Map<const Eigen::Vector3d>(lv1.data() + 3 * l));
    }
  }
  // (c) covariance (StateHelper.cpp:552-558 grew it by 3 per accepted feature)
  ovgpu_shim::StateAccess::cov(*state) = Eigen::Map<const Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>>(Pout.data(), N1, N1);
  // (d) the new landmarks (:203-221, :233-235) and the side effects on feature_vec (:235-239)
  size_t f = 0;
  auto it2 = feature_vec.begin();
  while (it2 != feature_vec.end()) {
    (*it2)->to_delete = true;
    if (new_cov[f] < 0) {
      it2 = feature_vec.erase(it2);
      f++;
      continue;
    }
    auto landmark = std::make_shared<Landmark>(lsz);
    landmark->_featid = (*it2)->featid;
    landmark->_feat_representation = rep;
    if (relative) {
      landmark->_anchor_cam_id = (int)cam_ids[acam[f]];
      landmark->_anchor_clone_timestamp = fs.clone_times[aclone[f]];
    }
    landmark->_unique_camera_id = relative ? landmark->_anchor_cam_id : (*it2)->anchor_cam_id;
    if (single) { // bearing + inverse depth (Landmark::set_from_xyz, Landmark.cpp:124-140)
      landmark->uv_norm_zero << new_val[3 * f], new_val[3 * f + 1], 1.0;
      landmark->uv_norm_zero_fej << new_fej[3 * f], new_fej[3 * f + 1], 1.0;
      landmark->set_value(Eigen::Matrix<double, 1, 1>(new_val[3 * f + 2]));
      landmark->set_fej(Eigen::Matrix<double, 1, 1>(new_fej[3 * f + 2]));
    } else {
      landmark->set_value(Eigen::Map<const Eigen::Vector3d>(new_val.data() + 3 * f));
      landmark->set_fej(Eigen::Map<const Eigen::Vector3d>(new_fej.data() + 3 * f));
    }
    landmark->set_local_id(new_cov[f]); // StateHelper.cpp:572-573
    ovgpu_shim::StateAccess::variables(*state).push_back(landmark);
    state->_features_SLAM.insert({(*it2)->featid, landmark});
    it2++;
    f++;
  }
}
