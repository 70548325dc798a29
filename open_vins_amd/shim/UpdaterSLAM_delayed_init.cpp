// Mode-B replacement for the body of ov_msckf::UpdaterSLAM::delayed_init (ov_msckf/src/update/UpdaterSLAM.cpp:61-251,
// rpng/open_vins v2.7): triangulation, the per-feature StateHelper::initialize chain and every EKF update of it run on the GPU
// (ovgpu_slam_delayed_init); the host writes the posterior back.
//
// Unlike the mode-A shims this one has to WRITE State::_Cov and State::_variables, which are private with `friend class
// StateHelper` (State.h:182-192): it needs the one line of ovgpu_state_access.h added to the reference.  Without that patch keep
// the reference's delayed_init: it keeps working on the GPU triangulation through the FeatureInitializer shim, with the
// StateHelper::initialize chain on the CPU.  Every feature is initialised in the representation the reference gives it —
// StateOptions::feat_rep_aruco for an ArUco corner, feat_rep_slam otherwise (:160-166; ovgpu_set_feature_reps, ABI 7) — with the
// ArUco corners' sigma and chi2 multiplier from _options_aruco, next to resident landmarks of any representation.
#include "UpdaterSLAM.h"

#include "ovgpu_shim_common.h"
#include "ovgpu_state_access.h"

using namespace ov_core;
using namespace ov_type;
using namespace ov_msckf;

void UpdaterSLAM::delayed_init(std::shared_ptr<State> state, std::vector<std::shared_ptr<Feature>> &feature_vec) {
  if (feature_vec.empty()) return; // :64-65
  const auto rep = state->_options.feat_rep_slam;
  auto is_single = [](LandmarkRepresentation::Representation r) { return r == LandmarkRepresentation::Representation::ANCHORED_INVERSE_DEPTH_SINGLE; };
  const ovgpu_shim::StateSnapshot snap(state);
  const ovgpu_shim::CloneIndex clones(snap.fs.clone_times);
  const int N0 = snap.fs.N;

  // ---- landmarks already in the state: resident so that every update of the chain corrects them
  ovgpu_shim::FlatLandmarks old;
  for (const auto &kv : state->_features_SLAM) old.add(kv.second, snap, clones);

  // ---- 1. clean the tracks (:75-96) and flatten them
  static thread_local ovgpu_shim::FlatFeatures ff; // reused from update to update: a fresh 3.5 MB of buffers per call costs more in page faults than the flattening itself
  ff.clear();
  std::vector<double> f_sigma, f_mult; // ArUco corners use _options_aruco (:226-232)
  std::vector<int32_t> f_rep;          // ... and StateOptions::feat_rep_aruco (:160-166)
  int Nmax = N0;
  bool any_aruco = false;
  for (auto it = feature_vec.begin(); it != feature_vec.end();) {
    if (ovgpu_shim::flatten_track(**it, snap, clones, ff) < 2) { // :91-93
      (*it)->to_delete = true;
      it = feature_vec.erase(it);
      continue;
    }
    ovgpu_shim::append_track(**it, snap, clones, ff);
    const bool is_aruco = (int)(*it)->featid < state->_options.max_aruco_features;
    any_aruco |= is_aruco;
    f_sigma.push_back(is_aruco ? _options_aruco.sigma_pix : _options_slam.sigma_pix);
    f_mult.push_back(is_aruco ? _options_aruco.chi2_multipler : _options_slam.chi2_multipler);
    const auto frep = is_aruco ? state->_options.feat_rep_aruco : rep;
    f_rep.push_back((int32_t)frep);
    Nmax += is_single(frep) ? 1 : 3; // landmark size, :199
    ++it;
  }
  if (feature_vec.empty()) return;

  // ---- 2..4 on the GPU
  ovgpu_shim::Context &cx = ovgpu_shim::context_for(ovgpu_shim::make_options(_options_slam, initializer_feat->config(), state->_options, (int)rep));
  ovgpu_ctx *ctx = cx.get();
  const ovgpu_state_view sv = snap.fs.view();
  const ovgpu_features_view fv = ff.view();
  const ovgpu_landmarks_view lv = old.view();
  cx.check(ovgpu_set_state(ctx, &sv), "ovgpu_set_state");
  cx.check(ovgpu_set_landmarks(ctx, &lv), "ovgpu_set_landmarks");
  cx.check(ovgpu_set_features(ctx, &fv), "ovgpu_set_features");
  if (any_aruco) cx.check(ovgpu_set_feature_options(ctx, f_sigma.data(), f_mult.data()), "ovgpu_set_feature_options");
  if (any_aruco && state->_options.feat_rep_aruco != rep) cx.check(ovgpu_set_feature_reps(ctx, f_rep.data()), "ovgpu_set_feature_reps");
  const int F = fv.F;
  std::vector<int32_t> status(F), new_cov(F), acam(F), aclone(F), tri_anchor(F);
  std::vector<double> new_val(3 * (size_t)F), new_fej(3 * (size_t)F), dx_seq((size_t)F * Nmax), Pout((size_t)Nmax * Nmax), pA(3 * (size_t)F), pG(3 * (size_t)F);
  int32_t N1 = 0;
  cx.check(ovgpu_slam_delayed_init(ctx, (int32_t)rep, status.data(), nullptr, nullptr, new_cov.data(), new_val.data(), new_fej.data(), acam.data(),
                                   aclone.data(), dx_seq.data(), &N1, Pout.data(), nullptr),
           "ovgpu_slam_delayed_init");
  cx.check(ovgpu_get_triangulation(ctx, pA.data(), pG.data(), tri_anchor.data()), "ovgpu_get_triangulation");

  // ---- write the posterior back
  // (a) variables the library does not hold (IMU, time offset, IMU intrinsics): the corrections of the chain, in order
  //     (StateHelper.cpp:185-187)
  std::vector<char> held(N0, 0);
  auto mark = [&](const std::shared_ptr<Type> &v) {
    for (int i = 0; i < v->size(); i++) held[v->id() + i] = 1;
  };
  for (const auto &c : snap.clone_vars) mark(c);
  for (size_t id : snap.cam_ids) {
    if (state->_options.do_calib_camera_pose) mark(state->_calib_IMUtoCAM.at(id));
    if (state->_options.do_calib_camera_intrinsics) mark(state->_cam_intrinsics.at(id));
  }
  for (const auto &lm : old.lm) mark(lm);
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
    cx.check(ovgpu_get_state(ctx, nullptr, cq.data(), kq.data(), iq.data()), "ovgpu_get_state");
    for (size_t i = 0; i < snap.clone_vars.size(); i++) snap.clone_vars[i]->set_value(Eigen::Map<const Eigen::Matrix<double, 7, 1>>(cq.data() + 7 * i));
    for (size_t k = 0; k < snap.cam_ids.size(); k++) {
      if (state->_options.do_calib_camera_pose)
        state->_calib_IMUtoCAM.at(snap.cam_ids[k])->set_value(Eigen::Map<const Eigen::Matrix<double, 7, 1>>(kq.data() + 7 * k));
      if (state->_options.do_calib_camera_intrinsics)
        state->_cam_intrinsics.at(snap.cam_ids[k])->set_value(Eigen::Map<const Eigen::Matrix<double, 8, 1>>(iq.data() + 8 * k));
    }
    ovgpu_shim::StateAccess::refresh_cameras(*state); // StateHelper.cpp:191-196: the trackers undistort through the camera objects
    int32_t L1 = 0;
    std::vector<double> lv1(3 * (old.lm.size() + (size_t)F));
    cx.check(ovgpu_get_landmarks(ctx, &L1, lv1.data(), nullptr, nullptr, nullptr, nullptr), "ovgpu_get_landmarks");
    for (size_t l = 0; l < old.lm.size(); l++) {
      if (is_single(old.lm[l]->_feat_representation)) old.lm[l]->set_value(Eigen::Matrix<double, 1, 1>(lv1[3 * l + 2]));
      else old.lm[l]->set_value(Eigen::Map<const Eigen::Vector3d>(lv1.data() + 3 * l));
    }
  }
  // (c) covariance (StateHelper.cpp:552-558 grew it by the landmark size per accepted feature)
  ovgpu_shim::StateAccess::cov(*state) = Eigen::Map<const Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>>(Pout.data(), N1, N1);
  // (d) the new landmarks (:203-221, :233-235) and the side effects on feature_vec (:235-239)
  size_t f = 0;
  for (auto it = feature_vec.begin(); it != feature_vec.end(); f++) {
    // the triangulation's side effects on the Feature, for every representation (FeatureInitializer.cpp:45-46, :109-110, :333-335)
    ovgpu_shim::write_triangulation(**it, snap, ff, tri_anchor[f], &pA[3 * f], &pG[3 * f]);
    (*it)->to_delete = true;
    if (new_cov[f] < 0) {
      it = feature_vec.erase(it);
      continue;
    }
    const auto frep = (LandmarkRepresentation::Representation)f_rep[f];
    const bool single = is_single(frep), relative = LandmarkRepresentation::is_relative_representation(frep);
    auto landmark = std::make_shared<Landmark>(single ? 1 : 3);
    landmark->_featid = (*it)->featid;
    landmark->_feat_representation = frep;
    landmark->_unique_camera_id = (*it)->anchor_cam_id; // :214 (VioManager.cpp:469-471 asserts it and decides should_marg with it)
    if (relative) {
      landmark->_anchor_cam_id = (int)snap.cam_ids[acam[f]];
      landmark->_anchor_clone_timestamp = snap.fs.clone_times[aclone[f]];
    }
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
    state->_features_SLAM.insert({(*it)->featid, landmark});
    ++it;
  }
}
