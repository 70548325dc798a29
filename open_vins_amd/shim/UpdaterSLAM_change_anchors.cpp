// Mode-B replacement for the body of ov_msckf::UpdaterSLAM::change_anchors (ov_msckf/src/update/UpdaterSLAM.cpp:481-504,
// rpng/open_vins v2.7): every anchored landmark whose anchor clone is about to be marginalised moves to the newest clone, same
// camera; perform_anchor_change (:506-647: both representation Jacobians, Phi = H_f_new^-1 [H_x_old | H_f_old | -H_x_new],
// StateHelper::EKFPropagation) runs on the device for all of them (ovgpu_slam_change_anchors).  Needs the StateAccess line of
// ovgpu_state_access.h (it writes State::_Cov); without it keep the reference's change_anchors.
#include "UpdaterSLAM.h"

#include "ovgpu_shim_common.h"
#include "ovgpu_state_access.h"

using namespace ov_core;
using namespace ov_type;
using namespace ov_msckf;

void UpdaterSLAM::change_anchors(std::shared_ptr<State> state) {
  if ((int)state->_clones_IMU.size() <= state->_options.max_clone_size) return; // :484-486
  const double marg_timestep = state->margtimestep();
  // every landmark of the state in ONE view, each with its own representation (ABI 7): the global ones are skipped by the library as
  // :493-496 skips them, the anchored ones move one after the other on the covariance the previous move left
  const ovgpu_shim::StateSnapshot snap(state);
  const ovgpu_shim::CloneIndex clones(snap.fs.clone_times);
  ovgpu_shim::FlatLandmarks fl;
  bool any = false;
  for (const auto &kv : state->_features_SLAM) {
    fl.add(kv.second, snap, clones);
    any |= LandmarkRepresentation::is_relative_representation(kv.second->_feat_representation) && kv.second->_anchor_clone_timestamp == marg_timestep; // :498-500
  }
  if (!any) return;
  FeatureInitializerOptions fo;
  ovgpu_shim::Context &cx = ovgpu_shim::context_for(ovgpu_shim::make_options(_options_slam, fo, state->_options, OVGPU_REP_GLOBAL_3D));
  const ovgpu_state_view sv = snap.fs.view();
  const ovgpu_landmarks_view lv = fl.view();
  cx.check(ovgpu_set_state(cx.get(), &sv), "ovgpu_set_state");
  cx.check(ovgpu_set_landmarks(cx.get(), &lv), "ovgpu_set_landmarks");
  int32_t moved = 0;
  cx.check(ovgpu_slam_change_anchors(cx.get(), clones.find(marg_timestep), clones.find(state->_timestamp), &moved), "ovgpu_slam_change_anchors");
  if (moved == 0) return;
  // ---- write back: covariance, and value / first estimate / anchor of every landmark that moved
  std::vector<double> P((size_t)sv.N * sv.N);
  cx.check(ovgpu_get_state(cx.get(), P.data(), nullptr, nullptr, nullptr), "ovgpu_get_state");
  ovgpu_shim::StateAccess::cov(*state) = Eigen::Map<const Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>>(P.data(), sv.N, sv.N);
  int32_t L = 0;
  std::vector<double> val(3 * fl.lm.size()), fej(3 * fl.lm.size());
  std::vector<int32_t> acam(fl.lm.size()), aclone(fl.lm.size());
  cx.check(ovgpu_get_landmarks(cx.get(), &L, val.data(), fej.data(), nullptr, acam.data(), aclone.data()), "ovgpu_get_landmarks");
  for (size_t l = 0; l < fl.lm.size(); l++) {
    Landmark &lm = *fl.lm[l];
    if (!LandmarkRepresentation::is_relative_representation(lm._feat_representation) || aclone[l] < 0) continue;
    const double t_new = snap.fs.clone_times[aclone[l]];
    if (t_new == lm._anchor_clone_timestamp) continue; // not moved
    lm._anchor_cam_id = (int)snap.cam_ids[acam[l]], lm._anchor_clone_timestamp = t_new, lm.has_had_anchor_change = true; // :642-646
    if (lm._feat_representation == LandmarkRepresentation::Representation::ANCHORED_INVERSE_DEPTH_SINGLE) {
      lm.uv_norm_zero << val[3 * l], val[3 * l + 1], 1.0;
      lm.uv_norm_zero_fej << fej[3 * l], fej[3 * l + 1], 1.0;
      lm.set_value(Eigen::Matrix<double, 1, 1>(val[3 * l + 2]));
      lm.set_fej(Eigen::Matrix<double, 1, 1>(fej[3 * l + 2]));
    } else {
      lm.set_value(Eigen::Map<const Eigen::Vector3d>(val.data() + 3 * l));
      lm.set_fej(Eigen::Map<const Eigen::Vector3d>(fej.data() + 3 * l));
    }
  }
}
