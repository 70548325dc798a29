// ovgpu_retri.h — VioManager::retriangulate_active_tracks (ov_msckf/src/core/VioManagerHelper.cpp:190-387, rpng/open_vins v2.7):
// the running linear triangulation of the active tracks on the device (ovgpu_retriangulate).  The function keeps its display
// code and the SLAM landmarks it appends from the state (:311-327); the loop over the cameras and observations (:221-301) and the
// re-projection of the triangulated tracks into cam0 (:345-379) become ONE call of this helper, which fills the same two members.
// Mode A: nothing of the reference's private state is touched.
#pragma once
#include "ovgpu_shim_common.h"

namespace ovgpu_shim {

// last_obs / last_ids: TrackBase::get_last_obs() / get_last_ids() with the pixel as a pair (cv::KeyPoint::pt.x, .y);
// sensor_ids: CameraData::sensor_ids (the camera order of the frame); time: CameraData::timestamp (must be a clone, :199).
// Fills active_tracks_posinG with the triangulated MSCKF tracks and active_tracks_uvd with (u, v, depth) of those seen in cam0.
template <class ObsMap, class IdMap, class OptionsT>
inline void retriangulate(const std::shared_ptr<ov_msckf::State> &state, double time, const std::vector<int> &sensor_ids, const ObsMap &last_obs,
                          const IdMap &last_ids, const OptionsT &updater_options, const ov_core::FeatureInitializerOptions &featinit_options,
                          std::unordered_map<size_t, Eigen::Vector3d> &active_tracks_posinG, std::unordered_map<size_t, Eigen::Vector3d> &active_tracks_uvd) {
  const StateSnapshot snap(state);
  const CloneIndex clones(snap.fs.clone_times);
  Context &cx = context_for(make_options(updater_options, featinit_options, state->_options, OVGPU_REP_GLOBAL_3D));
  const ovgpu_state_view sv = snap.fs.view();
  cx.check(ovgpu_set_state(cx.get(), &sv), "ovgpu_set_state");
  std::vector<int64_t> ids;
  std::vector<int32_t> cams;
  std::vector<float> uv, uvn;
  for (int cam_id : sensor_ids) {
    const auto &obs = last_obs.at(cam_id);
    const auto &fid = last_ids.at(cam_id);
    const auto &camera = state->_cam_intrinsics_cameras.at(cam_id);
    for (size_t i = 0; i < obs.size(); i++) {
      if (state->_features_SLAM.find(fid[i]) != state->_features_SLAM.end()) continue; // :248-250
      Eigen::Vector2f pd;
      pd << obs[i].first, obs[i].second;
      const Eigen::Vector2f pn = camera->undistort_f(pd); // CamBase::undistort_cv (:253) on the same floats
      ids.push_back((int64_t)fid[i]), cams.push_back(snap.cam_index.at((size_t)cam_id));
      uv.push_back(pd(0)), uv.push_back(pd(1)), uvn.push_back(pn(0)), uvn.push_back(pn(1));
    }
  }
  const int n = (int)ids.size();
  std::vector<int64_t> out_id(std::max(n, 1));
  std::vector<double> pos(3 * (size_t)std::max(n, 1)), uvd(3 * (size_t)std::max(n, 1));
  int32_t nt = 0;
  const int cam0 = snap.cam_index.count(0) ? snap.cam_index.at(0) : -1;
  const auto &c0 = state->_cam_intrinsics_cameras.at(0);
  cx.check(ovgpu_retriangulate(cx.get(), clones.find(time), n, ids.data(), cams.data(), uv.data(), uvn.data(), cam0, c0->w(), c0->h(), &nt, out_id.data(),
                               pos.data(), uvd.data()),
           "ovgpu_retriangulate");
  active_tracks_posinG.clear(), active_tracks_uvd.clear();
  for (int t = 0; t < nt; t++) {
    if (!std::isnan(pos[3 * t])) active_tracks_posinG[(size_t)out_id[t]] = Eigen::Map<const Eigen::Vector3d>(&pos[3 * t]);
    if (!std::isnan(uvd[3 * t])) active_tracks_uvd[(size_t)out_id[t]] = Eigen::Map<const Eigen::Vector3d>(&uvd[3 * t]);
  }
}

} // namespace ovgpu_shim
