// ovgpu_shim_common.h — what the drop-in translation units of this directory share: the snapshot of ov_msckf::State into the
// flat views of include/ovgpu.h, the flattening of ov_core::Feature tracks, a context cache keyed by the option values, and
// the hand-over of a compressed system to the stock StateHelper::EKFUpdate.  Compiled inside the open_vins tree (it includes the
// reference's headers); tests/test_shim.py compiles it against the stand-ins of tests/shim_mock.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstring>
#include <map>
#include <memory>
#include <mutex>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include "cam/CamBase.h"
#include "cam/CamEqui.h"
#include "feat/Feature.h"
#include "feat/FeatureInitializer.h"
#include "feat/FeatureInitializerOptions.h"
#include "state/State.h"
#include "state/StateHelper.h"
#include "types/Landmark.h"
#include "types/LandmarkRepresentation.h"
#include "types/PoseJPL.h"
#include "types/Type.h"
#include "types/Vec.h"

#include "ovgpu.h"
#include "ovgpu_flatten.h"

namespace ovgpu_shim {

// One context per distinct option set (VioManager owns one UpdaterMSCKF and one UpdaterSLAM, so in practice one or two): the
// options are compared on every call, an updater constructed later with other thresholds gets its own context.
inline Context &context_for(const ovgpu_options &o) {
  static std::vector<std::pair<ovgpu_options, std::unique_ptr<Context>>> cache;
  static std::mutex cache_mutex; // (updaters of two filters on two threads may construct their contexts at once; a Context itself is one caller's at a time)
  const std::lock_guard<std::mutex> guard(cache_mutex);
  for (auto &e : cache)
    if (std::memcmp(&e.first, &o, sizeof(o)) == 0) return *e.second;
  cache.emplace_back(o, std::unique_ptr<Context>(new Context(o)));
  return *cache.back().second;
}

// UpdaterOptions + FeatureInitializerOptions + the StateOptions subset -> ovgpu_options (every byte defined: the struct is a cache key)
template <class UpdaterOptionsT>
inline ovgpu_options make_options(const UpdaterOptionsT &u, const ov_core::FeatureInitializerOptions &f, const ov_msckf::StateOptions &s, int feat_rep) {
  ovgpu_options o;
  std::memset(&o, 0, sizeof(o));
  ovgpu_default_options(&o);
  o.chi2_multipler = u.chi2_multipler, o.sigma_pix = u.sigma_pix;
  o.triangulate_1d = f.triangulate_1d, o.refine_features = f.refine_features, o.max_runs = f.max_runs;
  o.init_lamda = f.init_lamda, o.max_lamda = f.max_lamda, o.min_dx = f.min_dx, o.min_dcost = f.min_dcost, o.lam_mult = f.lam_mult;
  o.min_dist = f.min_dist, o.max_dist = f.max_dist, o.max_baseline = f.max_baseline, o.max_cond_number = f.max_cond_number;
  o.do_fej = s.do_fej, o.do_calib_camera_pose = s.do_calib_camera_pose, o.do_calib_camera_intrinsics = s.do_calib_camera_intrinsics;
  o.feat_rep_msckf = feat_rep;
  return o;
}

// ov_msckf::State -> FlatState (clone order = iteration order of _clones_IMU = ascending time; cameras by ascending id)
struct StateSnapshot {
  FlatState fs;
  std::vector<size_t> cam_ids;                               // camera index -> reference camera id
  std::unordered_map<size_t, int> cam_index;                 // and back
  std::vector<std::shared_ptr<ov_type::PoseJPL>> clone_vars; // clone index -> variable
  std::vector<std::shared_ptr<ov_type::Type>> var_of_cov;    // every variable that can own Jacobian columns

  // with_cov = false: values, ids and order only (the resident-covariance mode: P is on the device already);
  // through_helper = false: State::_Cov read directly (from inside StateHelper's own wrappers, ovgpu_resident_cov.h)
  explicit StateSnapshot(const std::shared_ptr<ov_msckf::State> &state, bool with_cov = true, bool through_helper = true) {
    for (const auto &c : state->_clones_IMU) {
      const Eigen::Vector4d q = c.second->quat(), qf = c.second->quat_fej();
      const Eigen::Vector3d p = c.second->pos(), pf = c.second->pos_fej();
      fs.add_clone(c.first, q.data(), p.data(), qf.data(), pf.data(), c.second->id());
      clone_vars.push_back(c.second), var_of_cov.push_back(c.second);
    }
    for (const auto &c : state->_calib_IMUtoCAM) cam_ids.push_back(c.first);
    std::sort(cam_ids.begin(), cam_ids.end());
    for (size_t k = 0; k < cam_ids.size(); k++) {
      const size_t id = cam_ids[k];
      cam_index[id] = (int)k;
      const auto &pose = state->_calib_IMUtoCAM.at(id);
      const Eigen::Vector4d q = pose->quat();
      const Eigen::Vector3d p = pose->pos();
      const Eigen::Matrix<double, 8, 1> intr = state->_cam_intrinsics.at(id)->value();
      const bool fisheye = std::dynamic_pointer_cast<ov_core::CamEqui>(state->_cam_intrinsics_cameras.at(id)) != nullptr;
      fs.add_camera(q.data(), p.data(), intr.data(), fisheye, state->_options.do_calib_camera_pose ? pose->id() : -1,
                    state->_options.do_calib_camera_intrinsics ? state->_cam_intrinsics.at(id)->id() : -1);
      var_of_cov.push_back(pose), var_of_cov.push_back(state->_cam_intrinsics.at(id));
    }
    fs.N = (int32_t)state->max_covariance_size();
    if (with_cov) {
      const Eigen::MatrixXd P = through_helper ? ov_msckf::StateHelper::get_full_covariance(state) : snapshot_cov_direct(*state); // symmetric: column-major == row-major
      fs.N = (int32_t)P.rows();
      fs.P.assign(P.data(), P.data() + P.size());
    }
  }
  static Eigen::MatrixXd snapshot_cov_direct(ov_msckf::State &s);
};

// Feature::clean_old_measurements (the containers end exactly as Feature.cpp:26-53 leaves them: observations outside the window erased,
// order kept, emptied cameras still present) and the flattening of what is left, in ONE walk over the track; returns the number of
// observations inside the window.  The track is appended to `ff` when it has at least `min_meas` of them, otherwise `ff` is unchanged.
inline int clean_append_track(ov_core::Feature &f, const StateSnapshot &snap, const CloneIndex &clones, FlatFeatures &ff, int min_meas = 0) {
  int total = 0;
  for (auto &pair : f.timestamps) { // iteration order of Feature::timestamps: the anchor rule depends on it
    auto &uvs = f.uvs.at(pair.first);
    auto &uvn = f.uvs_norm.at(pair.first);
    total += ff.add_camera_cleaning(
        snap.cam_index.at(pair.first), pair.second, [&](size_t i, float &a, float &b) { a = uvs[i](0), b = uvs[i](1); },
        [&](size_t i, float &a, float &b) { a = uvn[i](0), b = uvn[i](1); }, [&](size_t i, size_t w) { uvs[w] = std::move(uvs[i]), uvn[w] = std::move(uvn[i]); },
        [&](size_t w) { uvs.resize(w), uvn.resize(w); }, [&](size_t i) { __builtin_prefetch(&uvs[i](0)), __builtin_prefetch(&uvn[i](0)); }, clones);
  }
  if (total < min_meas) ff.drop_open_feature();
  else ff.end_feature();
  return total;
}
// the cleaning alone (the SLAM units clean, decide, then append_track)
inline int clean_track(ov_core::Feature &f, const StateSnapshot &snap) {
  f.clean_old_measurements(snap.fs.clone_times);
  int total = 0;
  for (const auto &pair : f.timestamps) total += (int)pair.second.size();
  return total;
}
inline int flatten_track(ov_core::Feature &f, const StateSnapshot &snap, const CloneIndex &, FlatFeatures &) { return clean_track(f, snap); } // (the SLAM units: clean, decide, then append_track)
// the flattening alone, of a track that is clean already
inline void append_track(const ov_core::Feature &f, const StateSnapshot &snap, const CloneIndex &clones, FlatFeatures &ff) {
  for (const auto &pair : f.timestamps) {
    const auto &uvs = f.uvs.at(pair.first), &uvn = f.uvs_norm.at(pair.first);
    ff.add_camera(snap.cam_index.at(pair.first), pair.second, [&](size_t i, float &a, float &b) { a = uvs[i](0), b = uvs[i](1); },
                  [&](size_t i, float &a, float &b) { a = uvn[i](0), b = uvn[i](1); }, clones);
  }
  ff.end_feature();
}

// A whole batch: clean + flatten every track of `feature_vec` into `ff` (cleared first); tracks with fewer than `min_meas` observations
// inside the window are flagged to_delete and leave the vector (UpdaterMSCKF.cpp:84-92), the others keep their order.  Above a few hundred
// tracks the walk is split over OVGPU_SHIM_FLATTEN_THREADS threads (contiguous ranges of the vector, each into its own buffers, joined in
// order: the batch is the same whatever the count); -DOVGPU_SHIM_FLATTEN_THREADS=1 keeps it on the caller's thread.
#ifndef OVGPU_SHIM_FLATTEN_THREADS
#define OVGPU_SHIM_FLATTEN_THREADS 4
#endif
inline void clean_flatten_batch(std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec, const StateSnapshot &snap, const CloneIndex &clones, FlatFeatures &ff,
                                int min_meas) {
  ff.clear();
  const size_t n = feature_vec.size();
  std::vector<uint8_t> keep(n);
  const int parts = (OVGPU_SHIM_FLATTEN_THREADS > 1 && n >= 256) ? OVGPU_SHIM_FLATTEN_THREADS : 1;
  if (parts == 1) {
    for (size_t i = 0; i < n; i++) {
      if (i + 2 < n) __builtin_prefetch(feature_vec[i + 2].get());
      keep[i] = clean_append_track(*feature_vec[i], snap, clones, ff, min_meas) >= min_meas;
    }
  } else {
    static ForkJoin pool(OVGPU_SHIM_FLATTEN_THREADS - 1);
    static std::vector<FlatFeatures> part_ff(OVGPU_SHIM_FLATTEN_THREADS); // reused from update to update, like `ff`
    static std::mutex pool_mutex; // the pool and its buffers are one per process: updaters on several threads take turns (ForkJoin::run is not re-entrant)
    const std::lock_guard<std::mutex> pool_guard(pool_mutex);
    const std::function<void(int)> work = [&](int p) {
      FlatFeatures &mine = part_ff[(size_t)p];
      mine.clear();
      for (size_t i = n * (size_t)p / (size_t)parts, e = n * ((size_t)p + 1) / (size_t)parts; i < e; i++) {
        if (i + 2 < e) __builtin_prefetch(feature_vec[i + 2].get());
        keep[i] = clean_append_track(*feature_vec[i], snap, clones, mine, min_meas) >= min_meas;
      }
    };
    pool.run(parts, work);
    for (int p = 0; p < parts; p++) ff.append_batch(part_ff[(size_t)p]);
  }
  size_t w = 0;
  for (size_t i = 0; i < n; i++) {
    if (!keep[i]) {
      feature_vec[i]->to_delete = true;
      continue;
    }
    if (w != i) feature_vec[w] = std::move(feature_vec[i]);
    w++;
  }
  feature_vec.resize(w);
}

// FeatureInitializer's side effects on a Feature (FeatureInitializer.cpp:45-46, :109-110, :333-335, :373) from the device's outputs
inline void write_triangulation(ov_core::Feature &feat, const StateSnapshot &snap, const FlatFeatures &ff, int anchor_meas, const double *pA, const double *pG) {
  if (anchor_meas < 0) return;
  feat.anchor_cam_id = (int)snap.cam_ids[ff.cam_idx[anchor_meas]];
  feat.anchor_clone_timestamp = ff.meas_time[anchor_meas];
  feat.p_FinA = Eigen::Map<const Eigen::Vector3d>(pA);
  feat.p_FinG = Eigen::Map<const Eigen::Vector3d>(pG);
}

// mode A: the compressed (H, r) in the canonical column order -> the stock StateHelper::EKFUpdate (UpdaterMSCKF.cpp:280-285)
inline void ekf_update_with(const std::shared_ptr<ov_msckf::State> &state, const std::vector<std::shared_ptr<ov_type::Type>> &var_of_cov, const int32_t *col_cov,
                            int D, int rows, const double *H, const double *r, double sigma_pix_sq) {
  std::vector<std::shared_ptr<ov_type::Type>> Hx_order_big;
  for (int c = 0; c < D;) {
    std::shared_ptr<ov_type::Type> v;
    for (const auto &cand : var_of_cov)
      if (cand->id() == col_cov[c]) v = cand;
    if (!v) throw std::runtime_error("ovgpu: Jacobian column without a state variable");
    Hx_order_big.push_back(v);
    c += v->size();
  }
  Eigen::MatrixXd Hx_big = Eigen::Map<const Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>>(H, rows, D);
  Eigen::VectorXd res_big = Eigen::Map<const Eigen::VectorXd>(r, rows);
  Eigen::MatrixXd R_big = sigma_pix_sq * Eigen::MatrixXd::Identity(rows, rows);
  ov_msckf::StateHelper::EKFUpdate(state, Hx_order_big, Hx_big, res_big, R_big);
}

// Landmark::value() / fej() in the 3-double form the library holds (a single-depth landmark: constant bearing + inverse depth,
// Landmark.cpp:57-60, :124-140), its anchor as (camera index, clone index)
struct FlatLandmarks {
  std::vector<double> value, fej;
  std::vector<int32_t> cov, anchor_cam, anchor_clone, rep; // rep: Landmark::_feat_representation of each (ABI 7: one view, any mix)
  std::vector<std::shared_ptr<ov_type::Landmark>> lm;
  void add(const std::shared_ptr<ov_type::Landmark> &l, const StateSnapshot &snap, const CloneIndex &clones) {
    const auto rep = l->_feat_representation;
    const bool single = rep == ov_type::LandmarkRepresentation::Representation::ANCHORED_INVERSE_DEPTH_SINGLE;
    const bool relative = ov_type::LandmarkRepresentation::is_relative_representation(rep);
    Eigen::Vector3d v, vf;
    if (single) {
      v << l->uv_norm_zero(0), l->uv_norm_zero(1), l->value()(0);
      vf << l->uv_norm_zero_fej(0), l->uv_norm_zero_fej(1), l->fej()(0);
    } else {
      v = l->value(), vf = l->fej();
    }
    value.insert(value.end(), v.data(), v.data() + 3), fej.insert(fej.end(), vf.data(), vf.data() + 3);
    cov.push_back(l->id());
    this->rep.push_back((int32_t)rep); // ovgpu_feat_rep follows the enum order of LandmarkRepresentation.h:38-46
    anchor_cam.push_back(relative ? snap.cam_index.at(l->_anchor_cam_id) : -1);
    anchor_clone.push_back(relative ? clones.find(l->_anchor_clone_timestamp) : -1);
    lm.push_back(l);
  }
  ovgpu_landmarks_view view() const {
    ovgpu_landmarks_view lv;
    lv.L = (int32_t)cov.size(), lv.feat_rep = rep.empty() ? 0 : rep[0], lv.feat_rep_each = rep.data();
    lv.p_value = value.data(), lv.p_fej = fej.data(), lv.cov_id = cov.data(), lv.anchor_cam = anchor_cam.data(), lv.anchor_clone = anchor_clone.data();
    return lv;
  }
};

#ifndef OVGPU_SHIM_RESIDENT_COV
// (only the resident-covariance build reads State::_Cov without going through StateHelper: ovgpu_resident_cov.h)
inline Eigen::MatrixXd StateSnapshot::snapshot_cov_direct(ov_msckf::State &) { throw std::logic_error("ovgpu: StateSnapshot(through_helper = false) outside the resident-covariance build"); }
#endif

} // namespace ovgpu_shim
