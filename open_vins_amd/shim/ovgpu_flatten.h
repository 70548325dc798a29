// ovgpu_flatten.h — host-side marshalling between the reference's containers and the C ABI of include/ovgpu.h.
//
// Header-only, no Eigen / ROS / Boost: the drop-in translation units (UpdaterMSCKF.cpp, FeatureInitializer.cpp in
// this directory) include the reference's own headers and hand plain pointers to these helpers; the self-test
// (selftest.cpp) exercises the same code with std containers.  Nothing here computes: the arithmetic of the path
// lives in libovgpu.so.
//
// Reference types being flattened:
//   ov_core::Feature                (ov_core/src/feat/Feature.h:39-98)     -> FlatFeatures
//   ov_msckf::State (subset)        (ov_msckf/src/state/State.h:137-192)   -> FlatState
#pragma once
#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <vector>

#include "../../include/ovgpu.h"

namespace ovgpu_shim {

// Feature::clean_old_measurements (Feature.cpp:26-53) keeps a measurement when its timestamp is, with exact
// double equality, one of the clone times.  The same test gives the clone index the ABI wants.
class CloneIndex {
public:
  explicit CloneIndex(const std::vector<double> &clone_times) {
    for (size_t i = 0; i < clone_times.size(); i++) index_[clone_times[i]] = (int32_t)i;
  }
  int32_t find(double t) const {
    auto it = index_.find(t);
    return it == index_.end() ? -1 : it->second;
  }

private:
  std::unordered_map<double, int32_t> index_;
};

struct FlatFeatures {
  std::vector<int32_t> meas_offsets{0};
  std::vector<float> uv, uvn;
  std::vector<int32_t> clone_idx, cam_idx;
  std::vector<double> meas_time; // kept on the host: Feature::anchor_clone_timestamp is read back from it

  int32_t F() const { return (int32_t)meas_offsets.size() - 1; }
  int32_t M() const { return (int32_t)clone_idx.size(); }

  // Appends one camera's measurements of the feature being built, in the order they are stored
  // (Feature::uvs[cam], uvs_norm[cam], timestamps[cam]); measurements whose time is not a clone time are
  // dropped exactly as clean_old_measurements would.  Call once per camera in the iteration order of
  // Feature::timestamps — the anchor rule of FeatureInitializer.cpp:36-46 is evaluated on that order.
  template <class GetUV, class GetUVN>
  int add_camera(int cam, const std::vector<double> &times, GetUV get_uv, GetUVN get_uvn, const CloneIndex &clones) {
    int kept = 0;
    for (size_t i = 0; i < times.size(); i++) {
      const int32_t ci = clones.find(times[i]);
      if (ci < 0) continue;
      float a, b;
      get_uv(i, a, b);
      uv.push_back(a), uv.push_back(b);
      get_uvn(i, a, b);
      uvn.push_back(a), uvn.push_back(b);
      clone_idx.push_back(ci), cam_idx.push_back(cam), meas_time.push_back(times[i]);
      kept++;
    }
    return kept;
  }
  void end_feature() { meas_offsets.push_back(M()); }

  ovgpu_features_view view() const {
    ovgpu_features_view v;
    v.F = F(), v.M = M();
    v.meas_offsets = meas_offsets.data();
    v.uv = uv.data(), v.uvn = uvn.data();
    v.clone_idx = clone_idx.data(), v.cam_idx = cam_idx.data();
    return v;
  }
};

struct FlatState {
  int32_t N = 0;
  std::vector<double> P;                                 // [N*N] row-major
  std::vector<double> clone_q_p, clone_q_p_fej;          // [7C]
  std::vector<int32_t> clone_cov_id;                     // [C]
  std::vector<double> clone_times;                       // [C] (host only)
  std::vector<double> calib_q_p, intrinsics;             // [7K], [8K]
  std::vector<uint8_t> cam_is_fisheye;                   // [K]
  std::vector<int32_t> calib_cov_id, intr_cov_id;        // [K]

  void add_clone(double t, const double q_xyzw[4], const double p[3], const double q_fej[4], const double p_fej[3], int cov_id) {
    clone_times.push_back(t);
    clone_q_p.insert(clone_q_p.end(), q_xyzw, q_xyzw + 4), clone_q_p.insert(clone_q_p.end(), p, p + 3);
    clone_q_p_fej.insert(clone_q_p_fej.end(), q_fej, q_fej + 4), clone_q_p_fej.insert(clone_q_p_fej.end(), p_fej, p_fej + 3);
    clone_cov_id.push_back(cov_id);
  }
  void add_camera(const double q_ItoC[4], const double p_IinC[3], const double intr8[8], bool fisheye, int calib_id, int intr_id) {
    calib_q_p.insert(calib_q_p.end(), q_ItoC, q_ItoC + 4), calib_q_p.insert(calib_q_p.end(), p_IinC, p_IinC + 3);
    intrinsics.insert(intrinsics.end(), intr8, intr8 + 8);
    cam_is_fisheye.push_back(fisheye ? 1 : 0);
    calib_cov_id.push_back(calib_id), intr_cov_id.push_back(intr_id);
  }
  ovgpu_state_view view() const {
    ovgpu_state_view v;
    v.N = N, v.C = (int32_t)clone_cov_id.size(), v.K = (int32_t)calib_cov_id.size(), v._pad0 = 0;
    v.P = P.data();
    v.clone_q_p = clone_q_p.data(), v.clone_q_p_fej = clone_q_p_fej.data(), v.clone_cov_id = clone_cov_id.data();
    v.calib_q_p = calib_q_p.data(), v.intrinsics = intrinsics.data(), v.cam_is_fisheye = cam_is_fisheye.data();
    v.calib_cov_id = calib_cov_id.data(), v.intr_cov_id = intr_cov_id.data();
    return v;
  }
};

// RAII owner of an ovgpu_ctx; throws with ovgpu_last_error() on failure (the reference's update path has no
// error channel: UpdaterMSCKF::update returns void and StateHelper::EKFUpdate exits on a bad covariance).
class Context {
public:
  Context(const ovgpu_options &o, int device = 0) {
    const int rc = ovgpu_create(&o, device, &ctx_);
    if (rc != OVGPU_OK) throw std::runtime_error(std::string("ovgpu_create: ") + ovgpu_last_error());
  }
  ~Context() { ovgpu_destroy(ctx_); }
  Context(const Context &) = delete;
  Context &operator=(const Context &) = delete;
  ovgpu_ctx *get() const { return ctx_; }
  void check(int rc, const char *what) const {
    if (rc != OVGPU_OK) throw std::runtime_error(std::string(what) + ": " + ovgpu_last_error());
  }

private:
  ovgpu_ctx *ctx_ = nullptr;
};

} // namespace ovgpu_shim
