// Drop-in replacement for ov_core/src/feat/FeatureInitializer.cpp (rpng/open_vins v2.7): the class of FeatureInitializer.h:40-159
// with its single-feature entry points served by libovgpu.
//
// The reference calls these per feature inside loops (UpdaterMSCKF.cpp:117-142, UpdaterSLAM.cpp:113-147,
// VioManagerHelper.cpp:251-293).  The MSCKF shim does not go through here (it hands the whole batch to the library); this file
// keeps the class usable for the remaining callers: one feature = a batch of one.  single_triangulation runs the device
// triangulation WITHOUT the Gauss-Newton refinement (a context with refine_features = 0); single_gaussnewton runs the refinement
// alone (ovgpu_refine) from the anchor and p_FinA the Feature carries, as the reference does (FeatureInitializer.cpp:199-216).
// Contexts are keyed by the option values of the calling instance.
#include "FeatureInitializer.h"

#include "Feature.h" // (FeatureInitializer.h only forward-declares the class: FeatureInitializer.cpp:24 includes it, too)

#include <cstring>
#include <memory>
#include <stdexcept>
#include <unordered_map>
#include <vector>

#include "ovgpu.h"
#include "ovgpu_flatten.h"

using namespace ov_core;

namespace {
ovgpu_shim::Context &context(const FeatureInitializerOptions &f, int refine, int one_d) {
  ovgpu_options o;
  std::memset(&o, 0, sizeof(o));
  ovgpu_default_options(&o);
  o.max_runs = f.max_runs, o.init_lamda = f.init_lamda, o.max_lamda = f.max_lamda, o.min_dx = f.min_dx, o.min_dcost = f.min_dcost;
  o.lam_mult = f.lam_mult, o.min_dist = f.min_dist, o.max_dist = f.max_dist, o.max_baseline = f.max_baseline, o.max_cond_number = f.max_cond_number;
  o.refine_features = refine, o.triangulate_1d = one_d;
  static std::vector<std::pair<ovgpu_options, std::unique_ptr<ovgpu_shim::Context>>> cache;
  for (auto &e : cache)
    if (std::memcmp(&e.first, &o, sizeof(o)) == 0) return *e.second;
  cache.emplace_back(o, std::unique_ptr<ovgpu_shim::Context>(new ovgpu_shim::Context(o)));
  return *cache.back().second;
}

typedef std::unordered_map<size_t, std::unordered_map<double, FeatureInitializer::ClonePose>> ClonesCam;

// flattens clonesCAM (camera id -> clone time -> ClonePose) and one feature; returns false when < 2 usable measurements
struct OneFeature {
  std::vector<double> R, p, times;
  std::vector<size_t> cam_ids;
  ovgpu_shim::FlatFeatures ff;
  int C = 0, K = 0;
};
bool flatten(const std::shared_ptr<Feature> &feat, ClonesCam &clonesCAM, OneFeature &o) {
  std::unordered_map<double, int> tindex;
  for (auto &cam : clonesCAM) {
    o.cam_ids.push_back(cam.first);
    for (auto &t : cam.second)
      if (!tindex.count(t.first)) tindex[t.first] = (int)o.times.size(), o.times.push_back(t.first);
  }
  o.K = (int)o.cam_ids.size(), o.C = (int)o.times.size();
  o.R.assign((size_t)9 * o.K * o.C, 0.0), o.p.assign((size_t)3 * o.K * o.C, 0.0);
  std::unordered_map<size_t, int> cindex;
  for (int k = 0; k < o.K; k++) {
    cindex[o.cam_ids[k]] = k;
    for (auto &t : clonesCAM.at(o.cam_ids[k])) { // ClonePose::Rot() / pos() are not const (FeatureInitializer.h:78-81)
      const int i = k * o.C + tindex.at(t.first);
      const Eigen::Matrix<double, 3, 3, Eigen::RowMajor> Rm = t.second.Rot();
      const Eigen::Vector3d pv = t.second.pos();
      std::copy(Rm.data(), Rm.data() + 9, o.R.begin() + 9 * i);
      std::copy(pv.data(), pv.data() + 3, o.p.begin() + 3 * i);
    }
  }
  const ovgpu_shim::CloneIndex clones(o.times);
  int total = 0;
  for (const auto &pair : feat->timestamps) { // iteration order of Feature::timestamps: the anchor rule depends on it
    const auto &uvs = feat->uvs.at(pair.first), &uvn = feat->uvs_norm.at(pair.first);
    total += o.ff.add_camera(cindex.at(pair.first), pair.second, [&](size_t i, float &a, float &b) { a = uvs[i](0), b = uvs[i](1); },
                             [&](size_t i, float &a, float &b) { a = uvn[i](0), b = uvn[i](1); }, clones);
  }
  o.ff.end_feature();
  return total >= 2;
}

// seeded: single_gaussnewton on the estimate the Feature already carries (its anchor and p_FinA, FeatureInitializer.cpp:199-216)
bool run(ovgpu_shim::Context &ctx, const std::shared_ptr<Feature> &feat, ClonesCam &clonesCAM, bool seeded = false) {
  OneFeature o;
  if (!flatten(feat, clonesCAM, o)) return false;
  ctx.check(ovgpu_set_camera_poses(ctx.get(), o.C, o.K, o.R.data(), o.p.data()), "ovgpu_set_camera_poses");
  const ovgpu_features_view fv = o.ff.view();
  ctx.check(ovgpu_set_features(ctx.get(), &fv), "ovgpu_set_features");
  double pA[3], pG[3];
  int32_t anchor = -1, status = 0;
  if (seeded) {
    for (size_t i = 0; i < o.ff.cam_idx.size(); i++)
      if ((int)o.cam_ids[o.ff.cam_idx[i]] == feat->anchor_cam_id && o.ff.meas_time[i] == feat->anchor_clone_timestamp) anchor = (int32_t)i;
    if (anchor < 0) return false; // the Feature's anchor is not among its measurements in this window
    const double seed[3] = {feat->p_FinA(0), feat->p_FinA(1), feat->p_FinA(2)};
    ctx.check(ovgpu_refine(ctx.get(), seed, &anchor, pA, pG, &status), "ovgpu_refine");
  } else {
    ctx.check(ovgpu_triangulate(ctx.get(), pA, pG, &anchor, &status), "ovgpu_triangulate");
  }
  if (anchor >= 0) { // FeatureInitializer.cpp:45-46
    feat->anchor_cam_id = (int)o.cam_ids[o.ff.cam_idx[anchor]];
    feat->anchor_clone_timestamp = o.ff.meas_time[anchor];
  }
  if (status != OVGPU_FEAT_USED) return false;
  feat->p_FinA = Eigen::Map<const Eigen::Vector3d>(pA); // :109-110, :333-335
  feat->p_FinG = Eigen::Map<const Eigen::Vector3d>(pG);
  return true;
}
} // namespace

bool FeatureInitializer::single_triangulation(std::shared_ptr<Feature> feat, std::unordered_map<size_t, std::unordered_map<double, ClonePose>> &clonesCAM) {
  return run(context(_options, 0, 0), feat, clonesCAM);
}

bool FeatureInitializer::single_triangulation_1d(std::shared_ptr<Feature> feat, std::unordered_map<size_t, std::unordered_map<double, ClonePose>> &clonesCAM) {
  return run(context(_options, 0, 1), feat, clonesCAM);
}

bool FeatureInitializer::single_gaussnewton(std::shared_ptr<Feature> feat, std::unordered_map<size_t, std::unordered_map<double, ClonePose>> &clonesCAM) {
  return run(context(_options, 1, _options.triangulate_1d ? 1 : 0), feat, clonesCAM, true);
}
