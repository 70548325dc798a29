// Drop-in replacement for the body of ov_msckf::UpdaterSLAM::update (ov_msckf/src/update/UpdaterSLAM.cpp:253-479,
// rpng/open_vins v2.7).  Delete that definition from UpdaterSLAM.cpp and compile this file next to it (the class declaration,
// the constructor and perform_anchor_change stay the reference's).  Mode A by default, -DOVGPU_SHIM_MODE_B as in UpdaterMSCKF.cpp.
// Landmark representations: all six (LandmarkRepresentation.h:38-46), in any mix: every landmark is handed over with its own
// _feat_representation (ovgpu_landmarks_view::feat_rep_each, ABI 7), so SLAM landmarks in feat_rep_slam and ArUco corners in a
// different feat_rep_aruco share ONE stacked system and one EKF update, as :427-447 builds them (two passes before round 5).
#include "UpdaterSLAM.h"

#include "ovgpu_shim_common.h"
#ifdef OVGPU_SHIM_MODE_B
#include "ovgpu_state_access.h"
#endif

using namespace ov_core;
using namespace ov_type;
using namespace ov_msckf;

namespace {
void update_all(UpdaterOptions &opt_slam, UpdaterOptions &opt_aruco, std::shared_ptr<State> state, std::vector<std::shared_ptr<Feature>> &feature_vec) {
  const ovgpu_shim::StateSnapshot snap(state);
  const ovgpu_shim::CloneIndex clones(snap.fs.clone_times);
  static thread_local ovgpu_shim::FlatFeatures ff; // reused from update to update: a fresh 3.5 MB of buffers per call costs more in page faults than the flattening itself
  ff.clear();
  ovgpu_shim::FlatLandmarks fl;
  std::vector<int32_t> lm_index;
  std::vector<double> f_sigma, f_mult; // per-feature options: ArUco corners use _options_aruco (:392-394, :408-409)
  bool any_aruco = false;
  std::vector<std::shared_ptr<Type>> var_of_cov = snap.var_of_cov;
  for (auto it = feature_vec.begin(); it != feature_vec.end();) {
    std::shared_ptr<Landmark> landmark = state->_features_SLAM.at((*it)->featid);
    // :283-295.  The single-depth representation needs two measurements (its bearing is projected out); a landmark with exactly one
    // is dropped from THIS update without to_delete, so that FeatureDatabase::cleanup keeps the measurement for the next frame.
    const int ct_meas = ovgpu_shim::flatten_track(**it, snap, clones, ff);
    const int required_meas = (landmark->_feat_representation == LandmarkRepresentation::Representation::ANCHORED_INVERSE_DEPTH_SINGLE) ? 2 : 1;
    if (ct_meas < 1) {
      (*it)->to_delete = true;
      it = feature_vec.erase(it);
      continue;
    } else if (ct_meas < required_meas) {
      it = feature_vec.erase(it);
      continue;
    }
    ovgpu_shim::append_track(**it, snap, clones, ff);
    lm_index.push_back((int32_t)fl.cov.size());
    fl.add(landmark, snap, clones); // the library holds Landmark::value() / fej() and applies get_xyz itself (:345-353)
    var_of_cov.push_back(landmark);
    const bool is_aruco = (int)(*it)->featid < state->_options.max_aruco_features; // :392
    any_aruco |= is_aruco;
    f_sigma.push_back(is_aruco ? opt_aruco.sigma_pix : opt_slam.sigma_pix);
    f_mult.push_back(is_aruco ? opt_aruco.chi2_multipler : opt_slam.chi2_multipler);
    ++it;
  }
  if (feature_vec.empty()) return;

  // ---- 2..4 on the GPU
  FeatureInitializerOptions fo; // the SLAM update does not triangulate: defaults keep the context key stable
  ovgpu_shim::Context &ctx = ovgpu_shim::context_for(ovgpu_shim::make_options(opt_slam, fo, state->_options, OVGPU_REP_GLOBAL_3D));
  const ovgpu_state_view sv = snap.fs.view();
  const ovgpu_features_view fv = ff.view();
  const ovgpu_landmarks_view lv = fl.view();
  ctx.check(ovgpu_set_state(ctx.get(), &sv), "ovgpu_set_state");
  ctx.check(ovgpu_set_landmarks(ctx.get(), &lv), "ovgpu_set_landmarks");
  ctx.check(ovgpu_set_features(ctx.get(), &fv), "ovgpu_set_features");
  if (any_aruco) // rows come back scaled to _options_slam.sigma_pix, so R_big below stays isotropic
    ctx.check(ovgpu_set_feature_options(ctx.get(), f_sigma.data(), f_mult.data()), "ovgpu_set_feature_options");
  const int F = fv.F;
  std::vector<int32_t> status(F);
  int32_t rows = 0;
  ovgpu_update_stats stats;
#ifdef OVGPU_SHIM_MODE_B
  // the update is applied on the device; landmark values follow from dx like every other variable (Landmark::update = Vec::update)
  std::vector<double> dx_dev((size_t)sv.N), P_dev((size_t)sv.N * sv.N);
  ctx.check(ovgpu_slam_update(ctx.get(), lm_index.data(), status.data(), nullptr, nullptr, dx_dev.data(), P_dev.data(), nullptr, &stats), "ovgpu_slam_update");
  rows = stats.n_rows;
#else
  const int Dmax = 6 * sv.C + 14 * sv.K + 3 * lv.L; // upper bound (a single-depth landmark has one column)
  std::vector<int32_t> col_cov(Dmax);
  std::vector<double> H((size_t)Dmax * Dmax), r(Dmax);
  int32_t D = 0;
  ctx.check(ovgpu_slam_compress(ctx.get(), lm_index.data(), status.data(), nullptr, nullptr, &D, &rows, col_cov.data(), H.data(), r.data(), &stats),
            "ovgpu_slam_compress");
#endif
  // ---- side effects (UpdaterSLAM.cpp:410-420, :452-454): rejected tracks erased and flagged; the fail count of a non-ArUco landmark bumped
  size_t f = 0;
  for (auto it = feature_vec.begin(); it != feature_vec.end(); f++) {
    (*it)->to_delete = true;
    if (status[f] == OVGPU_FEAT_CHI2_REJECTED) {
      if ((int)(*it)->featid >= state->_options.max_aruco_features) state->_features_SLAM.at((*it)->featid)->update_fail_count++; // :409-416
      it = feature_vec.erase(it);
    } else {
      ++it;
    }
  }
  if (rows < 1) return; // :456-458
#ifdef OVGPU_SHIM_MODE_B
  ovgpu_shim::StateAccess::apply_update(*state, P_dev.data(), dx_dev.data(), sv.N); // StateHelper.cpp:166-196
#else
  ovgpu_shim::ekf_update_with(state, var_of_cov, col_cov.data(), D, rows, H.data(), r.data(), opt_slam.sigma_pix_sq); // :470
#endif
}
} // namespace

void UpdaterSLAM::update(std::shared_ptr<State> state, std::vector<std::shared_ptr<Feature>> &feature_vec) {
  if (feature_vec.empty()) return; // :256-257
  update_all(_options_slam, _options_aruco, state, feature_vec);
}
