// Drop-in replacement for ov_msckf/src/update/UpdaterMSCKF.cpp (rpng/open_vins v2.7).
//
// Same class, same constructor and method signatures, same side effects on `state` and `feature_vec`
// (SURVEY.md §8b); the Eigen loops of UpdaterMSCKF.cpp:97-283 are replaced by calls into libovgpu.so.
// This file is compiled INSIDE the open_vins tree (it includes the reference's headers), e.g. by swapping it
// for the original in ov_msckf/CMakeLists.txt and linking libovgpu — see INTEGRATION.md.  It is "mode A":
// the GPU returns the compressed system and the stock StateHelper::EKFUpdate applies it, because
// State::_Cov / _variables are private to StateHelper (State.h:182-192).
#include "UpdaterMSCKF.h"

#include "UpdaterHelper.h"
#include "feat/Feature.h"
#include "feat/FeatureInitializer.h"
#include "state/State.h"
#include "state/StateHelper.h"
#include "types/LandmarkRepresentation.h"
#include "utils/print.h"

#include "ovgpu.h"
#include "ovgpu_flatten.h"
#ifdef OVGPU_SHIM_MODE_B
#include "ovgpu_state_access.h" // needs `friend struct ovgpu_shim::StateAccess;` in State.h
#endif

using namespace ov_core;
using namespace ov_type;
using namespace ov_msckf;

namespace {
std::unique_ptr<ovgpu_shim::Context> g_ctx; // one context per process: VioManager owns one UpdaterMSCKF (VioManager.cpp:155)

ovgpu_options make_options(const UpdaterOptions &u, const FeatureInitializerOptions &f, const StateOptions &s) {
  ovgpu_options o;
  ovgpu_default_options(&o);
  o.chi2_multipler = u.chi2_multipler, o.sigma_pix = u.sigma_pix;
  o.triangulate_1d = f.triangulate_1d, o.refine_features = f.refine_features, o.max_runs = f.max_runs;
  o.init_lamda = f.init_lamda, o.max_lamda = f.max_lamda, o.min_dx = f.min_dx, o.min_dcost = f.min_dcost, o.lam_mult = f.lam_mult;
  o.min_dist = f.min_dist, o.max_dist = f.max_dist, o.max_baseline = f.max_baseline, o.max_cond_number = f.max_cond_number;
  o.do_fej = s.do_fej, o.do_calib_camera_pose = s.do_calib_camera_pose, o.do_calib_camera_intrinsics = s.do_calib_camera_intrinsics;
  // UpdaterMSCKF.cpp:180-183: the single-depth representation is mapped to the MSCKF inverse depth one
  LandmarkRepresentation::Representation rep = s.feat_rep_msckf;
  if (rep == LandmarkRepresentation::Representation::ANCHORED_INVERSE_DEPTH_SINGLE) rep = LandmarkRepresentation::Representation::ANCHORED_MSCKF_INVERSE_DEPTH;
  o.feat_rep_msckf = (int32_t)rep;
  return o;
}
} // namespace

UpdaterMSCKF::UpdaterMSCKF(UpdaterOptions &options, FeatureInitializerOptions &feat_init_options) : _options(options) {
  _options.sigma_pix_sq = std::pow(_options.sigma_pix, 2);
  // kept so that config() and the SLAM updater keep working; the chi2 table lives in the library
  initializer_feat = std::shared_ptr<FeatureInitializer>(new FeatureInitializer(feat_init_options));
}

void UpdaterMSCKF::update(std::shared_ptr<State> state, std::vector<std::shared_ptr<Feature>> &feature_vec) {
  if (feature_vec.empty()) return; // UpdaterMSCKF.cpp:61-62

  // ---- state snapshot (clone order = iteration order of _clones_IMU = ascending time)
  ovgpu_shim::FlatState fs;
  std::vector<std::shared_ptr<Type>> var_of_cov; // variables that can own Jacobian columns
  for (const auto &c : state->_clones_IMU) {
    const Eigen::Vector4d q = c.second->quat(), qf = c.second->quat_fej();
    const Eigen::Vector3d p = c.second->pos(), pf = c.second->pos_fej();
    fs.add_clone(c.first, q.data(), p.data(), qf.data(), pf.data(), c.second->id());
    var_of_cov.push_back(c.second);
  }
  std::vector<size_t> cam_ids; // camera index <-> reference camera id
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
  const Eigen::MatrixXd P = StateHelper::get_full_covariance(state); // symmetric: column-major == row-major
  fs.N = (int32_t)P.rows();
  fs.P.assign(P.data(), P.data() + P.size());

  // ---- 1. clean + flatten the tracks (UpdaterMSCKF.cpp:71-93)
  const ovgpu_shim::CloneIndex clones(fs.clone_times);
  ovgpu_shim::FlatFeatures ff;
  auto it0 = feature_vec.begin();
  while (it0 != feature_vec.end()) {
    (*it0)->clean_old_measurements(fs.clone_times);
    int ct_meas = 0;
    for (const auto &pair : (*it0)->timestamps) ct_meas += (int)pair.second.size();
    if (ct_meas < 2) {
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
    it0++;
  }
  if (feature_vec.empty()) return;

  // ---- 2..5 on the GPU: triangulate, Jacobians, nullspace, chi2 gate, stack, compress
  if (!g_ctx) g_ctx.reset(new ovgpu_shim::Context(make_options(_options, initializer_feat->config(), state->_options)));
  const ovgpu_state_view sv = fs.view();
  const ovgpu_features_view fv = ff.view();
  g_ctx->check(ovgpu_set_state(g_ctx->get(), &sv), "ovgpu_set_state");
  g_ctx->check(ovgpu_set_features(g_ctx->get(), &fv), "ovgpu_set_features");
  const int F = fv.F, Dmax = 6 * sv.C + 14 * sv.K;
  std::vector<int32_t> status(F), anchor(F), col_cov(Dmax);
  std::vector<double> pA(3 * F), pG(3 * F), H((size_t)Dmax * Dmax), r(Dmax);
  int32_t D = 0, rows = 0;
  ovgpu_update_stats stats;
  g_ctx->check(ovgpu_triangulate(g_ctx->get(), pA.data(), nullptr, anchor.data(), nullptr), "ovgpu_triangulate"); // Feature::p_FinA / anchor
#ifdef OVGPU_SHIM_MODE_B
  // mode B: the device applies the update itself (Gram matrix of the stack on the matrix cores + the update whitened by the
  // prior: half the time of compress -> EKFUpdate, same dx and P'); dx and P' come back and are written through StateAccess
  std::vector<double> dx_dev((size_t)sv.N), P_dev((size_t)sv.N * sv.N);
  g_ctx->check(ovgpu_msckf_update(g_ctx->get(), status.data(), nullptr, nullptr, pG.data(), dx_dev.data(), P_dev.data(), &stats), "ovgpu_msckf_update");
  rows = stats.n_rows;
#else
  g_ctx->check(ovgpu_msckf_compress(g_ctx->get(), status.data(), nullptr, nullptr, pG.data(), &D, &rows, col_cov.data(), H.data(), r.data(), &stats),
               "ovgpu_msckf_compress");
#endif

  // ---- side effects on the features: triangulation results, erase the rejected, flag the used
  size_t f = 0;
  auto it1 = feature_vec.begin();
  while (it1 != feature_vec.end()) {
    Feature &feat = **it1;
    const int a = anchor[f];
    if (a >= 0) {
      feat.anchor_cam_id = (int)cam_ids[ff.cam_idx[a]];
      feat.anchor_clone_timestamp = ff.meas_time[a];
      feat.p_FinA = Eigen::Map<const Eigen::Vector3d>(&pA[3 * f]);
      feat.p_FinG = Eigen::Map<const Eigen::Vector3d>(&pG[3 * f]);
    }
    feat.to_delete = true; // :137, :226, :262 — every feature that reached this point is flagged
    if (status[f] != OVGPU_FEAT_USED) it1 = feature_vec.erase(it1);
    else it1++;
    f++;
  }
  if (rows < 1) return; // :266-268 / :276-278

#ifdef OVGPU_SHIM_MODE_B
  ovgpu_shim::StateAccess::apply_update(*state, P_dev.data(), dx_dev.data(), sv.N); // StateHelper.cpp:166-195
  return;
#endif
  // ---- 6. the stock EKF update on the compressed system (UpdaterMSCKF.cpp:280-285)
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
  Eigen::MatrixXd R_big = _options.sigma_pix_sq * Eigen::MatrixXd::Identity(rows, rows);
  StateHelper::EKFUpdate(state, Hx_order_big, Hx_big, res_big, R_big);
}
