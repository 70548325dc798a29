// Drop-in replacement for ov_msckf/src/update/UpdaterMSCKF.cpp (rpng/open_vins v2.7).
//
// Same class, same constructor and method signatures, same side effects on `state` and `feature_vec`; the Eigen
// loops of UpdaterMSCKF.cpp:97-283 are replaced by calls into libovgpu.so.  Compiled INSIDE the open_vins tree (it includes the
// reference's headers): swap it for the original in ov_msckf/CMakeLists.txt and link libovgpu — see INTEGRATION.md.
// Mode A (default): the GPU returns the compressed system, the stock StateHelper::EKFUpdate applies it (State::_Cov / _variables
// are private to StateHelper, State.h:182-192).  -DOVGPU_SHIM_MODE_B: the device applies the update, dx and P' are written back
// through ovgpu_shim::StateAccess (one friend line in State.h).
#include "UpdaterMSCKF.h"

#include "UpdaterHelper.h"
#include "utils/print.h"

#include "ovgpu_shim_common.h"
#ifdef OVGPU_SHIM_MODE_B
#include "ovgpu_state_access.h"
#endif
#ifdef OVGPU_SHIM_RESIDENT_COV
#ifndef OVGPU_SHIM_MODE_B
#error "OVGPU_SHIM_RESIDENT_COV needs OVGPU_SHIM_MODE_B: the device applies the update to the covariance it holds"
#endif
#include "ovgpu_resident_cov.h" // opt-in: the covariance lives in the library's context between frames (with shim/StateHelper_resident.cpp)
#endif

using namespace ov_core;
using namespace ov_type;
using namespace ov_msckf;

#ifdef OVGPU_SHIM_TIMING
// test builds: where one update's host time goes (tests/dropin_probe.py `time` reads the sums through ovgpu_shim_update_laps)
#include <chrono>
static double g_laps[9];
extern "C" void ovgpu_shim_update_laps(double *out9, int reset) {
  for (int i = 0; i < 9; i++) out9[i] = g_laps[i], g_laps[i] = reset ? 0.0 : g_laps[i];
}
struct Lap {
  std::chrono::steady_clock::time_point t = std::chrono::steady_clock::now();
  void operator()(int i) {
    const auto n = std::chrono::steady_clock::now();
    g_laps[i] += std::chrono::duration<double, std::milli>(n - t).count(), t = n;
  }
};
#define OVGPU_LAP(i) lap(i)
#else
#define OVGPU_LAP(i) (void)0
#endif


UpdaterMSCKF::UpdaterMSCKF(UpdaterOptions &options, FeatureInitializerOptions &feat_init_options) : _options(options) {
  _options.sigma_pix_sq = std::pow(_options.sigma_pix, 2);
  // kept so that config() and the SLAM updater keep working; the chi2 table lives in the library
  initializer_feat = std::shared_ptr<FeatureInitializer>(new FeatureInitializer(feat_init_options));
}

void UpdaterMSCKF::update(std::shared_ptr<State> state, std::vector<std::shared_ptr<Feature>> &feature_vec) {
  if (feature_vec.empty()) return; // UpdaterMSCKF.cpp:61-62
#ifdef OVGPU_SHIM_TIMING
  Lap lap;
  g_laps[8] += 1.0;
#endif
#ifdef OVGPU_SHIM_RESIDENT_COV
  const ovgpu_shim::StateSnapshot snap(state, /*with_cov=*/false); // values, ids, order: the covariance is (or will be, below) on the device
#else
  const ovgpu_shim::StateSnapshot snap(state);
#endif
  const ovgpu_shim::CloneIndex clones(snap.fs.clone_times);
  OVGPU_LAP(0); // snapshot

  // ---- 1. clean + flatten the tracks (UpdaterMSCKF.cpp:71-93)
  static thread_local ovgpu_shim::FlatFeatures ff; // reused from update to update: a fresh 3.5 MB of buffers per call costs more in page faults than the flattening itself
  ovgpu_shim::clean_flatten_batch(feature_vec, snap, clones, ff, 2); // :84-92: a track with fewer than two observations in the window is flagged and dropped
  if (feature_vec.empty()) return;

  OVGPU_LAP(1); // clean + flatten
  // ---- 2..5 on the GPU: triangulate, Jacobians, nullspace, chi2 gate, stack, compress (options re-read on every call)
  LandmarkRepresentation::Representation rep = state->_options.feat_rep_msckf; // :180-183: the single depth maps to the MSCKF inverse depth
  if (rep == LandmarkRepresentation::Representation::ANCHORED_INVERSE_DEPTH_SINGLE) rep = LandmarkRepresentation::Representation::ANCHORED_MSCKF_INVERSE_DEPTH;
  ovgpu_shim::Context &ctx = ovgpu_shim::context_for(ovgpu_shim::make_options(_options, initializer_feat->config(), state->_options, (int)rep));
  const ovgpu_state_view sv = snap.fs.view();
#ifdef OVGPU_SHIM_RESIDENT_COV
  // the state is resident: uploaded once (and again after anything on the host wrote the covariance: another updater's host path,
  // an initialisation), kept current by StateHelper's device-side wrappers in between
  // (the record of THIS State; its lock is held to the end of the update: StateHelper's wrappers on other threads — a marginal covariance
  //  for the odometry thread, Propagator.cpp:147 — reach the same context and wait)
  ovgpu_shim::ResidentCov &rcov = ovgpu_shim::ResidentCov::of(state);
  const ovgpu_shim::ResidentCov::Guard rguard = rcov.lock();
  rcov.attach(ctx, *state);
  rcov.ensure_device(state);
#else
  ctx.check(ovgpu_set_state(ctx.get(), &sv), "ovgpu_set_state");
#endif
  OVGPU_LAP(2); // state hand-over
  const ovgpu_features_view fv = ff.view();
  ctx.check(ovgpu_set_features(ctx.get(), &fv), "ovgpu_set_features");
  const int F = fv.F;
  OVGPU_LAP(3); // track hand-over
  std::vector<int32_t> status(F), anchor(F);
  std::vector<double> pA(3 * (size_t)F), pG(3 * (size_t)F);
  int32_t rows = 0;
  ovgpu_update_stats stats;
#ifdef OVGPU_SHIM_MODE_B
  // the device applies the update itself (Gram matrix of the prior-whitened stack on the matrix cores); p_FinA / the anchors
  // are read back from the same triangulation afterwards
#ifdef OVGPU_SHIM_RESIDENT_COV
  std::vector<double> dx_dev((size_t)sv.N);
  ctx.check(ovgpu_msckf_update(ctx.get(), status.data(), nullptr, nullptr, pG.data(), dx_dev.data(), nullptr, &stats), "ovgpu_msckf_update"); // P' stays where it is
#else
  std::vector<double> dx_dev((size_t)sv.N), P_dev((size_t)sv.N * sv.N);
  ctx.check(ovgpu_msckf_update(ctx.get(), status.data(), nullptr, nullptr, pG.data(), dx_dev.data(), P_dev.data(), &stats), "ovgpu_msckf_update");
#endif
  OVGPU_LAP(4); // the call
  ctx.check(ovgpu_get_triangulation(ctx.get(), pA.data(), nullptr, anchor.data()), "ovgpu_get_triangulation");
  rows = stats.n_rows;
#else
  const int Dmax = 6 * sv.C + 14 * sv.K;
  std::vector<int32_t> col_cov(Dmax);
  std::vector<double> H((size_t)Dmax * Dmax), r(Dmax);
  int32_t D = 0;
  ctx.check(ovgpu_msckf_compress(ctx.get(), status.data(), nullptr, nullptr, pG.data(), &D, &rows, col_cov.data(), H.data(), r.data(), &stats),
            "ovgpu_msckf_compress");
  ctx.check(ovgpu_get_triangulation(ctx.get(), pA.data(), nullptr, anchor.data()), "ovgpu_get_triangulation"); // ONE triangulation serves both
#endif

  OVGPU_LAP(5); // triangulation read-back
  // ---- side effects on the features: triangulation results, erase the rejected, flag the used
  size_t used = 0; // (one pass: the accepted tracks move up, in order — erasing the rejected one by one is quadratic in the batch)
  for (size_t f = 0; f < feature_vec.size(); f++) {
    std::shared_ptr<Feature> &ft = feature_vec[f];
    if (f + 4 < feature_vec.size()) __builtin_prefetch(feature_vec[f + 4].get(), 1);
    ovgpu_shim::write_triangulation(*ft, snap, ff, anchor[f], &pA[3 * f], &pG[3 * f]);
    ft->to_delete = true; // :137, :226, :262 — every feature that reached this point is flagged
    if (status[f] != OVGPU_FEAT_USED) continue;
    if (used != f) feature_vec[used] = std::move(ft);
    used++;
  }
  feature_vec.resize(used);
  OVGPU_LAP(6); // side effects on the tracks
  if (rows < 1) return; // :266-268 / :276-278
#if defined(OVGPU_SHIM_RESIDENT_COV)
  ovgpu_shim::StateAccess::apply_dx(*state, dx_dev.data(), sv.N); // StateHelper.cpp:185-196: the values follow on the host, the covariance was updated in place
  rcov.device_written();
#elif defined(OVGPU_SHIM_MODE_B)
  ovgpu_shim::StateAccess::apply_update(*state, P_dev.data(), dx_dev.data(), sv.N); // StateHelper.cpp:166-196
#else
  ovgpu_shim::ekf_update_with(state, snap.var_of_cov, col_cov.data(), D, rows, H.data(), r.data(), _options.sigma_pix_sq); // :280-285
#endif
  OVGPU_LAP(7); // state write-back
}
