// ovgpu_zupt.h — the linear algebra of ov_msckf::UpdaterZeroVelocity::try_update (ov_msckf/src/update/UpdaterZeroVelocity.cpp,
// rpng/open_vins v2.7) on the device, for a mode-B build (ovgpu_state_access.h): the function keeps selecting IMU readings,
// filling H / res (:100-181), compressing them (:183 — see below) and taking its disparity / velocity decision (:210-247); the
// calls on the covariance are replaced:
//
//   :183  UpdaterHelper::measurement_compress_inplace(H, res)            STAYS the reference's own call
//   :193-198  get_marginal_covariance + Q_bias, S, chi2               ->  ovgpu_shim::zupt_chi2(...)
//   :268-274  StateHelper::EKFPropagation(state, Phi_order, ...)      ->  ovgpu_shim::zupt_apply(...)
//   :277  StateHelper::EKFUpdate(state, Hx_order, H, res, R)              (same call)
//
// Why the compression stays on the host: the stacked ZUPT Jacobian has rank 6 (every IMU sample contributes the same 6 x 9 block),
// so three rows of the 9 x 9 compressed system depend on the ELIMINATION ORDER; the chi2 of :198 sums their squares.  The reference's
// Givens sweep on a 9-column system costs microseconds; running it keeps the accept / reject decision of :241 bit-compatible, which
// the device's Householder compression (round 2) did not.
// INTEGRATION.md shows the patch.  No Eigen decomposition is used here: the 9 x 9 (or 12 x 12) chi2 solve is a plain Cholesky.
#pragma once
#include "ovgpu_shim_common.h"
#include "ovgpu_state_access.h"

namespace ovgpu_shim {

struct ZuptPending { // what zupt_apply needs from zupt_chi2 of the same call
  Context *ctx = nullptr;
  std::vector<int32_t> cols;    // covariance index of every column of H
  std::vector<double> H, r;     // compressed system, row-major rows x cols
  int rows = 0, N = 0;
};

// H [rows x h] over the variables of Hx_order, res [rows]: the system AFTER the reference's measurement_compress_inplace (:183);
// Q_bias: 6 x 6 added to the (bg, ba) block of the marginal covariance when model_time_varying_bias (variables 1 and 2 of
// Hx_order).  Returns res^T (H P_marg H^T + noise_multiplier I)^-1 res (:197-198) with P_marg read from the device-resident state.
template <class UpdaterOptionsT>
inline double zupt_chi2(const std::shared_ptr<ov_msckf::State> &state, const std::vector<std::shared_ptr<ov_type::Type>> &Hx_order,
                        const Eigen::MatrixXd &H, const Eigen::VectorXd &res, const Eigen::MatrixXd &Q_bias, bool model_time_varying_bias,
                        double noise_multiplier, const UpdaterOptionsT &options, ZuptPending &pend) {
  const StateSnapshot snap(state);
  ov_core::FeatureInitializerOptions fo;
  Context &cx = context_for(make_options(options, fo, state->_options, OVGPU_REP_GLOBAL_3D));
  const ovgpu_state_view sv = snap.fs.view();
  cx.check(ovgpu_set_state(cx.get(), &sv), "ovgpu_set_state");
  const int rows = (int)H.rows(), h = (int)H.cols();
  pend.ctx = &cx, pend.N = sv.N, pend.cols.clear();
  for (const auto &v : Hx_order)
    for (int i = 0; i < v->size(); i++) pend.cols.push_back(v->id() + i);
  pend.H.assign((size_t)std::max(rows, 1) * h, 0.0), pend.r.assign(std::max(rows, 1), 0.0);
  for (int i = 0; i < rows; i++) {
    pend.r[i] = res(i);
    for (int j = 0; j < h; j++) pend.H[(size_t)i * h + j] = H(i, j);
  }
  pend.rows = rows;
  if (rows < 1) return 0.0; // :184-186
  std::vector<double> Pm((size_t)h * h);
  cx.check(ovgpu_state_marginal_covariance(cx.get(), h, pend.cols.data(), Pm.data()), "ovgpu_state_marginal_covariance");
  if (model_time_varying_bias)
    for (int i = 0; i < 6; i++)
      for (int j = 0; j < 6; j++) Pm[(size_t)(3 + i) * h + 3 + j] += Q_bias(i, j); // :194-196
  // S = H P H^T + mult I, chi2 = r^T S^-1 r by a Cholesky of S (rows <= h <= 12)
  std::vector<double> T((size_t)rows * h, 0.0), S((size_t)rows * rows, 0.0), y(rows);
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < h; j++) {
      double a = 0.0;
      for (int k = 0; k < h; k++) a += pend.H[(size_t)i * h + k] * Pm[(size_t)k * h + j];
      T[(size_t)i * h + j] = a;
    }
  for (int i = 0; i < rows; i++)
    for (int j = 0; j < rows; j++) {
      double a = i == j ? noise_multiplier : 0.0;
      for (int k = 0; k < h; k++) a += T[(size_t)i * h + k] * pend.H[(size_t)j * h + k];
      S[(size_t)i * rows + j] = a;
    }
  double chi2 = 0.0;
  for (int i = 0; i < rows; i++) { // L L^T = S in place (lower), y = L^-1 r
    for (int j = 0; j <= i; j++) {
      double a = S[(size_t)i * rows + j];
      for (int k = 0; k < j; k++) a -= S[(size_t)i * rows + k] * S[(size_t)j * rows + k];
      S[(size_t)i * rows + j] = i == j ? std::sqrt(a) : a / S[(size_t)j * rows + j];
    }
    double a = pend.r[i];
    for (int k = 0; k < i; k++) a -= S[(size_t)i * rows + k] * y[k];
    y[i] = a / S[(size_t)i * rows + i];
    chi2 += y[i] * y[i];
  }
  return chi2;
}

// :268-277: the bias random walk (Phi = I, Q_bias on the contiguous block [bg, ba]) and the update with R = noise_multiplier I,
// applied on the device; the posterior goes back through StateAccess (needs the friend line of ovgpu_state_access.h).
inline void zupt_apply(const std::shared_ptr<ov_msckf::State> &state, const std::shared_ptr<ov_type::Type> &bg, const Eigen::MatrixXd &Q_bias,
                       bool model_time_varying_bias, double noise_multiplier, const ZuptPending &pend) {
  Context &cx = *pend.ctx;
  if (model_time_varying_bias) {
    double Phi[36], Q[36];
    int32_t ids[6];
    for (int i = 0; i < 6; i++) {
      ids[i] = bg->id() + i; // bg, ba are adjacent in the IMU block (IMU.h: q p v bg ba)
      for (int j = 0; j < 6; j++) Phi[6 * i + j] = i == j ? 1.0 : 0.0, Q[6 * i + j] = Q_bias(i, j);
    }
    cx.check(ovgpu_state_propagate(cx.get(), bg->id(), 6, 6, ids, Phi, Q), "ovgpu_state_propagate");
  }
  std::vector<double> dx(pend.N), P((size_t)pend.N * pend.N);
  cx.check(ovgpu_ekf_update(cx.get(), pend.rows, (int)pend.cols.size(), pend.cols.data(), pend.H.data(), pend.r.data(), noise_multiplier, dx.data(), P.data()),
           "ovgpu_ekf_update");
  StateAccess::apply_update(*state, P.data(), dx.data(), pend.N);
}

} // namespace ovgpu_shim
