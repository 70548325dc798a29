#include "update/UpdaterOptions.h"
#include "ovgpu_zupt.h"
#include "ovgpu_retri.h"
#include <map>
// instantiate the templates the way the reference would call them
double probe(std::shared_ptr<ov_msckf::State> state, std::vector<std::shared_ptr<ov_type::Type>> order, Eigen::MatrixXd &H, Eigen::VectorXd &res, ov_msckf::UpdaterOptions &o) {
  Eigen::MatrixXd Q = Eigen::MatrixXd::Identity(6, 6);
  ovgpu_shim::ZuptPending pend;
  const double chi2 = ovgpu_shim::zupt_chi2(state, order, H, res, Q, true, 10.0, o, pend);
  ovgpu_shim::zupt_apply(state, order[1], Q, true, 10.0, pend);
  std::map<size_t, std::vector<std::pair<float, float>>> obs;
  std::map<size_t, std::vector<size_t>> ids;
  ov_core::FeatureInitializerOptions fo;
  std::unordered_map<size_t, Eigen::Vector3d> pos, uvd;
  ovgpu_shim::retriangulate(state, 0.0, std::vector<int>{0, 1}, obs, ids, o, fo, pos, uvd);
  return chi2;
}
