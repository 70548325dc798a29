// stand-in for ov_msckf/src/state/StateHelper.h:45-240 (TEST INFRASTRUCTURE): the static entry points the shims call
#pragma once
#include <Eigen/Eigen>
#include <memory>
#include <vector>
#include "State.h"
namespace ov_msckf {
class StateHelper {
public:
  static void EKFUpdate(std::shared_ptr<State> state, const std::vector<std::shared_ptr<ov_type::Type>> &H_order, const Eigen::MatrixXd &H,
                        const Eigen::VectorXd &res, const Eigen::MatrixXd &R);
  static Eigen::MatrixXd get_full_covariance(std::shared_ptr<State> state);
  static void marginalize(std::shared_ptr<State> state, std::shared_ptr<ov_type::Type> marg);
};
} // namespace ov_msckf
