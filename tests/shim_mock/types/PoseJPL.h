// stand-in for ov_core/src/types/PoseJPL.h:35-150 (TEST INFRASTRUCTURE)
#pragma once
#include "Type.h"
namespace ov_type {
class PoseJPL : public Type {
public:
  PoseJPL() : Type(6) {}
  void update(const Eigen::VectorXd &) override {}
  Eigen::Matrix<double, 3, 3> Rot() const { return Eigen::Matrix<double, 3, 3>(); }
  Eigen::Matrix<double, 4, 1> quat() const { return Eigen::Matrix<double, 4, 1>(); }
  Eigen::Matrix<double, 4, 1> quat_fej() const { return Eigen::Matrix<double, 4, 1>(); }
  Eigen::Matrix<double, 3, 1> pos() const { return Eigen::Matrix<double, 3, 1>(); }
  Eigen::Matrix<double, 3, 1> pos_fej() const { return Eigen::Matrix<double, 3, 1>(); }
};
} // namespace ov_type
