// stand-in for ov_core/src/cam/CamBase.h (TEST INFRASTRUCTURE)
#pragma once
#include <Eigen/Eigen>
namespace ov_core {
class CamBase {
public:
  virtual ~CamBase() {}
  virtual void set_value(const Eigen::MatrixXd &) {}
};
} // namespace ov_core
