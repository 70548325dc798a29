// stand-in for ov_core/src/cam/CamBase.h (TEST INFRASTRUCTURE): set_value :56, undistort_f :89, w / h :170-173
#pragma once
#include <Eigen/Eigen>
namespace ov_core {
class CamBase {
public:
  virtual ~CamBase() {}
  virtual void set_value(const Eigen::MatrixXd &) {}
  virtual Eigen::Vector2f undistort_f(const Eigen::Vector2f &uv_dist) = 0;
  int w() { return _width; }
  int h() { return _height; }

protected:
  int _width = 0, _height = 0;
};
} // namespace ov_core
