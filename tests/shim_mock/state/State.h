// stand-in for ov_msckf/src/state/State.h:41-196 (TEST INFRASTRUCTURE)
#pragma once
#include <Eigen/Eigen>
#include <map>
#include <memory>
#include <unordered_map>
#include <vector>
#include "StateOptions.h"
#include "cam/CamBase.h"
#include "types/Landmark.h"
#include "types/PoseJPL.h"
#include "types/Type.h"
#include "types/Vec.h"
#ifdef OVGPU_SHIM_MODE_B
namespace ovgpu_shim { struct StateAccess; } // the forward declaration INTEGRATION.md asks the maintainer to add
#endif
namespace ov_msckf {
class StateHelper;
class State {
public:
  double margtimestep() { return _clones_IMU.empty() ? -1 : _clones_IMU.begin()->first; }
  int max_covariance_size() { return (int)_Cov.rows(); }
  double _timestamp = -1;
  StateOptions _options;
  std::map<double, std::shared_ptr<ov_type::PoseJPL>> _clones_IMU;
  std::unordered_map<size_t, std::shared_ptr<ov_type::Landmark>> _features_SLAM;
  std::shared_ptr<ov_type::Vec> _calib_dt_CAMtoIMU;
  std::unordered_map<size_t, std::shared_ptr<ov_type::PoseJPL>> _calib_IMUtoCAM;
  std::unordered_map<size_t, std::shared_ptr<ov_type::Vec>> _cam_intrinsics;
  std::unordered_map<size_t, std::shared_ptr<ov_core::CamBase>> _cam_intrinsics_cameras;
private:
  friend class StateHelper;
#ifdef OVGPU_SHIM_MODE_B
  friend struct ovgpu_shim::StateAccess; // the ONE line mode B adds to the reference
#endif
  Eigen::MatrixXd _Cov;
  std::vector<std::shared_ptr<ov_type::Type>> _variables;
};
} // namespace ov_msckf
