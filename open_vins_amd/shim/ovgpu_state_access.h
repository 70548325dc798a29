// ovgpu_state_access.h — the one piece of OpenVINS a mode-B integration has to open up.
//
// State::_Cov and State::_variables are private with `friend class StateHelper;` (ov_msckf/src/state/State.h:182-192).  The shims
// that let the GPU apply the EKF update itself (mode B: dx and P' come back, no StateHelper::EKFUpdate on the host) write
// them through this struct; the maintainer adds ONE line to State.h, next to the existing friend declaration:
//
//     friend struct ovgpu_shim::StateAccess;
//
// (and `namespace ovgpu_shim { struct StateAccess; }` before the class).  Without it the shims build in mode A
// (compressed (H, r) -> the stock StateHelper::EKFUpdate), which needs no change to the reference at all.
#pragma once
#include <Eigen/Eigen>
#include <memory>
#include <vector>

#include "state/State.h"

namespace ovgpu_shim {
struct StateAccess {
  // State::_Cov for an updater's write-back (the resident-covariance build is told: the device's copy is stale from here on) ...
  static Eigen::MatrixXd &cov(ov_msckf::State &s) {
    host_wrote_covariance(s);
    return s._Cov;
  }
  // ... and for the residency code itself (sizes, downloads: ovgpu_resident_cov.h, StateHelper_resident.cpp)
  static Eigen::MatrixXd &cov_raw(ov_msckf::State &s) { return s._Cov; }
  static std::vector<std::shared_ptr<ov_type::Type>> &variables(ov_msckf::State &s) { return s._variables; }

  // StateHelper::EKFUpdate's tail (StateHelper.cpp:166-196) with the numbers computed on the device: P' row-major N x N, dx N.
  // The negative-diagonal check of :171-182 has already happened on the device (OVGPU_ERR_NEGATIVE_DIAGONAL).
  static void apply_update(ov_msckf::State &s, const double *P_rowmajor, const double *dx, int N) {
    host_wrote_covariance(s);
    s._Cov = Eigen::Map<const Eigen::Matrix<double, Eigen::Dynamic, Eigen::Dynamic, Eigen::RowMajor>>(P_rowmajor, N, N);
    const Eigen::Map<const Eigen::VectorXd> d(dx, N);
    for (auto &var : s._variables) var->update(d.segment(var->id(), var->size())); // :185-187
    refresh_cameras(s);
  }
  // an updater's write-back into State::_Cov: in the resident-covariance build (ovgpu_resident_cov.h defines the hook) the device's copy is stale from here on
  static void host_wrote_covariance(ov_msckf::State &s);
  // the same with the covariance left where it is (the resident-covariance mode: P' stays on the device)
  static void apply_dx(ov_msckf::State &s, const double *dx, int N) {
    const Eigen::Map<const Eigen::VectorXd> d(dx, N);
    for (auto &var : s._variables) var->update(d.segment(var->id(), var->size()));
    refresh_cameras(s);
  }
  // the camera objects carry the intrinsics as well and the trackers undistort through them (StateHelper.cpp:191-196)
  static void refresh_cameras(ov_msckf::State &s) {
    if (!s._options.do_calib_camera_intrinsics) return;
    for (auto const &calib : s._cam_intrinsics) s._cam_intrinsics_cameras.at(calib.first)->set_value(calib.second->value());
  }
};
// (ovgpu_shim_common.h: StateSnapshot reads State::_Cov directly when it is called from inside StateHelper's wrappers)
} // namespace ovgpu_shim

#ifdef OVGPU_SHIM_RESIDENT_COV
#include "ovgpu_resident_cov.h" // (defines StateAccess::host_wrote_covariance)
#else
inline void ovgpu_shim::StateAccess::host_wrote_covariance(ov_msckf::State &) {}
#endif
