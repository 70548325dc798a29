// stand-in for ov_msckf/src/state/StateOptions.h:35-92 (TEST INFRASTRUCTURE): the fields the update path reads
#pragma once
#include "types/LandmarkRepresentation.h"
namespace ov_msckf {
struct StateOptions {
  bool do_fej = true;
  bool do_calib_camera_pose = false, do_calib_camera_intrinsics = false, do_calib_camera_timeoffset = false;
  int max_clone_size = 11, max_slam_features = 25, max_slam_in_update = 1000, max_msckf_in_update = 1000, max_aruco_features = 1024;
  int num_cameras = 1;
  ov_type::LandmarkRepresentation::Representation feat_rep_msckf = ov_type::LandmarkRepresentation::GLOBAL_3D;
  ov_type::LandmarkRepresentation::Representation feat_rep_slam = ov_type::LandmarkRepresentation::GLOBAL_3D;
  ov_type::LandmarkRepresentation::Representation feat_rep_aruco = ov_type::LandmarkRepresentation::GLOBAL_3D;
};
} // namespace ov_msckf
