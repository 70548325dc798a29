// stand-in for ov_core/src/types/Landmark.h:35-120 (TEST INFRASTRUCTURE)
#pragma once
#include "LandmarkRepresentation.h"
#include "Vec.h"
namespace ov_type {
class Landmark : public Vec {
public:
  Landmark(int dim) : Vec(dim) {}
  size_t _featid = 0;
  int _unique_camera_id = -1;
  int _anchor_cam_id = -1;
  double _anchor_clone_timestamp = -1;
  bool has_had_anchor_change = false;
  bool should_marg = false;
  int update_fail_count = 0;
  Eigen::Vector3d uv_norm_zero;
  Eigen::Vector3d uv_norm_zero_fej;
  LandmarkRepresentation::Representation _feat_representation = LandmarkRepresentation::GLOBAL_3D;
  Eigen::Matrix<double, 3, 1> get_xyz(bool) const { return Eigen::Matrix<double, 3, 1>(); }
  void set_from_xyz(Eigen::Matrix<double, 3, 1>, bool) {}
};
} // namespace ov_type
