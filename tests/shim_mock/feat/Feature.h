// stand-in for ov_core/src/feat/Feature.h:39-98 (TEST INFRASTRUCTURE)
#pragma once
#include <Eigen/Eigen>
#include <unordered_map>
#include <vector>
namespace ov_core {
class Feature {
public:
  size_t featid = 0;
  bool to_delete = false;
  std::unordered_map<size_t, std::vector<Eigen::VectorXf>> uvs;
  std::unordered_map<size_t, std::vector<Eigen::VectorXf>> uvs_norm;
  std::unordered_map<size_t, std::vector<double>> timestamps;
  int anchor_cam_id = -1;
  double anchor_clone_timestamp = -1;
  Eigen::Vector3d p_FinA;
  Eigen::Vector3d p_FinG;
  void clean_old_measurements(const std::vector<double> &) {}
};
} // namespace ov_core
