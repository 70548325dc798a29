// stand-in for ov_core/src/feat/FeatureInitializer.h:40-159 (TEST INFRASTRUCTURE)
#pragma once
#include <Eigen/Eigen>
#include <memory>
#include <unordered_map>
#include "Feature.h"
#include "FeatureInitializerOptions.h"
namespace ov_core {
class FeatureInitializer {
public:
  struct ClonePose {
    Eigen::Matrix<double, 3, 3> _Rot;
    Eigen::Matrix<double, 3, 1> _pos;
    const Eigen::Matrix<double, 3, 3> &Rot() { return _Rot; }
    const Eigen::Matrix<double, 3, 1> &pos() { return _pos; }
  };
  FeatureInitializer(FeatureInitializerOptions &options) : _options(options) {}
  bool single_triangulation(std::shared_ptr<Feature> feat, std::unordered_map<size_t, std::unordered_map<double, ClonePose>> &clonesCAM);
  bool single_triangulation_1d(std::shared_ptr<Feature> feat, std::unordered_map<size_t, std::unordered_map<double, ClonePose>> &clonesCAM);
  bool single_gaussnewton(std::shared_ptr<Feature> feat, std::unordered_map<size_t, std::unordered_map<double, ClonePose>> &clonesCAM);
  const FeatureInitializerOptions config() { return _options; }
protected:
  FeatureInitializerOptions _options;
};
} // namespace ov_core
