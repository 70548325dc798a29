// stand-in for ov_msckf/src/update/UpdaterMSCKF.h:40-82 (TEST INFRASTRUCTURE)
#pragma once
#include <Eigen/Eigen>
#include <map>
#include <memory>
#include <vector>
#include "UpdaterOptions.h"
#include "feat/FeatureInitializerOptions.h"
namespace ov_core {
class Feature;
class FeatureInitializer;
} // namespace ov_core
namespace ov_msckf {
class State;
class UpdaterMSCKF {
public:
  UpdaterMSCKF(UpdaterOptions &options, ov_core::FeatureInitializerOptions &feat_init_options);
  void update(std::shared_ptr<State> state, std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec);
protected:
  UpdaterOptions _options;
  std::shared_ptr<ov_core::FeatureInitializer> initializer_feat;
  std::map<int, double> chi_squared_table;
};
} // namespace ov_msckf
