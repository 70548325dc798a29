// stand-in for ov_msckf/src/update/UpdaterSLAM.h:42-110 (TEST INFRASTRUCTURE)
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
namespace ov_type {
class Landmark;
}
namespace ov_msckf {
class State;
class UpdaterSLAM {
public:
  UpdaterSLAM(UpdaterOptions &options_slam, UpdaterOptions &options_aruco, ov_core::FeatureInitializerOptions &feat_init_options);
  void update(std::shared_ptr<State> state, std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec);
  void delayed_init(std::shared_ptr<State> state, std::vector<std::shared_ptr<ov_core::Feature>> &feature_vec);
  void change_anchors(std::shared_ptr<State> state);
protected:
  void perform_anchor_change(std::shared_ptr<State> state, std::shared_ptr<ov_type::Landmark> landmark, double new_anchor_timestamp, size_t new_cam_id);
  UpdaterOptions _options_slam;
  UpdaterOptions _options_aruco;
  std::shared_ptr<ov_core::FeatureInitializer> initializer_feat;
  std::map<int, double> chi_squared_table;
};
} // namespace ov_msckf
