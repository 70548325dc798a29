// stand-in for ov_core/src/feat/FeatureInitializerOptions.h:33-69 (TEST INFRASTRUCTURE)
#pragma once
namespace ov_core {
struct FeatureInitializerOptions {
  bool triangulate_1d = false;
  bool refine_features = true;
  int max_runs = 5;
  double init_lamda = 1e-3, max_lamda = 1e10, min_dx = 1e-6, min_dcost = 1e-6, lam_mult = 10;
  double min_dist = 0.10, max_dist = 60, max_baseline = 40, max_cond_number = 10000;
};
} // namespace ov_core
