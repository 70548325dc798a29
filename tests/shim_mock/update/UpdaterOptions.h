// stand-in for ov_msckf/src/update/UpdaterOptions.h:32-48 (TEST INFRASTRUCTURE)
#pragma once
namespace ov_msckf {
struct UpdaterOptions {
  double chi2_multipler = 5;
  double sigma_pix = 1;
  double sigma_pix_sq = 1;
};
} // namespace ov_msckf
