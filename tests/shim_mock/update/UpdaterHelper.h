// stand-in for ov_msckf/src/update/UpdaterHelper.h (TEST INFRASTRUCTURE): the shims do not call into it
#pragma once
namespace ov_msckf {
class UpdaterHelper {};
} // namespace ov_msckf
