// stand-in for ov_core/src/types/LandmarkRepresentation.h:38-101 (TEST INFRASTRUCTURE, see ../README.md)
#pragma once
namespace ov_type {
class LandmarkRepresentation {
public:
  enum Representation { GLOBAL_3D, GLOBAL_FULL_INVERSE_DEPTH, ANCHORED_3D, ANCHORED_FULL_INVERSE_DEPTH, ANCHORED_MSCKF_INVERSE_DEPTH, ANCHORED_INVERSE_DEPTH_SINGLE, UNKNOWN };
  static inline bool is_relative_representation(Representation r) {
    return r == ANCHORED_3D || r == ANCHORED_FULL_INVERSE_DEPTH || r == ANCHORED_MSCKF_INVERSE_DEPTH || r == ANCHORED_INVERSE_DEPTH_SINGLE;
  }
};
} // namespace ov_type
