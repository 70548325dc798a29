// stand-in for ov_core/src/types/Vec.h (TEST INFRASTRUCTURE)
#pragma once
#include "Type.h"
namespace ov_type {
class Vec : public Type {
public:
  Vec(int dim) : Type(dim) {}
  void update(const Eigen::VectorXd &) override {}
};
} // namespace ov_type
