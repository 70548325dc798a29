// stand-in for ov_core/src/cam/CamRadtan.h (TEST INFRASTRUCTURE)
#pragma once
#include "CamBase.h"
namespace ov_core {
class CamRadtan : public CamBase {};
} // namespace ov_core
