// stand-in for ov_core/src/cam/CamEqui.h (TEST INFRASTRUCTURE)
#pragma once
#include "CamBase.h"
namespace ov_core {
class CamEqui : public CamBase {};
} // namespace ov_core
