// ovgpu_featw_tu.hip — the library's THIRD translation unit: the wavefront-per-feature kernel feat::k_feat_w (k_featw.h).
//
// Why a translation unit of its own: the kernel runs one wavefront per SIMD with the 36 accumulator tiles of its gate matrix (288
// registers) next to ~200 registers of operands, i.e. it needs the whole 512-register budget of a wavefront, split freely between the
// architectural and the accumulator file.  By default the AMDGPU back end selects the ACCUMULATOR-register form of every
// v_mfma whose result it can place there; with 36 tiles against 256 accumulator registers four tiles then live in scratch (loaded,
// updated and stored around their products: 132 bytes per lane, and every such access shares the counter the operand prefetch waits
// on — the first build ran 1.23 ms where k_feat_y runs 0.41).  -amdgpu-mfma-vgpr-form lets the register allocator choose the file per
// value: 256 + 190 registers, no scratch.  The switch is per compilation, and no other kernel of the library wants it.
#define OVG_TU_FEATY 1
#include <hip/hip_runtime.h>

#include "k_featw.h"

namespace ovg {
namespace feat {
#define X(NTM, F32) template __global__ void k_feat_w<NTM, F32>(OVG_FEATW_ARGS);
OVG_FEATW_SHAPES(X)
#undef X
} // namespace feat
} // namespace ovg
