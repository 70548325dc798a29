// ovgpu_featy_tu.hip — the library's SECOND translation unit: the fused per-feature kernel of the headline shape,
// feat::k_feat_y<4, 11, 2> (k_featy.h; float64 and float32 stack), compiled with the iterative-ILP machine scheduler
// (-mllvm -amdgpu-sched-strategy=iterative-ilp, Makefile: FEATY_FLAGS).
//
// Why a translation unit of its own: the scheduler strategy is a per-compilation switch, and it pays for THIS kernel only.  Same box,
// alternating bench lines at configs[2] (tools/gpu_ab3.sh, late round 4): the whole library under iterative-ilp runs the per-feature
// stage in 0.418 ms against 0.436 (the kernel is latency bound at two wavefronts per SIMD: a schedule built for instruction-level
// parallelism instead of occupancy hides more of its operand latency) but k_gram_il in 0.217 against 0.211 (215 instead of 151
// registers).  The arithmetic is the same instruction for instruction — no reassociation, no contraction change: the scheduler only
// orders independent instructions — and the two builds' outputs are compared BIT FOR BIT over the shapes of the parity suite
// (tools/dev_bitcompare.py).  ovgpu_api.hip declares these two instantiations `extern template`; the non-template kernels of
// the headers are compiled there only (OVG_TU_FEATY).
#define OVG_TU_FEATY 1
#include <hip/hip_runtime.h>

#include "k_featy.h"

namespace ovg {
namespace feat {
template __global__ void k_feat_y<4, 11, 2, false>(SysParams, int, const double *__restrict__, const int32_t *__restrict__, const double *__restrict__,
                                                   const double *__restrict__, const int32_t *__restrict__, const int32_t *__restrict__);
template __global__ void k_feat_y<4, 11, 2, true>(SysParams, int, const double *__restrict__, const int32_t *__restrict__, const double *__restrict__,
                                                  const double *__restrict__, const int32_t *__restrict__, const int32_t *__restrict__);
} // namespace feat
} // namespace ovg
