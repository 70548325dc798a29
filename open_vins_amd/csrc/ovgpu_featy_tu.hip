// ovgpu_featy_tu.hip — the library's SECOND translation unit: the fused per-feature kernels feat::k_feat_y (k_featy.h: the headline
// shape <4, 11, 2> and the 8-wavefront shape <8, 17, 1>, float64 and float32 stack) and feat::k_feat_y_big (k_featy_big.h: gate
// matrices factored block row by block row), compiled with the iterative-ILP machine scheduler
// (-mllvm -amdgpu-sched-strategy=iterative-ilp, Makefile: FEATY_FLAGS).
//
// Why a translation unit of its own: the scheduler strategy is a per-compilation switch, and it pays for THESE kernels only.  They
// are latency bound at two wavefronts per SIMD (256 registers), so a schedule built for instruction-level parallelism instead of
// occupancy hides more of their operand latency; k_gram_il, at one wavefront per SIMD with 367 registers, loses under it (151 -> 215
// vector registers, 0.211 -> 0.217 ms), and the chain kernels (triangulation, the factorisations, the tail) do not move.  Same box,
// alternating bench lines (tools/gpu_ab3.sh, tools/gpu_bitcompare.sh; profiles/r04_late_*):
//   configs[2]  per-feature stage 0.512 -> 0.495 ms, update 1.033 -> 1.008 (a box of the slower kind; 0.437 -> 0.418 on a fast one)
//   configs[1]  0.269 -> 0.260, update 0.615 -> 0.609
//   configs[3] on one GPU (k_feat_y<8, 17, 1>)  6.62 -> 6.47, update 9.76 -> 9.60
//   one rank's share of configs[4] (k_feat_y_big<8, 17>)  5.81 -> 5.66, update 8.80 -> 8.67
// The arithmetic is the same instruction for instruction — no reassociation, no contraction change: the scheduler only orders
// independent instructions — and the builds' outputs were compared BIT FOR BIT over 17 shapes of the parity suite (tools/dev_bitcompare.py:
// 221 arrays — status, chi2, p_FinG, dx, P', the state, mode A's (H, r) — 0 differ; the base build against itself as the control).
// ovgpu_api.hip declares these instantiations `extern template`; the non-template kernels of the headers are compiled there only
// (OVG_TU_FEATY).
#define OVG_TU_FEATY 1
#include <hip/hip_runtime.h>

#include "k_featy.h"
#include "k_featy_big.h"

namespace ovg {
namespace feat {
#define X(NW, TPW, OCC, F32, CB) template __global__ void k_feat_y<NW, TPW, OCC, F32, CB>(OVG_FEATY_ARGS);
OVG_FEATY_SHAPES(X)
#undef X
template __global__ void k_feat_y_big<8, 17, false>(SysParams, int, const double *__restrict__, const int32_t *__restrict__, const double *__restrict__,
    const double *__restrict__, const int32_t *__restrict__, const int32_t *__restrict__, double *);
template __global__ void k_feat_y_big<8, 17, true>(SysParams, int, const double *__restrict__, const int32_t *__restrict__, const double *__restrict__,
    const double *__restrict__, const int32_t *__restrict__, const int32_t *__restrict__, double *);
template __global__ void k_feat_y_big<8, 5, false>(SysParams, int, const double *__restrict__, const int32_t *__restrict__, const double *__restrict__,
    const double *__restrict__, const int32_t *__restrict__, const int32_t *__restrict__, double *);
} // namespace feat
} // namespace ovg
