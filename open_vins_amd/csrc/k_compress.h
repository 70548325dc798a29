// k_compress.h — measurement compression of the stacked system as a tall-skinny Householder QR.
//
//   UpdaterHelper::measurement_compress_inplace   UpdaterHelper.cpp:456-487
//
// The reference runs Givens rotations over all row pairs, a strictly serial O(rows * D^2) sweep.
// Here the stacked [H | r] (rows x LD, LD = D + 1) is reduced by TSQR:
//   leaf   : W workgroups each fold their slice of rows, B rows at a time, into a private
//            D x LD upper-triangular accumulator R_w by Householder reflections;
//   tree   : pairs of accumulators are merged (QR of two stacked triangles) until one is left.
// QR([R_1; R_2; ...]) has the same R^T R and R^T c as QR of the full stack, so the compressed
// system is the reference's up to row signs (SURVEY.md §7 "Tall-skinny QR").
//
// Data layout: one thread owns one COLUMN of the current row block and keeps its B entries in
// registers, so the reflector inner products v^T b_c are thread-local (no cross-lane traffic);
// only the reflector v itself (B doubles) is broadcast through LDS, one barrier per column.
// The accumulator rows live in HBM/L2 and are touched once per step with coalesced accesses.
#pragma once
#include "device_math.h"
#include "ovgpu_types.h"

namespace ovg {

struct QrAppendParams {
  int D, LD;
  double *dst;           // accumulators: dst + wg * dst_wg_stride * (D*LD)
  int64_t dst_wg_stride; // in triangles
  const double *src;     // rows to fold in: src + wg * src_wg_stride (in doubles)
  int64_t src_wg_stride;
  int64_t src_rows_per_wg;
  int64_t src_rows_total; // rows valid from src base over all workgroups (leaf) ; per wg for merges
  int triangular;         // source block is upper triangular (row s has zeros before column s)
  int zero_dst;           // clear the accumulator first (leaf phase)
};

template <int B>
__global__ void __launch_bounds__(512) k_qr_append(QrAppendParams p) {
  __shared__ __attribute__((aligned(16))) double vbuf[2][B + 2];
  const int t = threadIdx.x;
  const int D = p.D, LD = p.LD;
  const bool active = t < LD;
  double *R = p.dst + (size_t)blockIdx.x * p.dst_wg_stride * D * LD;
  const double *src = p.src + (size_t)blockIdx.x * p.src_wg_stride;
  int64_t nrows = p.src_rows_per_wg;
  if (!p.triangular) {
    const int64_t first = (int64_t)blockIdx.x * p.src_rows_per_wg;
    nrows = max((int64_t)0, min(p.src_rows_per_wg, p.src_rows_total - first));
  }
  if (p.zero_dst) {
    for (int e = t; e < D * LD; e += blockDim.x) R[e] = 0.0;
    __syncthreads();
  }
  int par = 0;
  for (int64_t blk = 0; blk < nrows; blk += B) {
    double b[B];
#pragma unroll
    for (int i = 0; i < B; i++) b[i] = (active && blk + i < nrows) ? src[(size_t)(blk + i) * LD + t] : 0.0;
    const int jstart = p.triangular ? (int)blk : 0;
    for (int j = jstart; j < D; j++) {
      const double rjt = active ? R[(size_t)j * LD + t] : 0.0; // issued before the barrier: latency overlaps the owner's work
      if (t == j) {
        // Householder reflector for [R_jj ; b]
        double sig = 0.0;
#pragma unroll
        for (int i = 0; i < B; i++) sig = fma(b[i], b[i], sig);
        double tau = 0.0;
        if (sig > 0.0) {
          const double alpha = rjt;
          double beta = sqrt(alpha * alpha + sig);
          if (alpha >= 0.0) beta = -beta;
          const double scale = 1.0 / (alpha - beta);
          tau = (beta - alpha) / beta;
#pragma unroll
          for (int i = 0; i < B; i++) {
            vbuf[par][i] = b[i] * scale;
            b[i] = 0.0;
          }
          R[(size_t)j * LD + j] = beta;
        }
        vbuf[par][B] = tau;
      }
      __syncthreads();
      const double tau = vbuf[par][B];
      if (tau != 0.0 && t > j && active) {
        double w = rjt;
#pragma unroll
        for (int i = 0; i < B; i++) w = fma(vbuf[par][i], b[i], w);
        w *= tau;
        R[(size_t)j * LD + t] = rjt - w;
#pragma unroll
        for (int i = 0; i < B; i++) b[i] = fma(-w, vbuf[par][i], b[i]);
      }
      par ^= 1;
    }
  }
}

} // namespace ovg
