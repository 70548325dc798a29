// k_unwhiten.h — X = R L^-1 for mode A through the Gram route (round 6: right-looking, four wavefronts per 16 rows).
//
// The per-feature stage leaves the stack whitened by the prior, [H L | r] with P_DD = L L^T; R is the pivoted Cholesky factor of its Gram
// matrix (k_pchol.h), so X = R L^-1 has X^T X = H^T H: (X, R's last column) is a compressed system of the reference's form for the stock
// StateHelper::EKFUpdate (StateHelper.cpp:116-197; UpdaterHelper.cpp:456-487 is what it replaces).  In place on the first D columns of R [D x LD].
//
// k_unwhiten (k_ekf.h, rounds 3-5) gives 16 rows to ONE wavefront: per tile column a product over every tile right of it, then sixteen
// substitution steps inside the tile, each a shuffle round trip — 13 tile columns in sequence, 95 us at D = 208.  Here:
//   * the 16 rows sit in the registers of FOUR wavefronts as accumulator tiles S_j (tile column j with wavefront j & 3); tile column J from the
//     right: its owner multiplies by the INVERSE of the diagonal tile — X_J = S_J (U_JJ^-1)^T, four matrix instructions; U_JJ^-1 is what the
//     prior's factorisation already wrote for its own followers (chol::CholParams::uinv) —, publishes X_J in LDS, and after one workgroup
//     barrier every wavefront takes X_J L(J, j) off its own tiles j < J (right-looking: no product grows with the distance from the right edge);
//   * every operand that comes from memory (L(J, j) = U1(j, J)^T read in place, U_JJ^-1) is independent of X and requested one step ahead.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>
#include <type_traits>

namespace ovg {

template <int NTM> // tile columns at most (D <= 16 NTM <= 256)
__global__ void __launch_bounds__(256) k_unwhiten_blk(int D, int LD, double *__restrict__ R, const double *__restrict__ Y1, int LA, const double *__restrict__ uinv,
                                                       const int32_t *pred, const int32_t *pred_not) {
  typedef double d4 __attribute__((ext_vector_type(4)));
  constexpr int SLOTS = (NTM + 3) / 4, TS = 17;
  __shared__ double T[4][16 * TS];  // a wavefront's own tile on its way from the accumulator layout to an A operand
  __shared__ double XT[2][16 * TS]; // X_J for everybody
  if ((pred && *pred == 0) || (pred_not && *pred_not != 0)) return; // (pred_not: the not-SPD flag of the prior block's factorisation)
  const int tid = threadIdx.x, lane = tid & 63, g = lane >> 4, cl = lane & 15;
  const int w = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int r0 = blockIdx.x * 16, NT = (D + 15) >> 4;
  auto wsync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  // S_j, j = w + 4 s
  d4 acc[SLOTS];
  const double *lrow[SLOTS]; // row 16 j + cl of U1 = column 16 j + cl of L, from its element g on
#pragma unroll
  for (int s = 0; s < SLOTS; s++) {
    const int j = w + 4 * s, col = 16 * j + cl;
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int row = r0 + 4 * q + g;
      const bool ok = j < NT && row < D && col < D;
      const double x = R[(size_t)(ok ? row : 0) * LD + (ok ? col : 0)];
      acc[s][q] = ok ? x : 0.0;
    }
    lrow[s] = Y1 + (size_t)min(col, D - 1) * LA + g;
  }
  // operands of step J that come from memory: bl[s][u] = L(16 J + 4 u + g, 16 j + cl) for the owned j < J; bu[u] = U_JJ^-1 (cl, 4 u + g) for the owner
  double bl[2][SLOTS][4], bu[2][4];
  auto request = [&](int J, int buf) {
    if (J < 0) return;
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
#pragma unroll
      for (int u = 0; u < 4; u++) {
        const int k = 16 * J + 4 * u; // (+ g inside lrow)
        bl[buf][s][u] = lrow[s][min(k, LA - 1 - 3)];
      }
    }
    if (w == (J & 3)) {
#pragma unroll
      for (int u = 0; u < 4; u++) bu[buf][u] = uinv[(size_t)J * 256 + cl * 16 + 4 * u + g];
    }
  };
  auto step = [&](int J, int buf, auto slot_c) {
    constexpr int SJ = decltype(slot_c)::value; // the slot of tile column J in its owner: J >> 2
    if (w == (J & 3)) { // X_J = S_J (U_JJ^-1)^T
      double *t = T[w];
#pragma unroll
      for (int q = 0; q < 4; q++) t[(4 * q + g) * TS + cl] = acc[SJ][q];
      wsync();
      d4 x = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int u = 0; u < 4; u++) x = __builtin_amdgcn_mfma_f64_16x16x4f64(t[cl * TS + 4 * u + g], bu[buf][u], x, 0, 0, 0);
      const int col = 16 * J + cl;
      double *xt = XT[J & 1];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int row = r0 + 4 * q + g;
        xt[(4 * q + g) * TS + cl] = x[q];
        if (row < D && col < D) R[(size_t)row * LD + col] = x[q];
      }
    }
    __syncthreads();
    if (J == 0) return;
    const double *xt = XT[J & 1];
    double a[4];
#pragma unroll
    for (int u = 0; u < 4; u++) a[u] = -xt[cl * TS + 4 * u + g];
#pragma unroll
    for (int s = 0; s < SLOTS; s++) {
      if (w + 4 * s < J) { // (wave-uniform)
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const bool kok = 16 * J + 4 * u + g < D; // (the last tile column's padding)
          acc[s] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], kok ? bl[buf][s][u] : 0.0, acc[s], 0, 0, 0);
        }
      }
    }
  };
  // tile columns from the right; fully unrolled over the template's tile columns (the owner's slot and the operand buffer are constants of each
  // copy): column J reads buffer (NTM - 1 - J) & 1 and requests column J - 1 into the other one
  if (((NTM - NT) & 1) != 0) request(NT - 1, 1);
  else request(NT - 1, 0);
#define OVG_UW_STEP(JJ)                                                                        \
  if constexpr ((JJ) < NTM) {                                                                  \
    if ((JJ) < NT) {                                                                           \
      request((JJ)-1, ((NTM - 1 - (JJ)) & 1) ^ 1);                                             \
      step((JJ), (NTM - 1 - (JJ)) & 1, std::integral_constant<int, ((JJ) >> 2)>{});            \
    }                                                                                          \
  }
  OVG_UW_STEP(15) OVG_UW_STEP(14) OVG_UW_STEP(13) OVG_UW_STEP(12) OVG_UW_STEP(11) OVG_UW_STEP(10) OVG_UW_STEP(9) OVG_UW_STEP(8)
  OVG_UW_STEP(7) OVG_UW_STEP(6) OVG_UW_STEP(5) OVG_UW_STEP(4) OVG_UW_STEP(3) OVG_UW_STEP(2) OVG_UW_STEP(1) OVG_UW_STEP(0)
#undef OVG_UW_STEP
}

} // namespace ovg
