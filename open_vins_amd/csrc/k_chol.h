// k_chol.h — blocked Cholesky with carried columns in ONE launch:  [A | C]  ->  [U | U^-T C],  A = U^T U  (A: D x D, D <= 256).
//
// Both factorisations of the Gram-form EKF update (k_ekf.h: the prior block P_DD carrying P(D, :), then T = I + G / s^2 carrying
// [B | h]) used to be 13 launches of k_ekf_chol_step each — a chain of 16-row steps in which every launch boundary, and every
// wavefront refactoring the diagonal block for itself, sat on the critical path (26 launches = 0.29 ms of a 1.15 ms update).
//
//   workgroup 0 ("factor")   8 wavefronts hold the upper triangle of A as 16 x 16 tiles in REGISTERS (accumulator layout of
//                            v_mfma_f64_16x16x4_f64; tile columns j and 15 - j to one wavefront: 17 tiles each).  Step k:
//                              the owner of tile (k, k) factors it (k_feat.h: diag_tile_factor, U_kk^-1 falls out of the same
//                              instruction stream)                                                    -> LDS, barrier
//                              every wavefront: W_kj = U_kk^-T S_kj on the matrix cores, W_kj -> LDS row panel, Y and (transposed) L
//                                                                                                     -> barrier, publish step k
//                              every wavefront: S_ij -= W_ki^T W_kj, operands from the LDS row panel
//                            The hand-offs of the chain are LDS + s_barrier; nothing leaves the compute unit on the critical path.
//   workgroups 1..           one wavefront per 16 carried columns (all 16-row tiles of those columns in registers).  They follow
//                            the factor workgroup through a per-step flag in memory (release / acquire at agent scope, bounded
//                            spin), read U_kk^-1 and the row panel from L2 and apply the same two products.  They trail the
//                            chain by one step.
//
// Padding: rows / columns D .. 16 ceil(D / 16) - 1 of A behave as an identity block and the carried columns are tiled from
// column D on, so no tile mixes matrix and carried columns; nothing outside [D x LA] is read or written.
#pragma once
#include "k_feat.h"

namespace ovg {
namespace chol {

typedef double d4 __attribute__((ext_vector_type(4)));

struct CholParams {
  int D, LA;              // A: D x D in the first D columns of the [D x LA] row-major work matrix, LA - D carried columns
  const double *A;        // [D x LA] input (not modified)
  double *Y;              // [D x LA] output [U | U^-T C] (U upper triangular, zeros below its diagonal are written too)
  double *Lt;             // optional [D x D]: U^T (lower triangular; the caller keeps the upper part zero)
  int32_t *flags;         // [0] = 1 when a pivot is not positive (or below pivot_tol * diag0)
  const double *diag0;    // optional [D]
  double pivot_tol;
  const int32_t *pred;    // optional: nothing happens when *pred == 0
  int32_t *prog;          // [16] step k published (zeroed before the launch)
  double *uinv;           // [16][256] U_kk^-1 of every step, row-major
  int32_t *err;           // sticky: a follower ran into its wait bound
};

constexpr int CH_TMAX = 16;  // tile rows: D <= 256
constexpr int CH_NW = 8;     // wavefronts per workgroup
constexpr int CH_SLOTS = 17; // tiles per factor wavefront

__device__ __forceinline__ double ld_a(const CholParams &p, int r, int c) { // element (r, c) of the padded matrix part
  return (r < p.D && c < p.D) ? p.A[(size_t)r * p.LA + c] : (r == c ? 1.0 : 0.0);
}

__global__ void __launch_bounds__(64 * CH_NW, 2) k_chol_pipe(CholParams p) {
  __shared__ __attribute__((aligned(16))) double panel[CH_TMAX][256];
  __shared__ __attribute__((aligned(16))) double st[2][256];
  if (p.pred && *p.pred == 0) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, cl = lane & 15;
  const int D = p.D, LA = p.LA, TM = (D + 15) >> 4;

  if (blockIdx.x == 0) {
    // ------------------------------------------------------------------ factor workgroup
    // slot s of wavefront w: s <= w -> tile (s, w); else tile (s - w - 1, 15 - w)
    d4 acc[CH_SLOTS];
#pragma unroll
    for (int s = 0; s < CH_SLOTS; s++) {
      const int i = s <= wv ? s : s - wv - 1, j = s <= wv ? wv : 15 - wv;
      d4 v = {0.0, 0.0, 0.0, 0.0};
      if (j < TM && i <= j) {
#pragma unroll
        for (int q = 0; q < 4; q++) v[q] = ld_a(p, 16 * i + g + 4 * q, 16 * j + cl);
      }
      acc[s] = v;
    }
    for (int k = 0; k < TM; k++) {
      // (1) diagonal tile
      const int ow = k < 8 ? k : 15 - k; // tile (k, k): column k < 8 -> wavefront k, slot k; column k >= 8 -> wavefront 15 - k, slot 16
      if (wv == ow) {
        const int slot = k < 8 ? k : 16;
        d4 av = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < CH_SLOTS; s++)
          if (s == slot) av = acc[s];
#pragma unroll
        for (int q = 0; q < 4; q++) st[0][(g + 4 * q) * 16 + cl] = av[q];
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        // (the pivots appear inside the chain: the routine reports the smallest one, relative to the matrix's own diagonal)
        const double worst = feat::diag_tile_factor_u(st[0], st[1], lane, p.diag0 ? p.diag0 + 16 * k : nullptr, D - 16 * k);
        if (!(worst > (p.diag0 ? p.pivot_tol : 0.0)) && lane == 0) p.flags[0] = 1;
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        // U_kk (st[0], row-major, zeros below the diagonal) -> Y and L^T; U_kk^-1 (st[1]) -> memory for the followers
#pragma unroll
        for (int q = 0; q < 4; q++) {
          const int r = 16 * k + g + 4 * q, c = 16 * k + cl;
          const double u = st[0][(g + 4 * q) * 16 + cl];
          if (r < D && c < D) {
            p.Y[(size_t)r * LA + c] = u;
            if (p.Lt) p.Lt[(size_t)c * D + r] = u;
          }
          p.uinv[(size_t)k * 256 + (g + 4 * q) * 16 + cl] = st[1][(g + 4 * q) * 16 + cl];
        }
      }
      __syncthreads();
      // (2) row panel W_kj = U_kk^-T S_kj
      {
        double ua[4];
#pragma unroll
        for (int u = 0; u < 4; u++) ua[u] = st[1][(4 * u + g) * 16 + cl];
#pragma unroll
        for (int s = 0; s < CH_SLOTS; s++) {
          const int i = s <= wv ? s : s - wv - 1, j = s <= wv ? wv : 15 - wv;
          if (i == k && j > k && j < TM) {
            d4 w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 4; u++) FEAT_MFMA(ua[u], acc[s][u], w);
            acc[s] = w;
#pragma unroll
            for (int q = 0; q < 4; q++) {
              panel[j][(g + 4 * q) * 16 + cl] = w[q];
              const int r = 16 * k + g + 4 * q, c = 16 * j + cl;
              if (r < D && c < D) {
                p.Y[(size_t)r * LA + c] = w[q];
                if (p.Lt) p.Lt[(size_t)c * D + r] = w[q];
              }
            }
          }
        }
        // rows of U below the diagonal of this tile row: zeros in Y (the factor is dense upper triangular there)
        for (int e = tid; e < 16 * 16 * k; e += 64 * CH_NW) {
          const int r = 16 * k + (e & 15), c = e >> 4;
          if (r < D) p.Y[(size_t)r * LA + c] = 0.0;
        }
      }
      __syncthreads(); // panel complete; all stores of the step issued and waited for (vmcnt(0) precedes the barrier)
      if (tid == 0) { // publish step k to the followers
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(p.prog + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      }
      // (3) trailing update S_ij -= W_ki^T W_kj
#pragma unroll
      for (int s = 0; s < CH_SLOTS; s++) {
        const int i = s <= wv ? s : s - wv - 1, j = s <= wv ? wv : 15 - wv;
        if (i > k && i <= j && j < TM) {
          double a[4], b[4];
#pragma unroll
          for (int u = 0; u < 4; u++) a[u] = -panel[i][(4 * u + g) * 16 + cl], b[u] = panel[j][(4 * u + g) * 16 + cl];
#pragma unroll
          for (int u = 0; u < 4; u++) FEAT_MFMA(a[u], b[u], acc[s]);
        }
      }
      // no barrier: the next step's diagonal tile goes through st[], its panel writes come behind its first barrier
    }
    return;
  }

  // -------------------------------------------------------------------- followers: one wavefront per 16 carried columns
  const int jc = (blockIdx.x - 1) * CH_NW + wv; // carried tile
  const int c0 = D + 16 * jc;
  if (c0 >= LA) return;
  const int col = c0 + cl;
  const bool colok = col < LA;
  d4 acc[CH_TMAX];
#pragma unroll
  for (int i = 0; i < CH_TMAX; i++) {
    d4 v = {0.0, 0.0, 0.0, 0.0};
    if (i < TM) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const int r = 16 * i + g + 4 * q;
        v[q] = (r < D && colok) ? p.A[(size_t)r * LA + col] : 0.0;
      }
    }
    acc[i] = v;
  }
  for (int k = 0; k < TM; k++) {
    // wait for step k of the factor workgroup
    int spins = 0;
    while (__hip_atomic_load(p.prog + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > (1 << 22)) {
        if (lane == 0) p.err[0] = 1;
        return;
      }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    double ua[4];
#pragma unroll
    for (int u = 0; u < 4; u++) ua[u] = p.uinv[(size_t)k * 256 + (4 * u + g) * 16 + cl];
    d4 w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int i = 0; i < CH_TMAX; i++) {
      if (i == k) {
#pragma unroll
        for (int u = 0; u < 4; u++) FEAT_MFMA(ua[u], acc[i][u], w);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int r = 16 * k + g + 4 * q;
      if (r < D && colok) p.Y[(size_t)r * LA + col] = w[q];
    }
    // the row panel's tiles (k, i), i > k, as A operands: element (4u + g, cl) of tile i = Y[16 k + 4u + g][16 i + cl]; 8 tiles in flight
#pragma unroll
    for (int h = 0; h < CH_TMAX; h += 8) {
      if (h + 7 <= k || h >= TM) continue;
      double wa[8][4];
#pragma unroll
      for (int ii = 0; ii < 8; ii++) {
        const int i = h + ii;
#pragma unroll
        for (int u = 0; u < 4; u++) {
          const int r = 16 * k + 4 * u + g, c = 16 * i + cl;
          wa[ii][u] = (i > k && i < TM && r < D && c < D) ? p.Y[(size_t)r * LA + c] : 0.0;
        }
      }
#pragma unroll
      for (int ii = 0; ii < 8; ii++) {
        const int i = h + ii;
        if (i > k && i < TM) {
#pragma unroll
          for (int u = 0; u < 4; u++) FEAT_MFMA(-wa[ii][u], w[u], acc[i]);
        }
      }
    }
  }
}

} // namespace chol
} // namespace ovg
