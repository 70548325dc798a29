// k_chol.h — blocked Cholesky with carried columns in ONE launch:  [A | C]  ->  [U | U^-T C],  A = U^T U  (A: D x D, D <= 256).
//
// Both factorisations of the Gram-form EKF update (k_ekf.h: the prior block P_DD carrying P(D, :), then T = I + G / s^2 carrying
// [B | h]) used to be 13 launches of k_ekf_chol_step each — a chain of 16-row steps in which every launch boundary, and every
// wavefront refactoring the diagonal block for itself, sat on the critical path (26 launches = 0.29 ms of a 1.15 ms update).
//
//   k_chol_factor            ONE workgroup: 15 wavefronts hold the upper triangle of A as 16 x 16 tiles in REGISTERS (accumulator
//                            layout of v_mfma_f64_16x16x4_f64; tiles dealt round-robin, 10 per wavefront), a 16th runs the
//                            chain of diagonal tiles and holds nothing else (the factorisation of a tile needs ~90 registers: in
//                            a wavefront that also holds tiles they end up in scratch, and scratch latency on the chain).  Step k:
//                              the owner of tile (k, k) hands it over through LDS, the chain wavefront factors it (k_feat.h:
//                              diag_tile_factor, U_kk^-1 falls out of the same instruction stream)     -> LDS, barrier
//                              every wavefront: W_kj = U_kk^-T S_kj on the matrix cores, W_kj -> LDS row panel, Y and (transposed) L
//                                                                                                     -> barrier, publish step k
//                              every wavefront: S_ij -= W_ki^T W_kj, operands from the LDS row panel
//                            The hand-offs of the chain are LDS + s_barrier; nothing leaves the compute unit on the critical path.
//   k_chol_follow            (second stream) one wavefront per 16 carried columns (all 16-row tiles of those columns in registers).  They follow
//                            the factor workgroup through a per-step counter in memory (bounded spin): the factor workgroup's
//                            wavefronts write U_kk^-1 and the row panel through to memory (sc1 stores) and count
//                            themselves in one step later, the followers read past the caches.  No fence, no vmcnt wait and
//                            no store sits on the factor workgroup's chain.
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
  double *Y;              // [D x LA] output [U | U^-T C] (U upper triangular; entries left of the diagonal TILES are not written)
  double *Lt;             // optional [D x D]: U^T (lower triangular; the caller keeps the upper part zero)
  int32_t *flags;         // [0] = 1 when a pivot is not positive (or below pivot_tol * diag0)
  const double *diag0;    // optional [D]
  double pivot_tol;
  const int32_t *pred;    // optional: nothing happens when *pred == 0
  int32_t *prog;          // [16] step k published (zeroed before the launch)
  double *uinv;           // [16][256] U_kk^-1 of every step, row-major
  int32_t *err;           // sticky: a follower ran into its wait bound
  int spin_limit;         // the followers' wait bound per step (polls of ~100 cycles; 1 << 22 ~ 0.2 s)
  long long *dbg;         // optional cycle counters (developer aid)
  // Where [A | C] comes from.  The two factorisations of the Gram-form update each had a small kernel in front that only assembled
  // the work matrix (k_tf_gather, k_tf_abh: a launch and a stream hand-over on the critical path each); the factorisation reads
  // every input element exactly once, so it can as well read it from where it lives:
  //   CH_SRC_MATRIX   A as stored
  //   CH_SRC_PRIOR    [P_DD | P(D, :) | 0] gathered from the covariance through col_cov (k_tf_gather)
  //   CH_SRC_WHITENED [I + G / s^2 | B | g / s^2]: G the Gram matrix of the whitened stack, B the carried columns of Y1 (k_tf_abh)
  int src = 0;
  int N = 0;                          // carried covariance columns (LA = D + N + 1)
  const int32_t *col_cov = nullptr;   // CH_SRC_PRIOR
  const double *P = nullptr;          // CH_SRC_PRIOR: [N x N]
  const double *G = nullptr;          // CH_SRC_WHITENED: [LG x LG]
  int LG = 0;
  double inv_sigma2 = 0.0;
  const double *Y1 = nullptr;         // CH_SRC_WHITENED: [D x LA], the first factorisation's result
  const int32_t *pred_not = nullptr;  // optional: nothing happens when *pred_not != 0 (the not-SPD / time-out flag of an earlier factorisation)
};
enum { CH_SRC_MATRIX = 0, CH_SRC_PRIOR = 1, CH_SRC_WHITENED = 2 };

constexpr int CH_TMAX = 16;  // tile rows: D <= 256
constexpr int CH_NW = 8;     // wavefronts per workgroup

__device__ __forceinline__ double ld_a(const CholParams &p, int r, int c) { // element (r, c) of the padded matrix part
  if (!(r < p.D && c < p.D)) return r == c ? 1.0 : 0.0;
  if (p.src == CH_SRC_PRIOR) return p.P[(size_t)p.col_cov[r] * p.N + p.col_cov[c]];
  if (p.src == CH_SRC_WHITENED) return p.G[(size_t)r * p.LG + c] * p.inv_sigma2 + (r == c ? 1.0 : 0.0);
  return p.A[(size_t)r * p.LA + c];
}
__device__ __forceinline__ double ld_c(const CholParams &p, int r, int col) { // carried column col (D <= col < LA) of row r < D
  const int cc = col - p.D;
  if (p.src == CH_SRC_PRIOR) return cc < p.N ? p.P[(size_t)p.col_cov[r] * p.N + cc] : 0.0;
  if (p.src == CH_SRC_WHITENED) return cc < p.N ? p.Y1[(size_t)r * p.LA + col] : p.G[(size_t)r * p.LG + p.D] * p.inv_sigma2;
  return p.A[(size_t)r * p.LA + col];
}
__device__ __forceinline__ bool chol_skipped(const CholParams &p) { return (p.pred && *p.pred == 0) || (p.pred_not && *p.pred_not != 0); }

constexpr int CH_FW = 15;  // tile wavefronts of the factor workgroup (wavefront 15 runs the diagonal chain and holds no tiles)
constexpr int CH_FT = 10;  // tiles per wavefront: 16 * 17 / 2 = 136 <= 15 * 10

__global__ void __launch_bounds__(64 * (CH_FW + 1), 4) k_chol_factor(CholParams p) {
  __shared__ __attribute__((aligned(16))) double panel[CH_TMAX][256];
  __shared__ __attribute__((aligned(16))) double st[2][256];
  __shared__ int diag_ready; // number of diagonal tiles handed to the chain wavefront so far (look-ahead hand-over, see below)
  __shared__ double d0s[16 * CH_TMAX]; // CH_SRC_PRIOR: the diagonal of the matrix before the factorisation (pivot test)
  if (chol_skipped(p)) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, cl = lane & 15;
  const int D = p.D, LA = p.LA, TM = (D + 15) >> 4, NTT = TM * (TM + 1) / 2;
  if (tid == 0) diag_ready = 0;
  if (p.src == CH_SRC_PRIOR && tid < 16 * CH_TMAX) d0s[tid] = tid < D ? p.P[(size_t)p.col_cov[tid] * p.N + p.col_cov[tid]] : 1.0;
  const double *diag0 = p.src == CH_SRC_PRIOR ? d0s : p.diag0;
  __syncthreads();
  // Barriers of the chain order LDS traffic only; memory traffic is fire-and-forget.  Data for the followers leaves with
  // write-through stores (sc1), and a wavefront adds itself to prog[k - 1] one step LATER, when those stores have long completed
  // (a release fence or a vmcnt(0) wait right behind the stores would sit on the chain: 2-6 us per step).
  auto lds_barrier = [] { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); };
  auto st_dev = [](double *ptr, double v) { __hip_atomic_store(ptr, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
  auto arrive = [&](int k) { // this wavefront's stores of step k are complete
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) (void)__hip_atomic_fetch_add(p.prog + k, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  };

  if (wv == CH_FW) {
    // ------------------------------------------------------------------ the diagonal chain: one wavefront, no tiles of its own
    const long long t_begin = clock64();
    long long t_diag = 0;
    for (int k = 0; k < TM; k++) {
      // Look-ahead: the owner of tile (k, k) brings it up to date FIRST in the trailing update of step k - 1, parks it in st[0] and
      // raises diag_ready; this wavefront factors it while the others are still in that trailing update.  (st[0] / st[1] are free
      // then: the panel solve of step k - 1 read st[1] before barrier B2, which this wavefront has passed.)
      while (__hip_atomic_load(&diag_ready, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) <= k) __builtin_amdgcn_s_sleep(1);
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      const long long t_d0 = clock64();
      d4 sv, ev;
#pragma unroll
      for (int q = 0; q < 4; q++) sv[q] = st[0][(g + 4 * q) * 16 + cl];
      __builtin_amdgcn_wave_barrier(); // st[0] becomes the factorisation's scratch
      const bool bad = feat::diag_tile_factor_blk(sv, ev, st[0], lane, diag0 ? diag0 + 16 * k : nullptr, p.pivot_tol, D - 16 * k);
      if (bad && lane == 0) __hip_atomic_store(p.flags, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); // written through: a consumer may start before this kernel ends
#pragma unroll
      for (int q = 0; q < 4; q++) st[1][cl * 16 + g + 4 * q] = ev[q]; // U^-T in accumulator layout -> U^-1 row-major
      t_diag += clock64() - t_d0;
      lds_barrier(); // B0 (kept so that every wavefront counts the same barriers)
      lds_barrier(); // B1: U_kk^-1 is in st[1]
      if (k > 0) arrive(k - 1);
#pragma unroll
      for (int q = 0; q < 4; q++) { // U_kk -> Y and L; U_kk^-1 -> memory for the followers
        const int r = 16 * k + g + 4 * q, c = 16 * k + cl;
        if (r < D && c < D) {
          p.Y[(size_t)r * LA + c] = sv[q];
          if (p.Lt) p.Lt[(size_t)c * D + r] = sv[q];
        }
        st_dev(p.uinv + (size_t)k * 256 + cl * 16 + g + 4 * q, ev[q]);
      }
      // (Y left of the diagonal tile is never read: the consumers of U take k >= row only)
      lds_barrier(); // B2
    }
    arrive(TM - 1);
    if (p.dbg && lane == 0) p.dbg[300] += clock64() - t_begin, p.dbg[301] += t_diag, p.dbg[302] += 1;
    return;
  }

  // -------------------------------------------------------------------- tile wavefronts: linear tile index t = s CH_FW + wv over the upper triangle, column by column
  int tij[CH_FT]; // (j << 8) | i, or -1
  d4 acc[CH_FT];
#pragma unroll
  for (int s = 0; s < CH_FT; s++) {
    const int t = s * CH_FW + wv;
    int i = -1, j = 0;
    if (t < NTT) {
      while ((j + 1) * (j + 2) / 2 <= t) j++;
      i = t - j * (j + 1) / 2;
    }
    tij[s] = i < 0 ? -1 : ((j << 8) | i);
    d4 v = {0.0, 0.0, 0.0, 0.0};
    if (i >= 0) {
#pragma unroll
      for (int q = 0; q < 4; q++) v[q] = ld_a(p, 16 * i + g + 4 * q, 16 * j + cl);
    }
    acc[s] = v;
  }
#define CTI(s) (tij[s] & 255)
#define CTJ(s) (tij[s] >> 8)
  // hands diagonal tile kd (already up to date in this wavefront's registers) to the chain wavefront
  auto hand_over = [&](int kd) {
    const int slot = (kd * (kd + 1) / 2 + kd) / CH_FW;
    d4 av = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int s = 0; s < CH_FT; s++)
      if (s == slot) av = acc[s];
#pragma unroll
    for (int q = 0; q < 4; q++) st[0][(g + 4 * q) * 16 + cl] = av[q];
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (lane == 0) __hip_atomic_store(&diag_ready, kd + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
  };
  if (wv == 0) hand_over(0); // tile (0, 0): t = 0 belongs to wavefront 0
  for (int k = 0; k < TM; k++) {
    // (1) the diagonal tile is already with the chain wavefront (hand_over in the previous trailing update)
    lds_barrier(); // B0
    lds_barrier(); // B1
    // (2) row panel W_kj = U_kk^-T S_kj -> LDS (for the trailing update) and memory (final: the tile is not needed here again).
    //     One copy of the code, the register slot is selected at run time; the stores are fire-and-forget.
    if (k > 0) arrive(k - 1); // the previous row's stores were issued a step ago
    {
      double ua[4];
#pragma unroll
      for (int u = 0; u < 4; u++) ua[u] = st[1][(4 * u + g) * 16 + cl];
      for (int j = k + 1; j < TM; j++) {
        const int t = j * (j + 1) / 2 + k;
        if (t % CH_FW != wv) continue;
        const int slot = t / CH_FW;
        d4 sv = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int s = 0; s < CH_FT; s++)
          if (s == slot) sv = acc[s];
        d4 w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int u = 0; u < 4; u++) FEAT_MFMA(ua[u], sv[u], w);
        double *pt = panel[j];
#pragma unroll
        for (int q = 0; q < 4; q++) {
          pt[(g + 4 * q) * 16 + cl] = w[q];
          const int r = 16 * k + g + 4 * q, c = 16 * j + cl;
          if (r < D && c < D) {
            st_dev(p.Y + (size_t)r * LA + c, w[q]);
            if (p.Lt) p.Lt[(size_t)c * D + r] = w[q];
          }
        }
      }
    }
    lds_barrier(); // B2
    // (3) trailing update S_ij -= W_ki^T W_kj; the next diagonal tile first, and straight to the chain wavefront
    const int tnext = (k + 1) * (k + 2) / 2 + k + 1;
    const bool own_next = k + 1 < TM && tnext % CH_FW == wv;
    const int slot_next = own_next ? tnext / CH_FW : -1;
    if (own_next) {
      const double *pi = panel[k + 1];
      double a[4], b[4];
#pragma unroll
      for (int u = 0; u < 4; u++) b[u] = pi[(4 * u + g) * 16 + cl], a[u] = -b[u];
      d4 av = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
      for (int s = 0; s < CH_FT; s++)
        if (s == slot_next) av = acc[s];
#pragma unroll
      for (int u = 0; u < 4; u++) FEAT_MFMA(a[u], b[u], av);
#pragma unroll
      for (int s = 0; s < CH_FT; s++)
        if (s == slot_next) acc[s] = av;
      hand_over(k + 1);
    }
#pragma unroll
    for (int s = 0; s < CH_FT; s++) {
      if (tij[s] >= 0 && CTI(s) > k && s != slot_next) {
        const double *pi = panel[CTI(s)], *pj = panel[CTJ(s)];
        double a[4], b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) a[u] = -pi[(4 * u + g) * 16 + cl], b[u] = pj[(4 * u + g) * 16 + cl];
#pragma unroll
        for (int u = 0; u < 4; u++) FEAT_MFMA(a[u], b[u], acc[s]);
      }
    }
  }
  arrive(TM - 1);
#undef CTI
#undef CTJ
}

// followers: one wavefront per 16 carried columns (launched on a second stream next to k_chol_factor)
__global__ void __launch_bounds__(64 * CH_NW, 2) k_chol_follow(CholParams p) {
  if (chol_skipped(p)) return;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, cl = lane & 15;
  const int D = p.D, LA = p.LA, TM = (D + 15) >> 4;
  const int jc = blockIdx.x * CH_NW + wv; // carried tile
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
        v[q] = (r < D && colok) ? ld_c(p, r, col) : 0.0;
      }
    }
    acc[i] = v;
  }
  const long long f_begin = clock64();
  long long f_wait = 0;
  for (int k = 0; k < TM; k++) {
    // wait for step k of the factor workgroup
    const long long f_w0 = clock64();
    int spins = 0;
    while (__hip_atomic_load(p.prog + k, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < CH_FW + 1) {
      __builtin_amdgcn_s_sleep(2);
      if (++spins > p.spin_limit) {
        // The factor workgroup never got scheduled next to us (a shared GPU, a full chip).  The carried columns stay unwritten, so
        // NOTHING behind this factorisation may run: err is raised for the host, and the update is switched off through the very
        // words the following kernels are predicated on — the not-SPD flag of a first factorisation (k_tf_abh derives `go` from
        // it), the `go` word itself for a second one.  The resident state is then untouched and the host repeats the update with
        // the step-wise kernels (finish_update / update_with_fallbacks).
        if (lane == 0) {
          p.err[0] = 1;
          p.flags[0] = 1;
          if (p.pred) *const_cast<int32_t *>(p.pred) = 0;
        }
        return;
      }
    }
    f_wait += clock64() - f_w0;
    // the factor workgroup's data was written through (sc1 stores): sc1 loads read it past this CU's L1
    auto ld_sys = [](const double *ptr) { return __hip_atomic_load(ptr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); };
    double ua[4];
#pragma unroll
    for (int u = 0; u < 4; u++) ua[u] = ld_sys(p.uinv + (size_t)k * 256 + (4 * u + g) * 16 + cl);
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
          wa[ii][u] = (i > k && i < TM && r < D && c < D) ? ld_sys(p.Y + (size_t)r * LA + c) : 0.0;
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
  if (p.dbg && blockIdx.x == 0 && tid == 0) p.dbg[303] += clock64() - f_begin, p.dbg[304] += f_wait;
}

} // namespace chol
} // namespace ovg
