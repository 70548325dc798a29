// k_ekf.h — StateHelper::EKFUpdate on the device (StateHelper.cpp:116-197) and the box-plus of
// the resident pose tables (JPLQuat.h:114-125, PoseJPL.h:74-91, Vec.h:55-58).
//
// With the compressed system [R | c] (D x D upper triangular, R_noise = sigma^2 I):
//   Mt = R P(cols,:)                       (D x N)   = (P H^T)^T            StateHelper.cpp:137-146
//   S  = R P(cols,cols) R^T + sigma^2 I    (D x D)                          :151-156
//   S  = U^T U ;  Y = U^-T Mt ; y = U^-T c           (replaces Sinv, K)     :160-162   (k_ekf_chol_step, one launch per 16 rows)
//   P' = P - Y^T Y                                   (= P - K M^T)          :166-167
//   dx = Y^T y                                       (= K res)              :185
// The reference forms S^-1 explicitly; Y^T Y = M S^-1 M^T is the same matrix and is symmetric by
// construction, which is what the upper-triangle-then-mirror of :166-167 achieves.
//
// The three dense products run on the matrix cores with v_mfma_f64_16x16x4_f64: one wavefront
// per 16x16 output tile, operands streamed from L2 (all matrices here are < 1.5 MB).
#pragma once
#include "device_math.h"
#include "ovgpu_types.h"

namespace ovg {

typedef double double4_t __attribute__((ext_vector_type(4)));

// One 16x16 tile:  acc += A(16 x K) * B(K x 16) with A(i,k) / B(k,j) supplied by functors.
// f64 MFMA operand layout (cdna_hip_programming.md §3): lane l holds A[l & 15][l >> 4] and
// B[l >> 4][l & 15]; result register q of lane l is C[(l >> 4) + 4 q][l & 15].
template <class FA, class FB>
__device__ __forceinline__ double4_t mfma_tile(FA fa, FB fb, int K, int lane) {
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const int i = lane & 15, kk = lane >> 4;
  // 4 k-slices per trip: the 8 operand loads (L2 hits, ~0.5 us each when taken one by one) are issued together
  for (int k0 = 0; k0 < K; k0 += 16) {
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 + 4 * u + kk;
      a[u] = (k < K) ? fa(i, k) : 0.0;
      b[u] = (k < K) ? fb(k, i) : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u], b[u], acc, 0, 0, 0);
  }
  return acc;
}

struct EkfParams {
  int N, D, LD, LA; // D = rows of the system, LA = D + N + 1 columns of the augmented matrix
  int DC;           // Jacobian columns (= D for a compressed, upper-triangular system; the residual is column DC)
  int tri;          // 1: R is upper triangular (entries left of the diagonal are not read)
  const int32_t *pred; // optional: the whole update is skipped when *pred == 0 (delayed initialisation: feature rejected)
  const double *R;  // [D x LD] system rows (last column = residual)
  const int32_t *col_cov;
  double *P;        // [N x N] in/out
  double *Mt;       // [D x N]
  double *A;        // [D x LA] augmented [S | Mt | c] (work matrix of the factorisation)
  double *Y;        // [D x LA] result rows [U | U^-T Mt | U^-T c]
  double *dx;       // [N]
  int32_t *flags;   // [0] = 1 if S not SPD, [1] = 1 if a diagonal of P' is negative
  double sigma2;
  const double *diag0 = nullptr; // optional [D]: the matrix's own diagonal before the factorisation; a pivot below pivot_tol of it
                                 // counts as not SPD (numerically singular prior block of the Gram-form update)
  double pivot_tol = 1e-13;
  const int32_t *pred_not = nullptr; // optional: the whole update is skipped when *pred_not != 0 (flags[0] of an earlier factorisation)
};
__device__ __forceinline__ bool ekf_skipped(const EkfParams &p) { return (p.pred && *p.pred == 0) || (p.pred_not && *p.pred_not != 0); }

// Mt = R * P(cols, :)     grid: tiles(D/16) x tiles(N/16) wavefronts, 4 per block
__global__ void __launch_bounds__(256) k_ekf_mt(EkfParams p) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int tn = (p.N + 15) / 16, tm = (p.D + 15) / 16;
  if (tile >= tn * tm || (p.pred && *p.pred == 0)) return;
  const int r0 = (tile / tn) * 16, c0 = (tile % tn) * 16;
  auto fa = [&](int i, int k) { const int r = r0 + i; return (r < p.D && (k >= r || !p.tri)) ? p.R[(size_t)r * p.LD + k] : 0.0; };
  auto fb = [&](int k, int j) { const int c = c0 + j; return (c < p.N) ? p.P[(size_t)p.col_cov[k] * p.N + c] : 0.0; };
  const double4_t acc = mfma_tile(fa, fb, p.DC, lane);
  const int col = c0 + (lane & 15);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = r0 + (lane >> 4) + 4 * q;
    if (row < p.D && col < p.N) {
      p.Mt[(size_t)row * p.N + col] = acc[q];
      p.A[(size_t)row * p.LA + p.D + col] = acc[q];
    }
  }
}

// S = Mt(:, cols) R^T + sigma^2 I  -> A[:, 0:D] ; c -> A[:, D+N]
__global__ void __launch_bounds__(256) k_ekf_s(EkfParams p) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int tm = (p.D + 15) / 16;
  if (tile >= tm * tm || (p.pred && *p.pred == 0)) return;
  const int r0 = (tile / tm) * 16, c0 = (tile % tm) * 16;
  auto fa = [&](int i, int k) { const int r = r0 + i; return (r < p.D) ? p.Mt[(size_t)r * p.N + p.col_cov[k]] : 0.0; };
  auto fb = [&](int k, int j) { const int c = c0 + j; return (c < p.D && (k >= c || !p.tri)) ? p.R[(size_t)c * p.LD + k] : 0.0; };
  const double4_t acc = mfma_tile(fa, fb, p.DC, lane);
  const int col = c0 + (lane & 15);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = r0 + (lane >> 4) + 4 * q;
    if (row < p.D && col < p.D) p.A[(size_t)row * p.LA + col] = acc[q] + (row == col ? p.sigma2 : 0.0);
  }
  if (tile == 0)
    for (int r = lane; r < p.D; r += 64) p.A[(size_t)r * p.LA + p.D + p.N] = p.R[(size_t)r * p.LD + p.DC];
}

// ---------------------------------------------------------------------------------------------------
// Blocked right-looking Cholesky S = U^T U carried through the augmented columns, one launch per block of 16
// rows, one WAVEFRONT per 16x16 tile of the trailing matrix (a single-workgroup version needed 0.6 ms for
// D = 208: 13 block steps of a D x LA update by one CU).
//
// Step kb:  U_kk = chol(A[kb:kb+16, kb:kb+16]);  W = U_kk^-T A[kb:kb+16, kb+16:LA];  A[i][j] -= sum_l W[l][i] W[l][j].
// Every wave factors the 16x16 diagonal block itself (120 multiply-adds, in registers, pivots and multipliers
// moved with v_readlane) and solves for the two 16-column slices of W its tile needs, so a step has no
// intra-launch dependency: tiles only READ rows kb..kb+15 of A and only WRITE rows >= kb+16.  The finished rows
// [U_kk | W] go to a second matrix Y, which the covariance update reads (rows of A are never read again).
// The tile update itself is 4 v_mfma_f64_16x16x4_f64.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ double ekf_bcast(double v, int lane) { // lane is wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

__global__ void __launch_bounds__(256) k_ekf_chol_step(EkfParams p, int kb) {
  __shared__ double stage[4][2][16 * 16];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int D = p.D, LA = p.LA;
  const int tb = kb >> 4, TM = (D + 15) >> 4, TL = (LA + 15) >> 4;
  const int nb = min(16, D - kb);
  if (p.pred && *p.pred == 0) return;
  // wave -> job: first the TL - tb "writer" jobs (finished rows of column tile jt -> Y), then the trailing tiles (it, jt >= it)
  int job = blockIdx.x * 4 + wv;
  const int n_writer = TL - tb;
  int it = -1, jt = 0;
  if (job < n_writer) {
    jt = tb + job;
  } else {
    job -= n_writer;
    it = tb + 1;
    while (it < TM && job >= TL - it) job -= TL - it, it++;
    if (it >= TM) return;
    jt = it + job;
  }
  // ---- 1. U_kk: lane j < 16 holds column j of the diagonal block (upper part), identity beyond nb
  double u[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const int j = lane & 15;
    u[i] = (i < nb && j < nb && j >= i) ? p.A[(size_t)(kb + i) * LA + kb + j] : (i == j ? 1.0 : 0.0);
  }
  double dinv[16];
#pragma unroll
  for (int k = 0; k < 16; k++) {
    const double dk = ekf_bcast(u[k], k);
    if ((!(dk > 0.0) || (p.diag0 && k < nb && dk <= p.pivot_tol * p.diag0[kb + k])) && it < 0 && jt == tb && lane == 0) p.flags[0] = 1;
    const double d = sqrt(dk), inv = 1.0 / d;
    dinv[k] = inv;
    u[k] = ((lane & 15) == k) ? d : u[k] * inv; // row k of U (lanes j > k); lanes j < k hold zeros there
#pragma unroll
    for (int i = k + 1; i < 16; i++) {
      const double uki = ekf_bcast(u[k], i); // U[k][i]
      u[i] = fma(-uki, u[k], u[i]);          // only lanes j >= i matter
    }
  }
  // ---- 2. W = U_kk^-T B: lanes 0-15 solve the columns of tile `it` (or nothing), lanes 16-31 those of tile jt
  const int half = lane >> 4, cc = lane & 15;
  const int tcol = (half == 0 ? it : jt) * 16 + cc;
  const bool wvalid = half < 2 && (half == 1 || it >= 0) && tcol < LA;
  double w[16];
#pragma unroll
  for (int l = 0; l < 16; l++) w[l] = (wvalid && l < nb) ? p.A[(size_t)(kb + l) * LA + tcol] : 0.0;
#pragma unroll
  for (int l = 0; l < 16; l++) {
    double sacc = w[l];
#pragma unroll
    for (int q = 0; q < l; q++) sacc = fma(-ekf_bcast(u[q], l), w[q], sacc); // U[q][l] lives in lane l
    w[l] = sacc * dinv[l];
  }
  if (it < 0) {
    // writer: rows kb .. kb+nb-1 of Y, columns of tile jt (the diagonal tile stores U_kk itself: lane j holds column j)
    if (jt == tb) {
      if (lane < nb) {
#pragma unroll
        for (int l = 0; l < 16; l++)
          if (l < nb) p.Y[(size_t)(kb + l) * LA + kb + lane] = (lane >= l) ? u[l] : 0.0;
      }
      if (half == 1 && cc >= nb && tcol < LA) { // a partial last block: the rest of its tile are right-hand-side columns
#pragma unroll
        for (int l = 0; l < 16; l++)
          if (l < nb) p.Y[(size_t)(kb + l) * LA + tcol] = w[l];
      }
    } else if (half == 1 && tcol < LA) {
#pragma unroll
      for (int l = 0; l < 16; l++)
        if (l < nb) p.Y[(size_t)(kb + l) * LA + tcol] = w[l];
    }
    return;
  }
  // ---- 3. tile update A[i0.., j0..] -= W_i^T W_j on the matrix cores (operands staged through LDS)
  double *wi = stage[wv][0], *wj = stage[wv][1];
  if (half < 2) {
    double *dst = half == 0 ? wi : wj;
#pragma unroll
    for (int l = 0; l < 16; l++) dst[l * 16 + cc] = w[l];
  }
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
  double4_t acc = {0.0, 0.0, 0.0, 0.0};
  const int m = lane & 15, kk = lane >> 4;
#pragma unroll
  for (int q = 0; q < 4; q++) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(wi[(4 * q + kk) * 16 + m], wj[(4 * q + kk) * 16 + m], acc, 0, 0, 0);
  const int col = jt * 16 + m;
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = it * 16 + kk + 4 * q;
    if (row < D && col < LA && col >= row) p.A[(size_t)row * LA + col] -= acc[q];
  }
}

// P' = P - Y^T Y,  Y = [:, D : D+N] of the factorised augmented matrix     one wavefront per 16x16 tile of P
__global__ void __launch_bounds__(256) k_ekf_pupdate(EkfParams p) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int tn = (p.N + 15) / 16;
  if (tile >= tn * tn || (p.pred && *p.pred == 0)) return;
  const int r0 = (tile / tn) * 16, c0 = (tile % tn) * 16;
  const double *Y = p.Y + p.D;
  auto fa = [&](int i, int k) { const int r = r0 + i; return (r < p.N) ? Y[(size_t)k * p.LA + r] : 0.0; };
  auto fb = [&](int k, int j) { const int c = c0 + j; return (c < p.N) ? Y[(size_t)k * p.LA + c] : 0.0; };
  const double4_t acc = mfma_tile(fa, fb, p.D, lane);
  const int col = c0 + (lane & 15);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = r0 + (lane >> 4) + 4 * q;
    if (row < p.N && col < p.N) {
      const double v = p.P[(size_t)row * p.N + col] - acc[q];
      p.P[(size_t)row * p.N + col] = v;
      if (row == col && v < 0.0) p.flags[1] = 1; // StateHelper.cpp:172-182
    }
  }
}

// dx = Y^T y     (one thread per state dof)
__device__ __forceinline__ void ekf_dx_item(const EkfParams &p, int i) {
  if (i >= p.N) return;
  if (ekf_skipped(p)) { // no update: the correction is zero
    p.dx[i] = 0.0;
    return;
  }
  const double *Y = p.Y + p.D;
  const double *y = p.Y + p.D + p.N;
  double s0 = 0.0, s1 = 0.0, s2 = 0.0, s3 = 0.0;
  int r = 0;
  // 64 loads in flight per trip (round 5): the dot product is a chain of memory round trips — at 8 loads per trip its 52 trips were 26 of
  // k_tf_tail's 29 us, all of them in its last block.  The four accumulators take the same rows in the same order as before.
  for (; r + 32 <= p.D; r += 32) {
    double a[32], b[32];
#pragma unroll
    for (int j = 0; j < 32; j++) a[j] = Y[(size_t)(r + j) * p.LA + i], b[j] = y[(size_t)(r + j) * p.LA];
#pragma unroll
    for (int j = 0; j < 32; j += 4) {
      s0 = fma(a[j], b[j], s0);
      s1 = fma(a[j + 1], b[j + 1], s1);
      s2 = fma(a[j + 2], b[j + 2], s2);
      s3 = fma(a[j + 3], b[j + 3], s3);
    }
  }
  if (r + 16 <= p.D) { // (D = 208 = 6 x 32 + 16)
    double a[16], b[16];
#pragma unroll
    for (int j = 0; j < 16; j++) a[j] = Y[(size_t)(r + j) * p.LA + i], b[j] = y[(size_t)(r + j) * p.LA];
#pragma unroll
    for (int j = 0; j < 16; j += 4) {
      s0 = fma(a[j], b[j], s0);
      s1 = fma(a[j + 1], b[j + 1], s1);
      s2 = fma(a[j + 2], b[j + 2], s2);
      s3 = fma(a[j + 3], b[j + 3], s3);
    }
    r += 16;
  }
  for (; r + 4 <= p.D; r += 4) { // 8 independent loads per trip
    s0 = fma(Y[(size_t)r * p.LA + i], y[(size_t)r * p.LA], s0);
    s1 = fma(Y[(size_t)(r + 1) * p.LA + i], y[(size_t)(r + 1) * p.LA], s1);
    s2 = fma(Y[(size_t)(r + 2) * p.LA + i], y[(size_t)(r + 2) * p.LA], s2);
    s3 = fma(Y[(size_t)(r + 3) * p.LA + i], y[(size_t)(r + 3) * p.LA], s3);
  }
  for (; r < p.D; r++) s0 = fma(Y[(size_t)r * p.LA + i], y[(size_t)r * p.LA], s0);
  p.dx[i] = (s0 + s1) + (s2 + s3);
}
__global__ void k_ekf_dx(EkfParams p) { ekf_dx_item(p, blockIdx.x * blockDim.x + threadIdx.x); }


// ---------------------------------------------------------------------------------------------------
// EKF update from the Gram matrix of the stack (k_gram.h), in coordinates whitened by the PRIOR.
//
// With G = H^T H, g = H^T r (columns of the involved variables "D"), P_DD = U1^T U1 and B = U1^-T P(D, :):
//     T = I + U1 G U1^T / sigma^2  = C^T C          (symmetric, eigenvalues >= 1: the Cholesky never sees a small pivot)
//     P' = P - B^T B + Y2^T Y2,   Y2 = C^-T B        (= P - P(:,D) (P_DD^-1 - P_DD^-1 P'_DD P_DD^-1) P(D,:), P'_DD = U1^T T^-1 U1)
//     dx = Y2^T y2,               y2 = C^-T (U1 g / sigma^2)
// which is StateHelper::EKFUpdate (StateHelper.cpp:116-197) written for the information H^T R^-1 H instead of H.  Why not
// R = chol(G) followed by the usual update: G is singular along the unobservable directions and weak wherever few features
// constrain a variable; its Cholesky factor drops / garbles information of order eps |G| there, and the filter's covariance is
// LARGE exactly in those directions — measured on the 52-frame closed loop: trajectory off by 6e-6 against 7e-14 with the
// Householder compression.  Here every rounding error is relative to the prior: delta P' / P ~ eps |T|.
// Both factorisations run through k_ekf_chol_step (carry columns [B | h]).
// ---------------------------------------------------------------------------------------------------
struct TformParams {
  int N, D, LA, LG;
  const int32_t *col_cov;
  const double *G;   // [LG x LG] Gram matrix of [H | r], symmetric
  const double *P;   // [N x N]
  double *A;         // [D x LA] work matrix of the factorisation in flight
  const double *Y1;  // [D x LA] = [U1 | B | 0] once the first factorisation is done
  double *W;         // [D x D]
  double *diag0;     // [D] diagonal of P_DD (k_tf_gather)
  double inv_sigma2;
  int whitened;      // 1: G is the Gram matrix of the WHITENED stack [H U1^T | r] (k_system with SysParams::Lw): T = I + G / sigma^2
  double *Lw;        // [D x D] L = U1^T with explicit zeros above the diagonal (k_tf_lt), read by the per-feature kernel
  const int32_t *go; // 0: the prior block was not positive definite — everything after its factorisation is skipped (nothing
                     // is written to P, dx = 0), and the host repeats the update through the Householder route
};

// go = the first factorisation succeeded (flags[0] is its not-SPD flag)
__global__ void k_tf_go(const int32_t *flags, int32_t *go) { go[0] = flags[0] == 0 ? 1 : 0; }

// A = [P_DD | P(D, :) | 0]      one thread per element
__global__ void k_tf_gather(TformParams p) {
  const int64_t e = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= (int64_t)p.D * p.LA) return;
  const int r = (int)(e / p.LA), c = (int)(e - (int64_t)r * p.LA);
  const double *Pr = p.P + (size_t)p.col_cov[r] * p.N;
  p.A[e] = c < p.D ? Pr[p.col_cov[c]] : (c < p.D + p.N ? Pr[c - p.D] : 0.0);
  if (c == r) p.diag0[r] = Pr[p.col_cov[r]];
}

// L = U1^T, row-major, zeros above the diagonal      one thread per element
__global__ void k_tf_lt(TformParams p) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.D * p.D) return;
  const int s = e / p.D, c = e - s * p.D;
  p.Lw[e] = c <= s ? p.Y1[(size_t)c * p.LA + s] : 0.0;
}

// whitened Gram matrix: A[:, 0:D] = I + G / sigma^2      one thread per element
__global__ void k_tf_a(TformParams p) {
  const int e = blockIdx.x * blockDim.x + threadIdx.x;
  if (e >= p.D * p.D || *p.go == 0) return;
  const int r = e / p.D, c = e - r * p.D;
  p.A[(size_t)r * p.LA + c] = p.G[(size_t)r * p.LG + c] * p.inv_sigma2 + (r == c ? 1.0 : 0.0);
}

// W = G U1^T       one wavefront per 16x16 tile
__global__ void __launch_bounds__(256) k_tf_w(TformParams p) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int tm = (p.D + 15) / 16;
  if (tile >= tm * tm || *p.go == 0) return;
  const int r0 = (tile / tm) * 16, c0 = (tile % tm) * 16;
  auto fa = [&](int i, int k) { const int r = r0 + i; return r < p.D ? p.G[(size_t)r * p.LG + k] : 0.0; };
  auto fb = [&](int k, int j) { const int c = c0 + j; return (c < p.D && k >= c) ? p.Y1[(size_t)c * p.LA + k] : 0.0; };
  const double4_t acc = mfma_tile(fa, fb, p.D, lane);
  const int col = c0 + (lane & 15);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = r0 + (lane >> 4) + 4 * q;
    if (row < p.D && col < p.D) p.W[(size_t)row * p.D + col] = acc[q];
  }
}

// A[:, 0:D] = I + U1 W / sigma^2      one wavefront per 16x16 tile
__global__ void __launch_bounds__(256) k_tf_t(TformParams p) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int tm = (p.D + 15) / 16;
  if (tile >= tm * tm || *p.go == 0) return;
  const int r0 = (tile / tm) * 16, c0 = (tile % tm) * 16;
  auto fa = [&](int i, int k) { const int r = r0 + i; return (r < p.D && k >= r) ? p.Y1[(size_t)r * p.LA + k] : 0.0; };
  auto fb = [&](int k, int j) { const int c = c0 + j; return c < p.D ? p.W[(size_t)k * p.D + c] : 0.0; };
  const double4_t acc = mfma_tile(fa, fb, p.D, lane);
  const int col = c0 + (lane & 15);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = r0 + (lane >> 4) + 4 * q;
    if (row < p.D && col < p.D) p.A[(size_t)row * p.LA + col] = acc[q] * p.inv_sigma2 + (row == col ? 1.0 : 0.0);
  }
}

// A[:, D : D+N] = B (from Y1);  A[:, D+N] = h = U1 g / sigma^2       one wavefront per row
__global__ void __launch_bounds__(256) k_tf_bh(TformParams p) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (r >= p.D || *p.go == 0) return;
  const double *y = p.Y1 + (size_t)r * p.LA;
  double *a = p.A + (size_t)r * p.LA;
  for (int c = lane; c < p.N; c += 64) a[p.D + c] = y[p.D + c];
  if (p.whitened) { // the stack was whitened row by row: column D of its Gram matrix is U1 g already
    if (lane == 0) a[p.D + p.N] = p.G[(size_t)r * p.LG + p.D] * p.inv_sigma2;
    return;
  }
  double s = 0.0;
  for (int k = r + lane; k < p.D; k += 64) s = fma(y[k], p.G[(size_t)k * p.LG + p.D], s);
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
  if (lane == 0) a[p.D + p.N] = s * p.inv_sigma2;
}

// Whitened route, k_tf_go + k_tf_a + k_tf_bh in one launch (they were three 6-us launches between the Gram reduction and the second
// factorisation): one wavefront per row r of A = [I + G / sigma^2 | B | g / sigma^2]; `go` is derived from the first factorisation's
// flag by every wavefront and published by the first for the kernels that follow.
__global__ void __launch_bounds__(256) k_tf_abh(TformParams p, const int32_t *flags, int32_t *go_out) {
  const int lane = threadIdx.x & 63;
  const int r = blockIdx.x * 4 + (threadIdx.x >> 6);
  const bool go = flags[0] == 0;
  if (blockIdx.x == 0 && threadIdx.x == 0) go_out[0] = go ? 1 : 0;
  if (r >= p.D || !go) return;
  const double *y = p.Y1 + (size_t)r * p.LA, *gr = p.G + (size_t)r * p.LG;
  double *a = p.A + (size_t)r * p.LA;
  for (int c = lane; c < p.D; c += 64) a[c] = gr[c] * p.inv_sigma2 + (r == c ? 1.0 : 0.0);
  for (int c = lane; c < p.N; c += 64) a[p.D + c] = y[p.D + c];
  if (lane == 0) a[p.D + p.N] = gr[p.D] * p.inv_sigma2; // the stack was whitened row by row: column D of its Gram matrix is U1 g already
}

// P' = P - (B^T B - Y2^T Y2)      one wavefront per 16x16 tile of P; both products are symmetric tile by tile
__device__ __forceinline__ void tf_pupdate_tile(const EkfParams &p, const double *Y1, int tile, int lane) {
  const int tn = (p.N + 15) / 16;
  if (tile >= tn * tn || ekf_skipped(p)) return;
  const int r0 = (tile / tn) * 16, c0 = (tile % tn) * 16;
  const double *B = Y1 + p.D, *Y = p.Y + p.D;
  // Both products in ONE loop, 16 operand loads per trip issued together at clamped addresses and masked afterwards (two
  // mfma_tile calls were 26 dependent round trips to L2 of 8 loads each, behind per-lane branches: 20 of this kernel's 28 us).
  // Each accumulator still sums its k-slices in ascending order.
  const int li = lane & 15, kk = lane >> 4;
  const bool rok = r0 + li < p.N;
  const int rr = rok ? r0 + li : p.N - 1, cc = c0 + li < p.N ? c0 + li : p.N - 1;
  double4_t bb = {0.0, 0.0, 0.0, 0.0}, yy = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 4
  for (int k0 = 0; k0 < p.D; k0 += 16) {
    double a1[4], b1[4], a2[4], b2[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 + 4 * u + kk;
      const size_t off = (size_t)(k < p.D ? k : p.D - 1) * p.LA;
      a1[u] = B[off + rr], b1[u] = B[off + cc], a2[u] = Y[off + rr], b2[u] = Y[off + cc];
      if (!(rok && k < p.D)) a1[u] = 0.0, a2[u] = 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) {
      bb = __builtin_amdgcn_mfma_f64_16x16x4f64(a1[u], b1[u], bb, 0, 0, 0);
      yy = __builtin_amdgcn_mfma_f64_16x16x4f64(a2[u], b2[u], yy, 0, 0, 0);
    }
  }
  const int col = c0 + (lane & 15);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = r0 + (lane >> 4) + 4 * q;
    if (row < p.N && col < p.N) {
      const double v = p.P[(size_t)row * p.N + col] - (bb[q] - yy[q]); // nothing accepted: G = 0, Y2 = B bit for bit, P' = P exactly
      p.P[(size_t)row * p.N + col] = v;
      if (row == col && v < 0.0) p.flags[1] = 1; // StateHelper.cpp:172-182
    }
  }
}
__global__ void __launch_bounds__(256) k_tf_pupdate(EkfParams p, const double *Y1) {
  tf_pupdate_tile(p, Y1, blockIdx.x * 4 + (threadIdx.x >> 6), threadIdx.x & 63);
}

// ---------------------------------------------------------------------------
// box-plus of the resident tables (Type::update of the variables the GPU holds)
// ---------------------------------------------------------------------------
__device__ __forceinline__ void pose_boxplus(double *qp, const double *dx6) {
  // JPLQuat::update: q <- quatnorm([0.5 dtheta; 1]) (x) q   (quat_ops.h:180-200, :496-501)
  double d0 = 0.5 * dx6[0], d1 = 0.5 * dx6[1], d2 = 0.5 * dx6[2], d3 = 1.0;
  const double dn = sqrt(d0 * d0 + d1 * d1 + d2 * d2 + d3 * d3);
  d0 /= dn, d1 /= dn, d2 /= dn, d3 /= dn;
  const double p0 = qp[0], p1 = qp[1], p2 = qp[2], p3 = qp[3];
  // Qm = [ q4 I - [q x], q ; -q^T, q4 ] with q = dq
  double t0 = d3 * p0 + d2 * p1 - d1 * p2 + d0 * p3;
  double t1 = -d2 * p0 + d3 * p1 + d0 * p2 + d1 * p3;
  double t2 = d1 * p0 - d0 * p1 + d3 * p2 + d2 * p3;
  double t3 = -d0 * p0 - d1 * p1 - d2 * p2 + d3 * p3;
  if (t3 < 0.0) t0 = -t0, t1 = -t1, t2 = -t2, t3 = -t3;
  const double tn = sqrt(t0 * t0 + t1 * t1 + t2 * t2 + t3 * t3);
  qp[0] = t0 / tn, qp[1] = t1 / tn, qp[2] = t2 / tn, qp[3] = t3 / tn;
  qp[4] += dx6[3], qp[5] += dx6[4], qp[6] += dx6[5]; // PoseJPL.h:88
}

__device__ __forceinline__ void boxplus_item(int t, int C, int K, const double *dx, const int32_t *clone_cov, const int32_t *calib_cov,
                                             const int32_t *intr_cov, double *clone_qp, double *calib_qp, double *intr) {
  if (t < C) pose_boxplus(clone_qp + 7 * t, dx + clone_cov[t]);
  if (t < K) {
    if (calib_cov[t] >= 0) pose_boxplus(calib_qp + 7 * t, dx + calib_cov[t]);
    if (intr_cov[t] >= 0)
      for (int i = 0; i < 8; i++) intr[8 * t + i] += dx[intr_cov[t] + i]; // Vec::update, Vec.h:55-58
  }
}
__global__ void k_boxplus(int C, int K, const double *__restrict__ dx, const int32_t *__restrict__ clone_cov, const int32_t *__restrict__ calib_cov,
                          const int32_t *__restrict__ intr_cov, double *clone_qp, double *calib_qp, double *intr, const int32_t *pred) {
  if (pred && *pred == 0) return; // a zero correction would still renormalise the quaternions
  boxplus_item(blockIdx.x * blockDim.x + threadIdx.x, C, K, dx, clone_cov, calib_cov, intr_cov, clone_qp, calib_qp, intr);
}


// ---------------------------------------------------------------------------------------------------
// X = R L^-1 for the compressed factor of the WHITENED stack (mode A through the Gram route): the rows left the per-feature stage as
// [H L | r] with P_DD = L L^T, R is the Cholesky factor of their Gram matrix (k_gram_chol: R^T R = L^T H^T H L), so X^T X = H^T H and
// (X, c) is a compressed system of the reference's form for the stock StateHelper::EKFUpdate (X dense instead of triangular: nothing
// in EKFUpdate asks for a triangle).  One workgroup per 16 rows, 16 lanes per row: column j from the right,
//        x_j = (r_j - sum_{k > j} x_k L[k][j]) / L[j][j],      L[k][j] = U1[j][k] read along a row of Y1 = [U1 | ..] (coalesced).
// In place on the first D columns of R [D x LD]; the residual column stays.
// ---------------------------------------------------------------------------------------------------
// Blocked form on the matrix cores, one wavefront per 16 rows of X: for tile column tj from the right
//   S = R(:, J) - X(:, > J) L(> J, J)     v_mfma_f64_16x16x4_f64, A = X from LDS, B = rows of U1 read in place (L(k, j) = U1(j, k))
//   X(:, J) = S L(J, J)^-1                16 substitution steps inside the tile: lane (g, cl) holds rows 4 q + g of column cl (the
//                                         accumulator layout), column j's finished values reach the lanes left of it by a 16-lane shuffle
// Everything tile column tj - 1 needs from memory (its strip of U1, its diagonal tile, its tile of R) is requested while tile column
// tj is being computed: U1 was written on another XCD a moment ago, a read costs ~1.5 us, and the chain of NT columns would
// otherwise pay it once per trip of the product loop (measured: 164 us at D = 208 that way; the column-at-a-time form before it ran
// D dependent steps of LDS round trip + shuffle reduction, 162 us, and with its fetches inside the step 0.45 ms).
// NTM = tile columns at most: 16 (D <= 256, the next column's strip prefetched) or 24 (D <= 384: one strip — two would not fit the
// registers of a wavefront — fetched and used in turn; configs[4]'s mode A, round 4)
template <int NTM> struct UnwhitenStrip {
  double b[4 * (NTM - 1)]; // B operands of up to NTM - 1 tiles right of the column: b[4 t + u] = U1[col][16 (tj + 1 + t) + 4 u + g]
  double ud[16]; // row cl of the diagonal tile from the diagonal on, identity beyond D
  double rr[4];  // R(rows 4 q + g, col)
};
template <int NTM>
__global__ void __launch_bounds__(64) k_unwhiten(int D, int LD, double *__restrict__ R, const double *__restrict__ Y1, int LA, const int32_t *pred,
                                                 const int32_t *pred_not) {
  constexpr int XS = 16 * NTM + 4; // LDS row stride: the A-operand read X[cl][k0 + g] touches 64 different banks
  using Strip = UnwhitenStrip<NTM>;
  __shared__ double X[16 * XS];
  if ((pred && *pred == 0) || (pred_not && *pred_not != 0)) return; // (pred_not: the not-SPD flag of the prior block's factorisation)
  const int lane = threadIdx.x, g = lane >> 4, cl = lane & 15;
  const int r0 = blockIdx.x * 16, NT = (D + 15) >> 4;
  for (int e = lane; e < 16 * XS; e += 64) X[e] = 0.0;
  auto wsync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  auto fetch = [&](int tj, Strip &s) { // clamped addresses, masked afterwards: one basic block of loads
    const int c0 = 16 * tj, col = c0 + cl;
    const bool cok = col < D;
    const double *urow = Y1 + (size_t)(cok ? col : D - 1) * LA; // row `col` of U1 = column `col` of L
#pragma unroll
    for (int i = 0; i < 4 * (NTM - 1); i++) {
      const int k = c0 + 16 + 16 * (i >> 2) + 4 * (i & 3) + g;
      s.b[i] = urow[k < D ? k : D - 1];
    }
#pragma unroll
    for (int j = 0; j < 16; j++) s.ud[j] = urow[c0 + j < D ? c0 + j : D - 1];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int row = r0 + 4 * q + g;
      s.rr[q] = R[(size_t)(row < D ? row : D - 1) * LD + (cok ? col : 0)];
    }
#pragma unroll
    for (int i = 0; i < 4 * (NTM - 1); i++) {
      const int k = c0 + 16 + 16 * (i >> 2) + 4 * (i & 3) + g;
      if (!(cok && k < D)) s.b[i] = 0.0;
    }
#pragma unroll
    for (int j = 0; j < 16; j++)
      if (!(cok && c0 + j < D && j >= cl)) s.ud[j] = j == cl ? 1.0 : 0.0;
#pragma unroll
    for (int q = 0; q < 4; q++)
      if (!(r0 + 4 * q + g < D && cok)) s.rr[q] = 0.0;
  };
  auto column = [&](int tj, const Strip &s) {
    const int c0 = 16 * tj, col = c0 + cl;
    const bool cok = col < D;
    double4_t acc = {s.rr[0], s.rr[1], s.rr[2], s.rr[3]}, acc2 = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int t = 0; t < NTM - 1; t++) {
      const int k0 = c0 + 16 + 16 * t;
      if (k0 < 16 * NT) { // (wave-uniform)
        const double *xa = X + cl * XS + k0 + g;
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[0], s.b[4 * t + 0], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[4], s.b[4 * t + 1], acc2, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[8], s.b[4 * t + 2], acc, 0, 0, 0);
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(-xa[12], s.b[4 * t + 3], acc2, 0, 0, 0);
      }
    }
    acc += acc2;
    double dinv = 1.0;
#pragma unroll
    for (int j = 0; j < 16; j++)
      if (j == cl) dinv = 1.0 / s.ud[j];
#pragma unroll
    for (int j = 15; j >= 0; j--) {
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double xq = acc[q] * dinv;           // meaningful in the lanes of column j
        const double xj = __shfl(xq, j, 16);       // ... and from there to the whole 16-lane group (same rows 4 q + g)
        acc[q] = cl == j ? xq : (cl < j ? fma(-xj, s.ud[j], acc[q]) : acc[q]);
      }
    }
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const int row = r0 + 4 * q + g;
      X[(4 * q + g) * XS + col] = cok ? acc[q] : 0.0;
      if (row < D && cok) R[(size_t)row * LD + col] = acc[q];
    }
    wsync();
  };
  if constexpr (NTM <= 16) {
    Strip s0, s1;
    fetch(NT - 1, s0);
    wsync();
    for (int tj = NT - 1; tj >= 0; tj -= 2) {
      if (tj >= 1) fetch(tj - 1, s1);
      column(tj, s0);
      if (tj < 1) break;
      if (tj >= 2) fetch(tj - 2, s0);
      column(tj - 1, s1);
    }
  } else {
    Strip s0;
    wsync();
    for (int tj = NT - 1; tj >= 0; tj--) {
      fetch(tj, s0);
      column(tj, s0);
    }
  }
}

} // namespace ovg
