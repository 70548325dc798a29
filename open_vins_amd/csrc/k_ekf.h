// k_ekf.h — StateHelper::EKFUpdate on the device (StateHelper.cpp:116-197) and the box-plus of
// the resident pose tables (JPLQuat.h:114-125, PoseJPL.h:74-91, Vec.h:55-58).
//
// With the compressed system [R | c] (D x D upper triangular, R_noise = sigma^2 I):
//   Mt = R P(cols,:)                       (D x N)   = (P H^T)^T            StateHelper.cpp:137-146
//   S  = R P(cols,cols) R^T + sigma^2 I    (D x D)                          :151-156
//   S  = U^T U ;  Y = U^-T Mt ; y = U^-T c           (replaces Sinv, K)     :160-162
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
  for (int k0 = 0; k0 < K; k0 += 4) {
    const int k = k0 + kk;
    const double a = (k < K) ? fa(i, k) : 0.0;
    const double b = (k < K) ? fb(k, i) : 0.0;
    acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0);
  }
  return acc;
}

struct EkfParams {
  int N, D, LD, LA; // LA = D + N + 1 columns of the augmented matrix
  const double *R;  // [D x LD] compressed system (last column = residual)
  const int32_t *col_cov;
  double *P;        // [N x N] in/out
  double *Mt;       // [D x N]
  double *A;        // [D x LA] augmented [S | Mt | c]
  double *dx;       // [N]
  int32_t *flags;   // [0] = 1 if S not SPD, [1] = 1 if a diagonal of P' is negative
  double sigma2;
};

// Mt = R * P(cols, :)     grid: tiles(D/16) x tiles(N/16) wavefronts, 4 per block
__global__ void __launch_bounds__(256) k_ekf_mt(EkfParams p) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int tn = (p.N + 15) / 16, tm = (p.D + 15) / 16;
  if (tile >= tn * tm) return;
  const int r0 = (tile / tn) * 16, c0 = (tile % tn) * 16;
  auto fa = [&](int i, int k) { const int r = r0 + i; return (r < p.D && k >= r) ? p.R[(size_t)r * p.LD + k] : 0.0; }; // R upper triangular
  auto fb = [&](int k, int j) { const int c = c0 + j; return (c < p.N) ? p.P[(size_t)p.col_cov[k] * p.N + c] : 0.0; };
  const double4_t acc = mfma_tile(fa, fb, p.D, lane);
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
  if (tile >= tm * tm) return;
  const int r0 = (tile / tm) * 16, c0 = (tile % tm) * 16;
  auto fa = [&](int i, int k) { const int r = r0 + i; return (r < p.D) ? p.Mt[(size_t)r * p.N + p.col_cov[k]] : 0.0; };
  auto fb = [&](int k, int j) { const int c = c0 + j; return (c < p.D && k >= c) ? p.R[(size_t)c * p.LD + k] : 0.0; };
  const double4_t acc = mfma_tile(fa, fb, p.D, lane);
  const int col = c0 + (lane & 15);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = r0 + (lane >> 4) + 4 * q;
    if (row < p.D && col < p.D) p.A[(size_t)row * p.LA + col] = acc[q] + (row == col ? p.sigma2 : 0.0);
  }
  if (tile == 0)
    for (int r = lane; r < p.D; r += 64) p.A[(size_t)r * p.LA + p.D + p.N] = p.R[(size_t)r * p.LD + p.D];
}

// Blocked right-looking Cholesky S = U^T U of the leading D x D block of A, carried through the
// augmented columns: on exit A[:, D:] = U^-T [Mt | c].  Single workgroup, 1024 threads; the
// matrix (D x LA doubles, < 1.5 MB) stays in L2.
static constexpr int CH_NB = 16;
__global__ void __launch_bounds__(1024) k_ekf_chol(EkfParams p) {
  extern __shared__ __attribute__((aligned(16))) double lds[];
  double *Ukk = lds;                // [16 x 16]
  double *Wp = lds + CH_NB * CH_NB; // [16 x LA]
  const int tid = threadIdx.x, NT = blockDim.x;
  const int D = p.D, LA = p.LA;
  for (int kb = 0; kb < D; kb += CH_NB) {
    const int nb = min(CH_NB, D - kb);
    // 1. diagonal block (upper part) to LDS, unblocked factorisation by the first 256 threads
    if (tid < CH_NB * CH_NB) {
      const int i = tid / CH_NB, j = tid % CH_NB;
      Ukk[tid] = (i < nb && j < nb && j >= i) ? p.A[(size_t)(kb + i) * LA + kb + j] : (i == j ? 1.0 : 0.0);
    }
    __syncthreads();
    for (int k = 0; k < nb; k++) {
      if (tid == 0) {
        const double d = Ukk[k * CH_NB + k];
        if (!(d > 0.0)) p.flags[0] = 1;
        Ukk[k * CH_NB + k] = sqrt(d);
      }
      __syncthreads();
      if (tid > k && tid < nb) Ukk[k * CH_NB + tid] /= Ukk[k * CH_NB + k];
      __syncthreads();
      if (tid < CH_NB * CH_NB) {
        const int i = tid / CH_NB, j = tid % CH_NB;
        if (i > k && j >= i && j < nb) Ukk[i * CH_NB + j] -= Ukk[k * CH_NB + i] * Ukk[k * CH_NB + j];
      }
      __syncthreads();
    }
    // 2. row panel: W = U_kk^-T A[kb:kb+nb, kb+nb:LA]  (thread per column), kept in LDS and written back
    for (int j = kb + nb + tid; j < LA; j += NT) {
      double w[CH_NB];
#pragma unroll
      for (int l = 0; l < CH_NB; l++) w[l] = (l < nb) ? p.A[(size_t)(kb + l) * LA + j] : 0.0;
#pragma unroll
      for (int l = 0; l < CH_NB; l++) {
        double s = w[l];
#pragma unroll
        for (int q = 0; q < CH_NB; q++)
          if (q < l) s = fma(-Ukk[q * CH_NB + l], w[q], s);
        w[l] = s / Ukk[l * CH_NB + l];
      }
#pragma unroll
      for (int l = 0; l < CH_NB; l++) {
        Wp[(size_t)l * LA + j] = w[l];
        if (l < nb) p.A[(size_t)(kb + l) * LA + j] = w[l];
      }
    }
    // write the factored diagonal block back
    if (tid < CH_NB * CH_NB) {
      const int i = tid / CH_NB, j = tid % CH_NB;
      if (i < nb && j < nb && j >= i) p.A[(size_t)(kb + i) * LA + kb + j] = Ukk[tid];
    }
    __syncthreads();
    // 3. trailing update: A[i][j] -= sum_l W[l][i] W[l][j]   for kb+nb <= i < D, j >= i
    {
      const int tj = tid & 511, ti = tid >> 9; // 512 column lanes x 2 row phases
      for (int j = kb + nb + tj; j < LA; j += 512) {
        double wj[CH_NB];
#pragma unroll
        for (int l = 0; l < CH_NB; l++) wj[l] = Wp[(size_t)l * LA + j];
        const int imax = min(j, D - 1);
        for (int i = kb + nb + ti; i <= imax; i += 2) {
          double s = 0.0;
#pragma unroll
          for (int l = 0; l < CH_NB; l++) s = fma(Wp[(size_t)l * LA + i], wj[l], s);
          p.A[(size_t)i * LA + j] -= s;
        }
      }
    }
    __syncthreads();
  }
}

// P' = P - Y^T Y,  Y = A[:, D : D+N]     one wavefront per 16x16 tile of P
__global__ void __launch_bounds__(256) k_ekf_pupdate(EkfParams p) {
  const int lane = threadIdx.x & 63;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  const int tn = (p.N + 15) / 16;
  if (tile >= tn * tn) return;
  const int r0 = (tile / tn) * 16, c0 = (tile % tn) * 16;
  const double *Y = p.A + p.D;
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
__global__ void k_ekf_dx(EkfParams p) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.N) return;
  const double *Y = p.A + p.D;
  const double *y = p.A + p.D + p.N;
  double s = 0.0;
  for (int r = 0; r < p.D; r++) s = fma(Y[(size_t)r * p.LA + i], y[(size_t)r * p.LA], s);
  p.dx[i] = s;
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

__global__ void k_boxplus(int C, int K, const double *__restrict__ dx, const int32_t *__restrict__ clone_cov, const int32_t *__restrict__ calib_cov,
                          const int32_t *__restrict__ intr_cov, double *clone_qp, double *calib_qp, double *intr) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  if (t < C) pose_boxplus(clone_qp + 7 * t, dx + clone_cov[t]);
  if (t < K) {
    if (calib_cov[t] >= 0) pose_boxplus(calib_qp + 7 * t, dx + calib_cov[t]);
    if (intr_cov[t] >= 0)
      for (int i = 0; i < 8; i++) intr[8 * t + i] += dx[intr_cov[t] + i]; // Vec::update, Vec.h:55-58
  }
}

} // namespace ovg
