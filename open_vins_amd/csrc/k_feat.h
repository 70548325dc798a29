// k_feat.h — the MSCKF fast path of the per-feature stage: Jacobian, chi2 gate on the matrix cores, nullspace projection and
// prior-whitened stacking of ONE feature per workgroup, several workgroups per compute unit.
//
//   UpdaterHelper::get_feature_jacobian_full            UpdaterHelper.cpp:192-424
//   UpdaterHelper::nullspace_project_inplace            UpdaterHelper.cpp:426-454
//   chi2 gate                                           UpdaterMSCKF.cpp:209-234, StateHelper.cpp:226-254
//   stacking into Hx_big / res_big                      UpdaterMSCKF.cpp:237-255
//
// Same mathematics as k_system (k_system.h: sparse rows, S0 = H P H^T + s^2 I on the unprojected rows with the right-hand
// sides [r | H_f], three Householder reflectors of H_f); what changes is where things live and what executes them:
//
//  * The gate matrix never touches LDS.  Its 16 x 16 tiles (upper triangle + one tile column of right-hand sides) are dealt
//    round-robin to the wavefronts and stay in REGISTERS in the accumulator layout of v_mfma_f64_16x16x4_f64 (lane (g, c) holds
//    rows g, g+4, g+8, g+12 of column c).  That layout is at once the B-operand layout of a tile and the A-operand layout of
//    its transpose, so the blocked Cholesky S0 = U^T U runs on the matrix cores without a single register shuffle:
//        W_kj = U_kk^-T S_kj        A = U_kk^-1 read from LDS in accumulator order, B = the tile's own registers
//        S_ij -= W_ki^T W_kj        A, B = panel tiles read from LDS in accumulator order
//    and U_kk^-1 comes out of the diagonal tile's factorisation for free: 16 extra lanes carry the identity through the same
//    instruction stream (row k scaled, rows below updated with the broadcast multipliers), which turns it into U_kk^-T.
//    LDS per workgroup drops from 117 KB to ~60 KB (two workgroups per CU at 30 clones x 2 cameras).
//  * T = H P is swept chunk by chunk INSIDE this kernel (thread per column, two columns per lane); the whitened, projected rows
//    Q^T [H L | r] for the Gram accumulation are a SECOND sweep in k_feat_out (round 2 moved them out: the gate needs only P, so the
//    prior block's factorisation and k_feat_qr / k_feat_z run beside it on the second stream).  Round 3's k_featy.h replaces both
//    sweeps by one on the matrix cores; these kernels remain as the legacy form (ovgpu_debug_option "legacy_feature_kernel").
//  * The Jacobian rows come from a thread-per-measurement pre-kernel (k_feat_rows) and the reflectors of H_f with z = T^T V^T [H L | r]
//    from a wavefront-per-feature pre-kernel (k_feat_qr): both are tiny, but inlined into the per-feature kernel they cost it a
//    third of its time (60 busy threads of 256, sin / cos / sqrt code, dependent L2 round trips of the dense product with L).
//    In the sweeps every Jacobian value is wave-uniform: read from the row store in HBM through the scalar cache it is an SGPR
//    operand of the multiply-add, not an LDS read per lane.
//    Measured alternatives of the sweeps (800 features, 30 clones x 2 cameras; this form 0.44 ms): LDS-broadcast operands 0.52 ms;
//    T = H P and Y = H L as dense 16 x 80 tile products on the matrix cores 0.59 ms (the clone blocks are zero-padded 8x and
//    FP64 MFMA has no rate advantage over FP64 FMA); calibration part on the matrix cores + clone part on the vector units,
//    at 2 or at 4 wavefronts per SIMD, 0.48 - 0.54 ms.
//    Measured alternatives of the gate matrix's tiles (2000 features; the per-lane dot products below: 191 kcycles per workgroup and
//    update): T chunk x sparse-gathered Jacobian rows on the matrix cores, 4-column steps selected by a per-tile-row bit mask, four
//    steps in flight: 246 kcycles (each step needs a gathered LDS operand and its own select; the 17 useful steps of a tile do not
//    amortise the set-up).  Jacobian operands of the T sweep as LDS broadcast reads instead of scalar loads: 391 vs 278 kcycles.
//  * Work is handed out through an atomic counter, longest tracks first.
//
// Restrictions (the host falls back to k_system otherwise): MSCKF features (3 projected columns, no landmark columns), global
// representations, one noise level, whitened output, tracks of at most 8 * NTmax measurements.
#pragma once
#include "k_system.h"

namespace ovg {
namespace feat {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int FT_CH = 8; // measurements per chunk = 16 rows = one tile row of the gate matrix

struct FeatLds {
  size_t minfo, rows, rhs, big, stage, sched, total;
};

// nt_max = tile rows of the longest track the instantiation accepts
__host__ __device__ inline FeatLds feat_lds_layout(int m_max, int RS, int D, int LD, int KC, int nt_max) {
  FeatLds L;
  size_t o = 0;
  auto take = [&](size_t bytes) {
    const size_t at = o;
    o += (bytes + 15) & ~(size_t)15;
    return at;
  };
  L.minfo = take((size_t)m_max * 8 * sizeof(int));
  L.rows = take((size_t)m_max * RS * sizeof(double));
  L.rhs = take((size_t)16 * nt_max * 4 * sizeof(double));
  const size_t tch = (size_t)2 * FT_CH * D, pan = (size_t)(nt_max + 1) * 256;
  L.big = take((tch > pan ? tch : pan) * sizeof(double));
  L.stage = take(2 * 256 * sizeof(double));
  L.sched = take(4 * sizeof(int));
  L.total = o;
  return L;
}

__device__ __forceinline__ double bcast_lane(double v, int lane) { // lane is wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// 1 / sqrt(d) and sqrt(d) to double precision from one v_rsq_f64 seed and Newton steps; NaN / inf for a non-positive d, like a
// broken-down LLT (no branch: the pivot chain of the tile factorisation stays one basic block)
__device__ __forceinline__ void rsqrt_pair(double dk, double &inv, double &d) {
  inv = __builtin_amdgcn_rsq(dk);
#pragma unroll
  for (int it = 0; it < 2; it++) inv = fma(0.5 * inv, fma(-dk * inv, inv, 1.0), inv);
  d = dk * inv;
  d = fma(0.5 * inv, fma(-d, d, dk), d);
}

// Cholesky of one 16 x 16 diagonal tile, in one wavefront.  st0: the tile, row-major (upper triangle read).  Lane j < 16 holds
// column j of the tile, lane 16 + j column j of the identity; step k scales row k by 1 / U_kk and subtracts U_ki * (row k) from
// every row i below, the multipliers broadcast from the matrix lanes.  The identity lanes end up with the columns of U^-T =
// rows of U^-1, written row-major to st1.  The next pivot's reciprocal square root is started as soon as its row is updated,
// ahead of the other rows' updates (the chain of a step is the 16 dependent rsq / Newton sequences, not the updates).
// WRITE_U: U itself goes back to st0 (row-major, zeros below the diagonal).  Returns true when a pivot is not above
// tol * diag0[k] (above 0 without diag0); first nb pivots only: the rest belong to the identity padding of a partial last tile.
template <bool WRITE_U> __device__ __forceinline__ bool diag_tile_factor_t(double *st0, double *st1, int lane, const double *diag0, double tol, int nb) {
  const int j = lane & 15;
  const bool rhsl = (lane >> 4) == 1;
  double u[16];
#pragma unroll
  for (int i = 0; i < 16; i++) {
    const double a = st0[i * 16 + j];
    u[i] = rhsl ? (i == j ? 1.0 : 0.0) : (i <= j ? a : 0.0);
  }
  bool bad = false;
  double dk = bcast_lane(u[0], 0), inv, d;
  rsqrt_pair(dk, inv, d);
#pragma unroll
  for (int k = 0; k < 16; k++) {
    if (WRITE_U && k < nb) bad = bad || !(dk > (diag0 ? tol * diag0[k] : 0.0));
    u[k] = (!rhsl && j == k) ? d : u[k] * inv;
    if (k < 15) {
      u[k + 1] = fma(-bcast_lane(u[k], k + 1), u[k], u[k + 1]); // U[k][i] comes from matrix lane i
      dk = bcast_lane(u[k + 1], k + 1);
      rsqrt_pair(dk, inv, d);
#pragma unroll
      for (int i = k + 2; i < 16; i++) u[i] = fma(-bcast_lane(u[k], i), u[k], u[i]);
    }
  }
  if (rhsl) {
#pragma unroll
    for (int i = 0; i < 16; i++) st1[j * 16 + i] = u[i]; // row j of U^-1
  } else if (WRITE_U && lane < 16) {
#pragma unroll
    for (int i = 0; i < 16; i++) st0[i * 16 + j] = i <= j ? u[i] : 0.0; // column j of U
  }
  return bad;
}
__device__ __forceinline__ void diag_tile_factor(double *st0, double *st1, int lane) { (void)diag_tile_factor_t<false>(st0, st1, lane, nullptr, 0.0, 16); }
__device__ __forceinline__ bool diag_tile_factor_u(double *st0, double *st1, int lane, const double *diag0, double tol, int nb) {
  return diag_tile_factor_t<true>(st0, st1, lane, diag0, tol, nb);
}

#define FEAT_MFMA(a, b, c) c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0)

// Cholesky of one 16 x 16 diagonal tile by a whole wavefront, blocked by 4 with the rank-4 updates on the matrix cores.
//   s   in: the tile S in accumulator layout (lane (g, c): rows g, g+4, g+8, g+12 of column c); out: U (upper triangular, S = U^T U)
//   e   out: U^-T in accumulator layout (the identity carried through the same elimination)
//   sh  128 doubles of LDS scratch private to the wavefront
// Block step b (rows / columns 4b .. 4b+3): the four rows of [S | I] go through LDS; EVERY lane factors the 4 x 4 diagonal block
// redundantly (four dependent rsq / Newton sequences, no cross-lane traffic) and forward-substitutes its own column of both
// halves, which gives rows 4b .. 4b+3 of U and of U^-T; the remaining rows take ONE v_mfma_f64_16x16x4_f64 each:
// S -= W^T W, E -= W^T W2, with lane (g, c) supplying W[g][c] as the A and as the B operand.
// The lane-per-column form above needs 120 lane broadcasts per tile (~10 k cycles); this one ~3.5 k.
// Returns true when a pivot is not above tol * diag0[k] (above 0 without diag0); pivots k >= nb belong to the identity padding.
__device__ __forceinline__ bool diag_tile_factor_blk(d4 &s, d4 &e, double *sh, int lane, const double *diag0, double tol, int nb) {
  const int g = lane >> 4, cl = lane & 15;
#pragma unroll
  for (int q = 0; q < 4; q++) e[q] = (g + 4 * q == cl) ? 1.0 : 0.0;
  bool bad = false;
#pragma unroll
  for (int b = 0; b < 4; b++) {
    const int kb = 4 * b;
    sh[g * 16 + cl] = s[b], sh[64 + g * 16 + cl] = e[b];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    double a[4][4], sv[4], ev[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
#pragma unroll
      for (int c2 = r; c2 < 4; c2++) a[r][c2] = sh[r * 16 + kb + c2]; // the diagonal block (upper part): same address in every lane
      sv[r] = cl >= kb ? sh[r * 16 + cl] : 0.0;                          // columns left of the block are finished: nothing to eliminate
      ev[r] = sh[64 + r * 16 + cl];
    }
    // 4 x 4 Cholesky a = u^T u, every lane for itself; w = u^-T sv, w2 = u^-T ev alongside
    double u[4][4], inv[4], w[4], w2[4];
#pragma unroll
    for (int r = 0; r < 4; r++) {
      double dk = a[r][r];
#pragma unroll
      for (int t = 0; t < r; t++) dk = fma(-u[t][r], u[t][r], dk);
      if (kb + r < nb) bad = bad || !(dk > (diag0 ? tol * diag0[kb + r] : 0.0));
      double d;
      rsqrt_pair(dk, inv[r], d);
      u[r][r] = d;
#pragma unroll
      for (int c2 = r + 1; c2 < 4; c2++) {
        double v = a[r][c2];
#pragma unroll
        for (int t = 0; t < r; t++) v = fma(-u[t][r], u[t][c2], v);
        u[r][c2] = v * inv[r];
      }
      double x = sv[r], y = ev[r];
#pragma unroll
      for (int t = 0; t < r; t++) x = fma(-u[t][r], w[t], x), y = fma(-u[t][r], w2[t], y);
      w[r] = x * inv[r], w2[r] = y * inv[r];
      if (cl >= kb && cl < kb + r) w[r] = 0.0; // the block's own columns: w is the column of U11, exactly zero below its diagonal
    }
    // own row of the block: lane (g, c) holds row kb + g
    double wg = w[0], w2g = w2[0];
#pragma unroll
    for (int r = 1; r < 4; r++) wg = g == r ? w[r] : wg, w2g = g == r ? w2[r] : w2g;
    s[b] = wg, e[b] = w2g;
    if (b < 3) { // rows below the block
      d4 us = {0.0, 0.0, 0.0, 0.0}, ue = {0.0, 0.0, 0.0, 0.0};
      FEAT_MFMA(-wg, wg, us);
      FEAT_MFMA(-wg, w2g, ue);
#pragma unroll
      for (int q = 0; q < 4; q++) {
        if (q > b) s[q] += us[q], e[q] += ue[q];
      }
    }
    __builtin_amdgcn_wave_barrier(); // sh is rewritten by the next block step
  }
  return bad;
}

// Row store in HBM, written by k_feat_rows: per measurement RS doubles (the sparse Jacobian rows of k_system.h, offsets RO_*) and
// 8 ints (camera, clone, first column of the clone / extrinsic / intrinsic block, their covariance ids).
struct FeatStore {
  double *rows;   // [M][RS]
  int32_t *minfo; // [M][8]
  double *V;      // [2M][3]  reflectors of H_f, row 2 gm + a
  double *z;      // [F][3][LD]  T^T V^T [H L | r]
  const int32_t *meas_feat; // [M] feature of each measurement
  double *w;                // [F][3][LD] T^T V^T [H | r] of every feature (k_feat_qr), the left operand of k_feat_z
};

// ---------------------------------------------------------------------------------------------------
// k_feat_rows: one thread per measurement -> its two Jacobian rows (UpdaterHelper.cpp:314-421)
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_feat_rows(SysParams p, FeatStore st, int M) {
  const int gm = blockIdx.x * 256 + threadIdx.x;
  if (gm >= M) return;
  const int f = st.meas_feat[gm];
  if (p.status[f] != OVGPU_FEAT_USED) return;
  const V3 p_FinG = load_v3(p.p_FinG + 3 * f); // fej == value for MSCKF features (UpdaterMSCKF.cpp:186-194)
  double hq[21];
  double *dl = hq + 12;
  if (p.opt.feat_rep == OVGPU_REP_GLOBAL_FULL_INVERSE_DEPTH) inv_depth_jac(p_FinG, dl); // UpdaterHelper.cpp:46
  else dl[0] = 1, dl[1] = 0, dl[2] = 0, dl[3] = 0, dl[4] = 1, dl[5] = 0, dl[6] = 0, dl[7] = 0, dl[8] = 1;
  sys_measurement_rows(p, gm, p_FinG, p_FinG, false, hq, st.minfo + (size_t)8 * gm, st.rows + (size_t)gm * p.row_stride);
}

// ---------------------------------------------------------------------------------------------------
// k_feat_qr: one wavefront per feature -> Householder reflectors of H_f (V, tau, T) and z = T^T V^T [H L | r]
// (role of UpdaterHelper.cpp:426-454; the projected rows are Q^T [H L | r] = [H L | r] - V z).
//   V^T [H | r] by sparse gather: a (camera, clone) -> measurement table replaces the scan over all rows of the feature;
//   then (V^T H) L, lane per column, 8 rows of L in flight.
// LDS per wavefront: hf [2 m_max][3], V [2 m_max][3], wv [3][LD], hq [16], mpos [K C].
// ---------------------------------------------------------------------------------------------------
__host__ __device__ inline size_t feat_qr_lds_per_wave(int m_max, int LD, int KC) {
  return (((size_t)12 * m_max + 3 * LD + 64 + 16) * sizeof(double) + (size_t)KC * sizeof(int) + 15) & ~(size_t)15;
}

__global__ void __launch_bounds__(256) k_feat_qr(SysParams p, FeatStore st) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int f = blockIdx.x * 4 + wv;
  if (f >= p.F) return;
  if (p.status[f] != OVGPU_FEAT_USED) return;
  const int D = p.D, LD = p.LD, RS = p.row_stride, KC = p.K * p.C;
  unsigned char *base = smem + (size_t)wv * feat_qr_lds_per_wave(p.m_max, LD, KC);
  double *hf = reinterpret_cast<double *>(base);
  double *V = hf + (size_t)6 * p.m_max;
  double *wvv = V + (size_t)6 * p.m_max;
  double *hq = wvv + (size_t)3 * LD; // [0..2] tau, [3..8] T, [58..60] diag(R1)
  int *mpos = reinterpret_cast<int *>(hq + 64 + 16);
  const int m0 = p.meas_offsets[f], m = p.meas_offsets[f + 1] - m0, n = 2 * m;
  const double *rows = st.rows + (size_t)m0 * RS;
  const int32_t *minfo = st.minfo + (size_t)8 * m0;
  auto wsync = [] {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  };
  for (int r = lane; r < n; r += 64) {
    const double *rd = rows + (size_t)(r >> 1) * RS + RO_HF + 3 * (r & 1);
    hf[3 * r] = rd[0], hf[3 * r + 1] = rd[1], hf[3 * r + 2] = rd[2];
  }
  for (int e = lane; e < KC; e += 64) mpos[e] = -1;
  wsync();
  for (int i = lane; i < m; i += 64) mpos[minfo[8 * i] * p.C + minfo[8 * i + 1]] = i;
  // H_f element (row, k) of the reflector routine = rows[(row >> 1) * RS + RO_HF + 3 (row & 1) + k]: with RS = 6 that is hf[3 row + k]
  sys_hf_householder(hf - RO_HF, 6, V, hq, n, 3, lane);
  wsync();
  for (int r = lane; r < n; r += 64) {
    double *vo = st.V + ((size_t)2 * m0 + r) * 3;
    vo[0] = V[3 * r], vo[1] = V[3 * r + 1], vo[2] = V[3 * r + 2];
  }
  // wv = V^T [H_x | r], lane per column
  for (int c = lane; c < LD; c += 64) {
    double y0 = 0, y1 = 0, y2 = 0;
    if (c == D) {
      for (int i = 0; i < m; i++) {
        const double *v = V + (size_t)6 * i;
        const double *rd = rows + (size_t)i * RS;
        const double r0 = rd[RO_RES], r1 = rd[RO_RES + 1];
        y0 = fma(v[0], r0, y0), y1 = fma(v[1], r0, y1), y2 = fma(v[2], r0, y2);
        y0 = fma(v[3], r1, y0), y1 = fma(v[4], r1, y1), y2 = fma(v[5], r1, y2);
      }
    } else {
      const int kind = p.col_kind[c], var = p.col_var[c], sub = p.col_sub[c];
      const int off = kind == COL_CLONE ? RO_CLONE + sub : (kind == COL_CALIB_POSE ? RO_CPOSE + sub : RO_CINTR + sub);
      const int str = kind == COL_CALIB_INTR ? 8 : 6;
      // a clone column: one measurement per camera at most; a calibration column: the camera's measurements
      const int cnt = kind == COL_CLONE ? p.K : p.C;
      for (int e = 0; e < cnt; e++) {
        const int i = kind == COL_CLONE ? mpos[e * p.C + var] : mpos[var * p.C + e];
        if (i >= 0) {
          const double *v = V + (size_t)6 * i;
          const double *rd = rows + (size_t)i * RS;
          const double h0 = rd[off], h1 = rd[off + str];
          y0 = fma(v[0], h0, y0), y1 = fma(v[1], h0, y1), y2 = fma(v[2], h0, y2);
          y0 = fma(v[3], h1, y0), y1 = fma(v[4], h1, y1), y2 = fma(v[5], h1, y2);
        }
      }
    }
    wvv[c] = y0, wvv[LD + c] = y1, wvv[2 * LD + c] = y2;
  }
  wsync();
  // z = T^T (wv L) = (T^T wv) L: the 3 x 3 factor is applied here, the product with L — a (3 F x D) x (D x D) matrix product over ALL
  // features — is k_feat_z's on the matrix cores (inside this kernel it was a chain of dependent L2 round trips: 86 us of the update's
  // critical path at 2000 features)
  const double T00 = hq[3], T01 = hq[4], T02 = hq[5], T11 = hq[6], T12 = hq[7], T22 = hq[8];
  double *wo = st.w + (size_t)f * 3 * LD;
  for (int c = lane; c < LD; c += 64) {
    const double y0 = wvv[c], y1 = wvv[LD + c], y2 = wvv[2 * LD + c];
    wo[c] = T00 * y0, wo[LD + c] = T01 * y0 + T11 * y1, wo[2 * LD + c] = T02 * y0 + T12 * y1 + T22 * y2;
  }
}

// ---------------------------------------------------------------------------------------------------
// k_feat_z: Z = W L for all features at once, W = [3 F x LD] (k_feat_qr), L = the prior block's lower-triangular factor; one
// wavefront per 16 x 16 tile of Z on v_mfma_f64_16x16x4_f64, the sum over s starting at the tile's first column (L[s][c] = 0 for
// s < c); the residual column D is copied (it is not whitened).  Rows of features that did not reach k_feat_qr hold stale values:
// nobody reads their z.
// ---------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) k_feat_z(int rows, int D, int LD, const double *__restrict__ W, const double *__restrict__ Lw, double *__restrict__ Z) {
  const int lane = threadIdx.x & 63;
  const int tcols = (LD + 15) / 16, trows = (rows + 15) / 16;
  const int tile = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (tile >= trows * tcols) return;
  const int r0 = (tile / tcols) * 16, c0 = (tile % tcols) * 16;
  d4 acc = {0.0, 0.0, 0.0, 0.0};
  const int i = lane & 15, kk = lane >> 4;
  const int ra = min(r0 + i, rows - 1), cb = min(c0 + i, D - 1);
  for (int k0 = c0; k0 < D; k0 += 16) { // 4 k-slices per trip, their 8 operand loads issued together
    double a[4], b[4];
#pragma unroll
    for (int u = 0; u < 4; u++) {
      const int k = k0 + 4 * u + kk;
      const int kc = min(k, D - 1);
      a[u] = k < D ? W[(size_t)(ra / 3) * 3 * LD + (size_t)(ra % 3) * LD + kc] : 0.0;
      b[u] = (k < D && c0 + i < D) ? Lw[(size_t)kc * D + cb] : 0.0;
    }
#pragma unroll
    for (int u = 0; u < 4; u++) FEAT_MFMA(a[u], b[u], acc);
  }
  const int col = c0 + (lane & 15);
#pragma unroll
  for (int q = 0; q < 4; q++) {
    const int row = r0 + (lane >> 4) + 4 * q;
    if (row < rows && col < LD) {
      const size_t o = (size_t)(row / 3) * 3 * LD + (size_t)(row % 3) * LD + col;
      Z[o] = col == D ? W[o] : acc[q];
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// k_feat_out: rows 3.. of Q^T [H L | r] = [H L | r] - V z of every accepted feature -> the stacked system (zero rows for the others).
// One workgroup per feature, thread = column; no LDS: every Jacobian value / reflector entry is a scalar-cache operand, the rows
// of L come from L2 (GY measurements ahead).  L is lower triangular and the calibration columns come first, so a clone block
// contributes nothing to the columns right of it and only the first wavefront's columns see the calibration blocks.
// Runs after the gate (k_feat) AND after the prior block's factorisation (L, z): the gate itself needs neither, which is what lets
// that factorisation run next to it on the second stream.
// ---------------------------------------------------------------------------------------------------
// NC = columns per lane.  NC = 2 (128 threads: columns t and t + 128) halves the number of wavefronts that stream a feature's
// Jacobian records through the scalar cache and the scalar bookkeeping per multiply-add — the two things the ablation of this kernel
// found it to be bound by.
template <int NC>
__global__ void __launch_bounds__(256 / NC) k_feat_out(SysParams p, const double *__restrict__ rowsG, const int32_t *__restrict__ minfoG,
                                                        const double *__restrict__ VG, const double *__restrict__ zG) {
  constexpr int NTH = 256 / NC;
  const int tid = threadIdx.x;
  const int D = p.D, LD = p.LD, RS = p.row_stride;
  int cN[NC], cw0[NC];
  const double *Lc[NC];
#pragma unroll
  for (int e = 0; e < NC; e++) {
    cN[e] = tid + NTH * e;                       // this lane's columns of [H L | r]
    cw0[e] = (tid & ~63) + NTH * e;              // smallest column of the wavefront in segment e: blocks of L above it contribute nothing
    Lc[e] = p.Lw + (cN[e] < D ? cN[e] : D - 1);
  }
  for (int slot = blockIdx.x; slot < p.F; slot += gridDim.x) {
    const int f = __builtin_amdgcn_readfirstlane(p.order ? p.order[slot] : slot);
    const int m0 = __builtin_amdgcn_readfirstlane(p.meas_offsets[f]);
    const int m = __builtin_amdgcn_readfirstlane(p.meas_offsets[f + 1]) - m0;
    const int64_t orow0 = p.row_off[f];
    const int n_out = (int)(p.row_off[f + 1] - orow0);
    if (p.status[f] != OVGPU_FEAT_USED) {
      for (int64_t e = tid; e < (int64_t)n_out * LD; e += NTH) p.Hbig[orow0 * LD + e] = 0.0;
      continue;
    }
    const double *frow = rowsG + (size_t)m0 * RS;
    const int32_t *finfo = minfoG + (size_t)8 * m0;
    const double *Vl = VG + (size_t)6 * m0;
    if (cN[0] < LD) {
      double z[NC][3];
#pragma unroll
      for (int e = 0; e < NC; e++) {
        const double *zf = zG + (size_t)f * 3 * LD + (cN[e] < LD ? cN[e] : 0);
        z[e][0] = zf[0], z[e][1] = zf[LD], z[e][2] = zf[2 * LD];
      }
      double lcp[6] = {0, 0, 0, 0, 0, 0}, lci[8] = {0, 0, 0, 0, 0, 0, 0, 0}; // calibration rows of L: segment 0 only (its columns come first)
      int cam_l = -1;
      double *out = p.Hbig + orow0 * LD;
      constexpr int GY = NC == 1 ? 4 : 2;
#pragma unroll 1
      for (int ib = 0; ib < m; ib += GY) {
        double lcl[GY][NC][6];
#pragma unroll
        for (int ii = 0; ii < GY; ii++) {
          const int i = min(ib + ii, m - 1);
          const int ccol = finfo[8 * i + 2];
#pragma unroll
          for (int e = 0; e < NC; e++) {
            const double *Lr = Lc[e] + (size_t)ccol * D;
            const bool live = ccol + 5 >= cw0[e]; // wave-uniform
#pragma unroll
            for (int s = 0; s < 6; s++) lcl[ii][e][s] = live ? Lr[(size_t)s * D] : 0.0;
          }
        }
#pragma unroll
        for (int ii = 0; ii < GY; ii++) {
          const int i = ib + ii;
          if (i < m) {
            const int32_t *mi = finfo + 8 * i;
            const double *rd = frow + (size_t)i * RS;
            const int camv = mi[0], cc2 = mi[2], cc3 = mi[3], cc4 = mi[4];
            double t0[NC], t1[NC];
#pragma unroll
            for (int e = 0; e < NC; e++) {
              t0[e] = 0.0, t1[e] = 0.0;
              if (cc2 + 5 >= cw0[e]) {
#pragma unroll
                for (int s = 0; s < 6; s++) t0[e] = fma(rd[RO_CLONE + s], lcl[ii][e][s], t0[e]), t1[e] = fma(rd[RO_CLONE + 6 + s], lcl[ii][e][s], t1[e]);
              }
            }
            if ((cc3 >= 0 && cc3 + 5 >= cw0[0]) || (cc4 >= 0 && cc4 + 7 >= cw0[0])) { // first wavefront, first segment only
              if (camv != cam_l) {
                cam_l = camv;
#pragma unroll
                for (int s = 0; s < 6; s++) lcp[s] = cc3 >= 0 ? Lc[0][(size_t)(cc3 + s) * D] : 0.0;
#pragma unroll
                for (int s = 0; s < 8; s++) lci[s] = cc4 >= 0 ? Lc[0][(size_t)(cc4 + s) * D] : 0.0;
              }
              double s0 = 0.0, s1 = 0.0;
#pragma unroll
              for (int s = 0; s < 6; s++) s0 = fma(rd[RO_CPOSE + s], lcp[s], s0), s1 = fma(rd[RO_CPOSE + 6 + s], lcp[s], s1);
#pragma unroll
              for (int s = 0; s < 8; s++) s0 = fma(rd[RO_CINTR + s], lci[s], s0), s1 = fma(rd[RO_CINTR + 8 + s], lci[s], s1);
              t0[0] += s0, t1[0] += s1;
            }
            const double *v = Vl + (size_t)6 * i;
            const int r = 2 * i;
#pragma unroll
            for (int e = 0; e < NC; e++) {
              if (cN[e] == D) t0[e] = rd[RO_RES], t1[e] = rd[RO_RES + 1]; // the residual column is not whitened
              t0[e] -= v[0] * z[e][0] + v[1] * z[e][1] + v[2] * z[e][2], t1[e] -= v[3] * z[e][0] + v[4] * z[e][1] + v[5] * z[e][2];
              if (cN[e] < LD) {
                if (r >= 3) out[(size_t)(r - 3) * LD + cN[e]] = t0[e];
                if (r + 1 >= 3) out[(size_t)(r + 1 - 3) * LD + cN[e]] = t1[e];
              }
            }
          }
        }
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------------
// k_feat: one feature per workgroup — the gate: T = H P, S0 tiles in registers, blocked Cholesky on the matrix cores, chi2.  NW wavefronts per workgroup, TPW gate tiles per wavefront:
// NT (NT + 1) / 2 + NT <= NW * TPW for every feature of the batch.
// The row store arrives as separate restrict-qualified parameters: reads at wave-uniform indices become scalar loads.
// ---------------------------------------------------------------------------------------------------
template <int NW, int TPW, int OCC>
__global__ void __launch_bounds__(64 * NW, OCC)
    k_feat(SysParams p, int nt_max, const double *__restrict__ rowsG, const int32_t *__restrict__ minfoG, const double *__restrict__ VG,
           const double *__restrict__ zG) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr int NCG = TPW <= 11 ? 2 : 1;                        // columns per lane in the T sweep (the wide shape has no registers to spare)
  constexpr int CGT = 256 / NCG, MH = 64 * NW / CGT, CHM = FT_CH / MH; // threads per column group, groups, measurements of a chunk per group
  constexpr int NTH = 64 * NW, GL = CHM < 4 / NCG ? CHM : 4 / NCG; // CHM measurements of a chunk per thread, loaded GL at a time
  static_assert(NW % 4 == 0 && FT_CH % MH == 0 && CHM % GL == 0, "column groups of 256 / NCG threads, each sweeping its share of a chunk's measurements");
  const int tid = threadIdx.x, lane = tid & 63;
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int g = lane >> 4, cl = lane & 15;
  const int colt = tid % CGT, half = __builtin_amdgcn_readfirstlane(tid / CGT);
  const int D = p.D, LD = p.LD, N = p.N, RS = p.row_stride;
  const FeatLds lo = feat_lds_layout(p.m_max, RS, D, LD, p.K * p.C, nt_max);
  int *minfo = reinterpret_cast<int *>(smem + lo.minfo);
  double *rows = reinterpret_cast<double *>(smem + lo.rows);
  double *rhs = reinterpret_cast<double *>(smem + lo.rhs);
  double *Tch = reinterpret_cast<double *>(smem + lo.big);
  double *panel = Tch; // the Cholesky's row panel takes the T chunk's place once the gate matrix is complete
  double *st0 = reinterpret_cast<double *>(smem + lo.stage), *st1 = st0 + 256;
  int *sched = reinterpret_cast<int *>(smem + lo.sched);
  const double sig2 = p.opt.sigma_pix_sq;
  const double *Pc[NCG]; // this thread's columns of P: colt, colt + CGT
#pragma unroll
  for (int e = 0; e < NCG; e++) Pc[e] = p.P + p.col_cov[min(colt + CGT * e, D - 1)];

  long long tlast = 0;
  const bool prof = p.dbg != nullptr && blockIdx.x == 0 && tid == 0;
  if (prof) tlast = clock64();
#define FEAT_T(i)                              \
  if (prof) {                                  \
    const long long tn = clock64();            \
    p.dbg[200 + (i)] += tn - tlast, tlast = tn; \
  }

  for (;;) {
    __syncthreads(); // the previous feature's LDS is fully consumed
    if (tid == 0) sched[0] = atomicAdd(p.work_counter, 1);
    __syncthreads();
    const int slot = __builtin_amdgcn_readfirstlane(sched[0]);
    if (slot >= p.F) break;
    const int f = __builtin_amdgcn_readfirstlane(p.order ? p.order[slot] : slot);
    const int m0 = __builtin_amdgcn_readfirstlane(p.meas_offsets[f]);
    const int m = __builtin_amdgcn_readfirstlane(p.meas_offsets[f + 1]) - m0;
    const int64_t orow0 = p.row_off[f];
    const int n_out = (int)(p.row_off[f + 1] - orow0); // 2m - 3 (0 when m < 2)
    if (p.status[f] != OVGPU_FEAT_USED) continue; // failed before the gate (k_feat_out writes its zero rows)
    const int n = 2 * m, NT = (n + 15) >> 4, NTT = NT * (NT + 1) / 2, ntiles = NTT + NT;
    // this wavefront's tiles: linear index t = s NW + wv over the upper triangle column by column, then the right-hand-side column NT
    int tij[TPW]; // (j << 8) | i, or -1 for an unused slot
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      const int t = s * NW + wv;
      int i = -1, j = 0;
      if (t < NTT) {
        while ((j + 1) * (j + 2) / 2 <= t) j++;
        i = t - j * (j + 1) / 2;
      } else if (t < ntiles) {
        j = NT, i = t - NTT;
      }
      tij[s] = i < 0 ? -1 : ((j << 8) | i);
    }
#define TI(s) (tij[s] & 255)
#define TJ(s) (tij[s] >> 8)
    d4 acc[TPW];
    const double *frow = rowsG + (size_t)m0 * RS;   // this feature's rows in the store (wave-uniform reads -> scalar loads)
    const int32_t *finfo = minfoG + (size_t)8 * m0;

    // ------------------------------------------------------------------ (a) LDS copies for the per-lane reads of the gate's tiles: rows, bookkeeping, [r | H_f]
    if ((RS & 1) == 0) { // 16-byte copies, four in flight per thread
      const double2 *src = reinterpret_cast<const double2 *>(frow);
      double2 *dst = reinterpret_cast<double2 *>(rows);
      const int n2 = (m * RS) >> 1;
#pragma unroll 4
      for (int e = tid; e < n2; e += NTH) dst[e] = src[e];
    } else {
      for (int e = tid; e < m * RS; e += NTH) rows[e] = frow[e];
    }
    {
      const int4 *src = reinterpret_cast<const int4 *>(finfo);
      int4 *dst = reinterpret_cast<int4 *>(minfo);
      for (int e = tid; e < 2 * m; e += NTH) dst[e] = src[e];
    }
    for (int i = tid; i < 8 * NT; i += NTH) {
      double *q0 = rhs + (size_t)8 * i;
      if (i < m) {
        const double *rd = frow + (size_t)i * RS;
        q0[0] = rd[RO_RES], q0[1] = rd[RO_HF], q0[2] = rd[RO_HF + 1], q0[3] = rd[RO_HF + 2];
        q0[4] = rd[RO_RES + 1], q0[5] = rd[RO_HF + 3], q0[6] = rd[RO_HF + 4], q0[7] = rd[RO_HF + 5];
      } else {
#pragma unroll
        for (int e = 0; e < 8; e++) q0[e] = 0.0;
      }
    }
    __syncthreads();
    FEAT_T(0)

    // ------------------------------------------------------------------ (d) T = H P chunk by chunk (thread = column) -> the gate matrix's tiles
    // Every Jacobian value is wave-uniform: read from the row store through the scalar cache it is an SGPR operand of the
    // multiply-add.  The 14 calibration rows of P are loaded once per camera, the 6 clone rows GL measurements ahead.
    // Two columns per lane (NCG): half as many wavefronts stream each record through the scalar cache, and the scalar bookkeeping
    // per multiply-add halves (T sweep 278 -> 238 kcycles per workgroup at 2000 features).
    {
      double pcp[NCG][6], pci[NCG][8];
#pragma unroll
      for (int e = 0; e < NCG; e++) {
#pragma unroll
        for (int s = 0; s < 6; s++) pcp[e][s] = 0.0;
#pragma unroll
        for (int s = 0; s < 8; s++) pci[e][s] = 0.0;
      }
      int cam_p = -1;
      for (int I = 0; I < NT; I++) {
        const int i_first = FT_CH * I;
        if (colt < D) {
#pragma unroll 1
          for (int gq = 0; gq < CHM; gq += GL) {
            double pcl[GL][NCG][6];
#pragma unroll
            for (int ii = 0; ii < GL; ii++) {
              const int i = min(i_first + MH * (gq + ii) + half, m - 1);
              const size_t prow = (size_t)finfo[8 * i + 5] * N;
#pragma unroll
              for (int e = 0; e < NCG; e++)
#pragma unroll
                for (int s = 0; s < 6; s++) pcl[ii][e][s] = Pc[e][prow + (size_t)s * N];
            }
#pragma unroll
            for (int ii = 0; ii < GL; ii++) {
              const int lr = MH * (gq + ii) + half; // measurement of the chunk
              const int i = i_first + lr;
              if (i < m) {
                const int32_t *mi = finfo + 8 * i;
                const double *rd = frow + (size_t)i * RS;
                const int camv = mi[0], cv6 = mi[6], cv7 = mi[7];
                if (camv != cam_p) {
                  cam_p = camv;
#pragma unroll
                  for (int e = 0; e < NCG; e++) {
                    if (cv6 >= 0) {
#pragma unroll
                      for (int s = 0; s < 6; s++) pcp[e][s] = Pc[e][(size_t)(cv6 + s) * N];
                    }
                    if (cv7 >= 0) {
#pragma unroll
                      for (int s = 0; s < 8; s++) pci[e][s] = Pc[e][(size_t)(cv7 + s) * N];
                    }
                  }
                }
#pragma unroll
                for (int e = 0; e < NCG; e++) {
                  double t0 = 0.0, t1 = 0.0, s0 = 0.0, s1 = 0.0;
#pragma unroll
                  for (int s = 0; s < 6; s++) t0 = fma(rd[RO_CLONE + s], pcl[ii][e][s], t0), t1 = fma(rd[RO_CLONE + 6 + s], pcl[ii][e][s], t1);
                  if (cv6 >= 0) {
#pragma unroll
                    for (int s = 0; s < 6; s++) s0 = fma(rd[RO_CPOSE + s], pcp[e][s], s0), s1 = fma(rd[RO_CPOSE + 6 + s], pcp[e][s], s1);
                  }
                  if (cv7 >= 0) {
#pragma unroll
                    for (int s = 0; s < 8; s++) t0 = fma(rd[RO_CINTR + s], pci[e][s], t0), t1 = fma(rd[RO_CINTR + 8 + s], pci[e][s], t1);
                  }
                  const int c = colt + CGT * e;
                  if (c < D) {
                    Tch[(size_t)(2 * lr) * D + c] = t0 + s0;
                    Tch[(size_t)(2 * lr + 1) * D + c] = t1 + s1;
                  }
                }
              }
            }
          }
        }
        __syncthreads();
        FEAT_T(3)
        // ---- tile row I of the gate matrix: S0[a][b] = T[a] . H[b] (+ s^2 on the diagonal), a in the chunk, b >= 16 I
        for (int j = I; j <= NT; j++) { // this wavefront's tiles of the row (one copy of the code; the slot is picked at the end)
          const int t = j < NT ? j * (j + 1) / 2 + I : NTT + I;
          if (t % NW != wv) continue;
          d4 av = {0.0, 0.0, 0.0, 0.0};
          if (j == NT) { // right-hand sides: columns [r | H_f], the rest of the tile is zero
#pragma unroll
            for (int q = 0; q < 4; q++) av[q] = cl < 4 ? rhs[(size_t)(16 * I + g + 4 * q) * 4 + cl] : 0.0;
          } else {
            const int b = 16 * j + cl;
            const int bq = min(b, n - 1);
            const int *mi = minfo + 8 * (bq >> 1);
            const double *rd = rows + (size_t)(bq >> 1) * RS;
            const int pa = bq & 1, c_cl = mi[2], c_po = mi[3], c_in = mi[4];
            double hcl[6], hpo[6], hin[8];
#pragma unroll
            for (int k = 0; k < 6; k++) hcl[k] = rd[RO_CLONE + 6 * pa + k];
#pragma unroll
            for (int k = 0; k < 6; k++) hpo[k] = c_po >= 0 ? rd[RO_CPOSE + 6 * pa + k] : 0.0;
#pragma unroll
            for (int k = 0; k < 8; k++) hin[k] = c_in >= 0 ? rd[RO_CINTR + 8 * pa + k] : 0.0;
            const int o_po = c_po >= 0 ? c_po : 0, o_in = c_in >= 0 ? c_in : 0;
#pragma unroll
            for (int q = 0; q < 4; q++) {
              const int lr = g + 4 * q, a = 16 * I + lr;
              const double *Tr = Tch + (size_t)lr * D;
              double e0 = 0.0, e1 = 0.0, e2 = 0.0;
#pragma unroll
              for (int k = 0; k < 6; k++) e0 = fma(Tr[c_cl + k], hcl[k], e0);
#pragma unroll
              for (int k = 0; k < 6; k++) e1 = fma(Tr[o_po + k], hpo[k], e1);
#pragma unroll
              for (int k = 0; k < 8; k++) e2 = fma(Tr[o_in + k], hin[k], e2);
              double sv = (e0 + e1) + e2 + (a == b ? sig2 : 0.0);
              if (a >= n || b >= n) sv = (a == b) ? 1.0 : 0.0; // padding of the last tile row / column: identity
              av[q] = sv;
            }
          }
          const int slot_t = t / NW;
#pragma unroll
          for (int s = 0; s < TPW; s++)
            if (s == slot_t) acc[s] = av;
        }
        __syncthreads(); // the T chunk is free again
        FEAT_T(4)
      }
    }

    // ------------------------------------------------------------------ (e) blocked Cholesky S0 = U^T U on the matrix cores, right-hand sides carried
    for (int k = 0; k < NT; k++) {
      // (1) the owner of the diagonal tile factors it and publishes U_kk^-1
      {
        const int tkk = k * (k + 1) / 2 + k;
        if (tkk % NW == wv) {
          const int slot_t = tkk / NW;
          d4 av = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
          for (int s = 0; s < TPW; s++)
            if (s == slot_t) av = acc[s];
          d4 ev;
          (void)diag_tile_factor_blk(av, ev, st0, lane, nullptr, 0.0, 16);
#pragma unroll
          for (int q = 0; q < 4; q++) st1[cl * 16 + g + 4 * q] = ev[q]; // U^-T in accumulator layout -> U^-1 row-major
        }
      }
      __syncthreads();
      FEAT_T(5)
      // (2) row panel: W_kj = U_kk^-T S_kj, published for the trailing update
      {
        double ua[4];
#pragma unroll
        for (int u = 0; u < 4; u++) ua[u] = st1[(4 * u + g) * 16 + cl]; // A[i][k'] = U^-1[k'][i]
#pragma unroll
        for (int s = 0; s < TPW; s++) {
          if (tij[s] >= 0 && TI(s) == k && TJ(s) > k) {
            d4 w = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int u = 0; u < 4; u++) FEAT_MFMA(ua[u], acc[s][u], w);
            acc[s] = w;
            double *pt = panel + (size_t)TJ(s) * 256;
#pragma unroll
            for (int q = 0; q < 4; q++) pt[(g + 4 * q) * 16 + cl] = w[q];
          }
        }
      }
      __syncthreads();
      FEAT_T(6)
      // (3) trailing update S_ij -= W_ki^T W_kj, k < i <= j (j = NT: the right-hand sides)
#pragma unroll
      for (int s = 0; s < TPW; s++) {
        if (tij[s] >= 0 && TI(s) > k) {
          const double *pi = panel + (size_t)TI(s) * 256, *pj = panel + (size_t)TJ(s) * 256;
          double a[4], b[4];
#pragma unroll
          for (int u = 0; u < 4; u++) a[u] = -pi[(4 * u + g) * 16 + cl], b[u] = pj[(4 * u + g) * 16 + cl];
#pragma unroll
          for (int u = 0; u < 4; u++) FEAT_MFMA(a[u], b[u], acc[s]);
        }
      }
      FEAT_T(7)
      // no barrier here: the next step's factorisation touches st0 / st1 only, and its panel writes come after its first barrier
    }
    // ------------------------------------------------------------------ (f) chi2 = |y_r|^2 - g^T G^-1 g,  y_r = U^-T r, Y_f = U^-T H_f
#pragma unroll
    for (int s = 0; s < TPW; s++) {
      if (tij[s] >= 0 && TJ(s) == NT && cl < 4) {
#pragma unroll
        for (int q = 0; q < 4; q++) rhs[(size_t)(16 * TI(s) + g + 4 * q) * 4 + cl] = acc[s][q];
      }
    }
    __syncthreads();
    if (wv == 0) {
      double a = 0, G00 = 0, G01 = 0, G02 = 0, G11 = 0, G12 = 0, G22 = 0, g0 = 0, g1 = 0, g2 = 0;
      for (int j = lane; j < n; j += 64) {
        const double yr = rhs[4 * j], y0 = rhs[4 * j + 1], y1 = rhs[4 * j + 2], y2 = rhs[4 * j + 3];
        a = fma(yr, yr, a);
        G00 = fma(y0, y0, G00), G01 = fma(y0, y1, G01), G02 = fma(y0, y2, G02);
        G11 = fma(y1, y1, G11), G12 = fma(y1, y2, G12), G22 = fma(y2, y2, G22);
        g0 = fma(y0, yr, g0), g1 = fma(y1, yr, g1), g2 = fma(y2, yr, g2);
      }
      a = wave_sum(a);
      G00 = wave_sum(G00), G01 = wave_sum(G01), G02 = wave_sum(G02), G11 = wave_sum(G11), G12 = wave_sum(G12), G22 = wave_sum(G22);
      g0 = wave_sum(g0), g1 = wave_sum(g1), g2 = wave_sum(g2);
      const M3 Gm{G00, G01, G02, G01, G11, G12, G02, G12, G22};
      const V3 gv{g0, g1, g2};
      const V3 x = colpiv_qr_solve3(Gm, gv);
      const double chi2 = a - dot(gv, x);
      const double thr = p.opt.chi2_multipler * p.chi2_table[min(n - 3, p.chi2_table_len - 1)]; // UpdaterMSCKF.cpp:216-222
      if (lane == 0) {
        p.chi2[f] = chi2;
        p.chi2_thresh[f] = thr;
        const bool reject = chi2 > thr; // :225
        sched[1] = reject ? 1 : 0;
        if (reject) p.status[f] = OVGPU_FEAT_CHI2_REJECTED;
        else if (p.rows_used) atomicAdd(p.rows_used, n_out);
      }
    }
    __syncthreads();
    FEAT_T(8)
    // (the rows of the stacked system are written by k_feat_out, which reads the status)

    FEAT_T(10)
  }
#undef FEAT_T
#undef TI
#undef TJ
}

} // namespace feat
} // namespace ovg
