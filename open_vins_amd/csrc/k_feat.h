// k_feat.h — what the kernels of the MSCKF fast path share (k_featy.h: the fused per-feature kernel; k_featy_big.h: its block-row
// form): the accumulator-register tile type, the diagonal-tile factorisations of the gate's blocked Cholesky, the row store.
//
//   chi2 gate                                           UpdaterMSCKF.cpp:209-234, StateHelper.cpp:226-254
//
// The gate matrix never touches LDS.  Its 16 x 16 tiles (upper triangle + one tile column of right-hand sides) are dealt round-robin
// to the wavefronts and stay in REGISTERS in the accumulator layout of v_mfma_f64_16x16x4_f64 (lane (g, c) holds rows g, g+4, g+8,
// g+12 of column c).  That layout is at once the B-operand layout of a tile and the A-operand layout of its transpose, so the blocked
// Cholesky S0 = U^T U runs on the matrix cores without a single register shuffle:
//        W_kj = U_kk^-T S_kj        A = U_kk^-1 read from LDS in accumulator order, B = the tile's own registers
//        S_ij -= W_ki^T W_kj        A, B = panel tiles read from LDS in accumulator order
// and U_kk^-1 comes out of the diagonal tile's factorisation for free: 16 extra lanes carry the identity through the same instruction
// stream, which turns it into U_kk^-T.
//
// Round 2's three-sweep kernels that lived here (k_feat_rows, k_feat_qr, k_feat_z, k_feat, k_feat_out: T = H P in the gate, Y = H L twice)
// were retired in round 4: since round 3 no default path reached them (the fused kernel holds every shape they held), and every
// kernel that stays switchable is a parity surface to keep green.  Their measurements stay in DESIGN.md section 4.
#pragma once
#include "k_system.h"

namespace ovg {
namespace feat {

typedef double d4 __attribute__((ext_vector_type(4)));

constexpr int FT_CH = 8; // measurements per chunk = 16 rows = one tile row of the gate matrix

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
  // The pivot thresholds, ALL sixteen before the chain starts (round 6): as `bad || !(dk > tol * diag0[k])` inside it every pivot had a branch
  // and — in the prior's factorisation, which has a diag0 — a dependent load with its own s_waitcnt in front of the next v_rsq_f64 (ISA of round
  // 5's k_chol_fused: sixteen flat_load / s_waitcnt vmcnt(0) pairs per diagonal tile on the one path nothing hides; -5 us per update)
  double thr[16];
#pragma unroll
  for (int k = 0; k < 16; k++) thr[k] = diag0 ? diag0[k < nb ? k : 0] : 0.0;
#pragma unroll
  for (int k = 0; k < 16; k++) thr[k] = diag0 ? tol * thr[k] : 0.0;
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
      bad = bad | ((kb + r < nb) & !(dk > thr[kb + r]));
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

// Round 6: the diagonal tile WITHOUT square roots and with a chain a third as long — block L D L^T by 4 x 4 blocks,
//        S = L D L^T,   L unit lower triangular by blocks, D = diag(A_0 .. A_3), A_b the 4 x 4 Schur complements,
// which is all the gate needs: the trailing update of the blocked elimination is S_ij -= S_ki^T S_kk^-1 S_kj = Y_ki^T X_kj with
// Y_kj = L^-1 S_kj, X_kj = D^-1 Y_kj, and the statistic r^T S^-1 r is an entry of the Schur complement of the augmented matrix.
//   s   in: the tile (accumulator layout); out: rows of the blocks >= nblk hold their Schur complement with respect to the eliminated
//       blocks (columns >= 4 nblk are meaningful), the other rows are scratch
//   e   out: E = L^-1 (rows of the blocks >= nblk: rows of the identity carried through the elimination)
//   f   out: F = D^-1 L^-1 (rows of the blocks >= nblk: zero)
//   sh  128 doubles of LDS scratch private to the wavefront;  nblk (wave-uniform): the leading 4 x 4 blocks to eliminate
// Block step b: the four rows of [S | E] go through LDS.  Lane (g, c) needs ROW g of A_b^-1 only (its part of the A operand of the
// rank-4 update, (A_b^-1 Z_b)[g][c]), so it reads the block PERMUTED with its own row last — (g+1, g+2, g+3, g) mod 4, a symmetric
// permutation of a positive definite block — and evaluates the last row of the inverse of [P Q; Q^T R] (2 x 2 blocks) with every
// division deferred:  T' = adj(P) Q,  S' = det(P) R - Q^T T',  last row = [ -T' adj(S')[:, 1] ; det(P) adj(S')[1, :] ] / det(S').
// ONE reciprocal per block step and nothing but its Newton correction behind it on the chain (det P -> S' -> det S' -> 1 / det S' ->
// one product -> the matrix instruction): ~13 dependent operations per four pivots where the Cholesky form above has ~36 (four
// reciprocal square roots with two Newton steps each), no lane selects, ~45 instead of ~100 instructions per block step.  The four
// lane groups invert four differently ordered copies of the block: the rows agree to rounding (pivoting order of a positive definite
// 4 x 4), which is what the elimination's other roundings are.  Lanes of the columns c < 4 b + 4 compute on finished columns: their
// products land in rows / columns nothing reads again.  Measured against the Cholesky form: profiles/r06_c_diagonal_tile_ldl.txt.
__device__ __forceinline__ void diag_tile_ldl_blk(d4 &s, d4 &e, d4 &f, double *sh, int lane, int nblk) {
  // (the lane id from the hardware, inside an opaque statement: the ten per-lane offsets below depend on nothing else, were hoisted out of the
  //  feature loop of k_feat_y<8, 17, 1> and spilled — thirteen scratch loads per block step on the one path nothing hides)
  int lane_o;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane_o));
  (void)lane;
  const int g = lane_o >> 4, cl = lane_o & 15;
  const int p0 = (g + 1) & 3, p1 = (g + 2) & 3, p2 = (g + 3) & 3, p3 = g;
  // (both triangles of a diagonal tile are kept — SYRK, trailing and in-tile updates write whole tiles — so the block is read where the
  //  permutation points, equal to its mirror image up to rounding)
  const int o00 = p0 * 16 + p0, o01 = p0 * 16 + p1, o11 = p1 * 16 + p1, o02 = p0 * 16 + p2, o03 = p0 * 16 + p3, o12 = p1 * 16 + p2, o13 = p1 * 16 + p3,
            o22 = p2 * 16 + p2, o23 = p2 * 16 + p3, o33 = p3 * 16 + p3;
#pragma unroll
  for (int q = 0; q < 4; q++) e[q] = (g + 4 * q == cl) ? 1.0 : 0.0, f[q] = 0.0;
#pragma unroll
  for (int b = 0; b < 4; b++) {
    if (b < nblk) {
      const int kb = 4 * b;
      sh[g * 16 + cl] = s[b], sh[64 + g * 16 + cl] = e[b];
      __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
      __builtin_amdgcn_wave_barrier();
      __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
      const double *ab = sh + kb;
      const double a00 = ab[o00], a01 = ab[o01], a11 = ab[o11];                   // P
      const double q00 = ab[o02], q01 = ab[o03], q10 = ab[o12], q11 = ab[o13];    // Q
      const double r00 = ab[o22], r01 = ab[o23], r11 = ab[o33];                   // R
      const double z0 = sh[p0 * 16 + cl], z1 = sh[p1 * 16 + cl], z2 = sh[p2 * 16 + cl], z3 = sh[p3 * 16 + cl];
      const double e0 = sh[64 + p0 * 16 + cl], e1 = sh[64 + p1 * 16 + cl], e2 = sh[64 + p2 * 16 + cl], e3 = sh[64 + p3 * 16 + cl];
      const double detp = fma(a00, a11, -(a01 * a01));
      const double t00 = fma(a11, q00, -(a01 * q10)), t01 = fma(a11, q01, -(a01 * q11)); // T' = adj(P) Q
      const double t10 = fma(a00, q10, -(a01 * q00)), t11 = fma(a00, q11, -(a01 * q01));
      const double s00 = fma(detp, r00, -fma(q00, t00, q10 * t10));                       // S' = det(P) R - Q^T T'
      const double s01 = fma(detp, r01, -fma(q00, t01, q10 * t11));
      const double s11 = fma(detp, r11, -fma(q01, t01, q11 * t11));
      const double dets = fma(s00, s11, -(s01 * s01));
      // numerators of the last row of the inverse: [ -m0, -m1, -det(P) s01, det(P) s00 ],  m = T' adj(S')[:, 1]
      const double m0 = fma(t01, s00, -(t00 * s01)), m1 = fma(t11, s00, -(t10 * s01));
      const double n2 = -(detp * s01), n3 = detp * s00;
      const double nz = fma(n3, z3, fma(n2, z2, -fma(m0, z0, m1 * z1)));
      const double ne = fma(n3, e3, fma(n2, e2, -fma(m0, e0, m1 * e1)));
      double rho = __builtin_amdgcn_rcp(dets); // ~2^-23; (1 + eps + eps^2) behind it leaves eps^3
      const double eps = fma(-dets, rho, 1.0);
      rho = fma(rho, fma(eps, eps, eps), rho);
      const double zi = nz * rho;
      f[b] = ne * rho;
      if (b < 3) { // rows below the block: [S | E] -= (A_b^-1 Z_b)^T [Z_b | E_b]  (finished rows of s take the update as well: scratch from here on)
        FEAT_MFMA(-zi, s[b], s);
        d4 ue = {0.0, 0.0, 0.0, 0.0};
        FEAT_MFMA(-zi, e[b], ue);
#pragma unroll
        for (int q = 0; q < 4; q++)
          if (q > b) e[q] += ue[q];
      }
      __builtin_amdgcn_wave_barrier(); // sh is rewritten by the next block step
    }
  }
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
  const int32_t *pos;       // [M] where each measurement's record goes: clone-major order inside its feature (computed with the batch's layout, api_state.inc)
};


} // namespace feat
} // namespace ovg
