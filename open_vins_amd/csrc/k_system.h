// k_system.h — per-feature linear system: Jacobian, chi2 gate, nullspace projection, stacking.
//
//   UpdaterHelper::get_feature_jacobian_full            UpdaterHelper.cpp:192-424
//   UpdaterHelper::get_feature_jacobian_representation  UpdaterHelper.cpp:32-190
//   UpdaterHelper::nullspace_project_inplace            UpdaterHelper.cpp:426-454
//   chi2 gate                                           UpdaterMSCKF.cpp:209-234, StateHelper.cpp:226-254
//   stacking into Hx_big / res_big                      UpdaterMSCKF.cpp:237-255
//
// One workgroup (256 threads) per feature, persistent over the feature list.
//
// Design notes (MI355X-first, not a translation of the Eigen code):
//  * Before the nullspace projection every Jacobian row is block-sparse: 6 clone columns,
//    6 extrinsic + 8 intrinsic columns of its camera (+ anchor blocks for anchored
//    representations).  Rows are kept in that 46-double sparse form in LDS; the dense
//    2m x d_f matrix of the reference is never materialised on chip.
//  * chi2 gate: with N an orthonormal basis of the left nullspace of H_f,
//        r'^T (N^T S0 N)^-1 r'  =  r^T W r - (H_f^T W r)^T (H_f^T W H_f)^-1 (H_f^T W r),
//    W = S0^-1, S0 = H P H^T + sigma^2 I on the UNPROJECTED rows.  So one Cholesky of the
//    2m x 2m matrix S0 with four extra right-hand sides [r | H_f] gives the statistic of
//    UpdaterMSCKF.cpp:212 without forming N^T H.  S0 is built from the sparse rows in
//    16-row chunks of T = H P (P stays L2-resident) and lives packed-lower in LDS.
//  * nullspace projection: 3 Householder reflectors of H_f in compact-WY form
//    Q^T = I - V T^T V^T.  Each thread owns one output column c of the canonical stacked
//    Jacobian, computes V^T h_c from the few non-zeros of h_c and streams rows 3..2m-1 of
//    Q^T h_c straight to HBM (consecutive threads -> consecutive addresses).  Any orthonormal
//    basis of the nullspace gives the same chi2 and the same compressed information as the
//    reference's Givens sequence.
#pragma once
#include "device_math.h"
#include "ovgpu_types.h"

namespace ovg {

static constexpr int SYS_NT = 256;  // threads per workgroup
static constexpr int SYS_RCM = 8;   // measurements per T-chunk (16 rows)

// LDS carve sizes in bytes for a batch whose longest track is m_max (host + device agree on this)
// rows_in_lds = false: the Jacobian records of the feature live in a per-workgroup global workspace (SysParams::rows_ws, k_system_t<true>):
// tracks whose records do not fit LDS next to the T chunk (~250 observations at 440 columns)
__host__ __device__ inline size_t sys_lds_fixed_bytes(int m_max, int row_stride, int D, bool rows_in_lds = true) {
  size_t b = 0;
  b += ((size_t)m_max * 8 * sizeof(int) + 15) & ~(size_t)15; // minfo
  if (rows_in_lds) b += (size_t)m_max * row_stride * sizeof(double); // rows
  b += (size_t)2 * m_max * 3 * sizeof(double);      // V
  b += 64 * sizeof(double);                         // tau, T, representation scratch
  b += (size_t)2 * SYS_RCM * D * sizeof(double);    // T chunk
  return (b + 15) & ~(size_t)15;
}
__host__ __device__ inline size_t sys_gate_doubles(int m) {
  const size_t n = 2 * (size_t)m;
  return n * (n + 1) / 2 + 4 * n;
}

// packed index of the gate matrix: rows 0..n-1 lower-triangular, rows n..n+3 dense (the 4 RHS)
__device__ __forceinline__ size_t sidx(int i, int j, int n) {
  return (i < n) ? ((size_t)i * (i + 1) / 2 + j) : ((size_t)n * (n + 1) / 2 + (size_t)(i - n) * n + j);
}

// offsets inside one measurement's row store
enum { RO_HF = 0, RO_CLONE = 6, RO_CPOSE = 18, RO_CINTR = 30, RO_RES = 46, RO_ANC = 48, RO_ACAL = 60 };

__device__ __forceinline__ bool rep_is_relative(int rep) { return rep >= OVGPU_REP_ANCHORED_3D; }

// UpdaterHelper.cpp:44-67 / :130-152 — d p / d (theta, phi, rho)
__device__ __forceinline__ void inv_depth_jac(const V3 &p, double *J) {
  const double rho = 1.0 / norm(p);
  const double phi = acos(rho * p.z);
  const double theta = atan2(p.y, p.x);
  const double sin_th = sin(theta), cos_th = cos(theta), sin_phi = sin(phi), cos_phi = cos(phi);
  J[0] = -(1.0 / rho) * sin_th * sin_phi, J[1] = (1.0 / rho) * cos_th * cos_phi, J[2] = -(1.0 / (rho * rho)) * cos_th * sin_phi;
  J[3] = (1.0 / rho) * cos_th * sin_phi, J[4] = (1.0 / rho) * sin_th * cos_phi, J[5] = -(1.0 / (rho * rho)) * sin_th * sin_phi;
  J[6] = 0.0, J[7] = -(1.0 / rho) * sin_phi, J[8] = -(1.0 / (rho * rho)) * cos_phi;
}

// UpdaterHelper::get_feature_jacobian_representation, anchored representations (UpdaterHelper.cpp:84-189):
//   dl [3x3] = d p_FinG / d lambda,  Ha [3x6] = d p_FinG / d (anchor clone pose),  Hc [3x6] = d p_FinG / d (anchor camera extrinsics)
// tab_cam_a = (R_ItoC, p_IinC) of the anchor camera, tab_clone_a = (R_GtoI, p_IinG, R_GtoI_fej, p_IinG_fej) of the anchor clone.
__device__ __forceinline__ void anchored_rep_jacobian(int rep, int do_fej, const double *tab_cam_a, const double *tab_clone_a, const V3 &p_FinA_in, double *dl,
                                                      double *Ha, double *Hc) {
  const M3 R_ItoC = load_m3(tab_cam_a);
  const V3 p_IinC = load_v3(tab_cam_a + 9);
  M3 R_GtoI = load_m3(tab_clone_a);
  V3 p_IinG = load_v3(tab_clone_a + 9);
  V3 p_FinA = p_FinA_in;
  if (do_fej) { // :93-100
    const V3 best = mulT(R_GtoI, mulT(R_ItoC, p_FinA_in - p_IinC)) + p_IinG;
    R_GtoI = load_m3(tab_clone_a + 12);
    p_IinG = load_v3(tab_clone_a + 21);
    p_FinA = mul(R_ItoC, mul(R_GtoI, best - p_IinG)) + p_IinC;
  }
  const M3 R_CtoG = mul(transpose(R_GtoI), transpose(R_ItoC)); // :101
  {
    const M3 blk = mul(transpose(R_GtoI), skew_x(mulT(R_ItoC, p_FinA - p_IinC))); // :105
    Ha[0] = -blk.a00, Ha[1] = -blk.a01, Ha[2] = -blk.a02, Ha[3] = 1, Ha[4] = 0, Ha[5] = 0;
    Ha[6] = -blk.a10, Ha[7] = -blk.a11, Ha[8] = -blk.a12, Ha[9] = 0, Ha[10] = 1, Ha[11] = 0;
    Ha[12] = -blk.a20, Ha[13] = -blk.a21, Ha[14] = -blk.a22, Ha[15] = 0, Ha[16] = 0, Ha[17] = 1;
  }
  {
    const M3 blk = mul(R_CtoG, skew_x(p_FinA - p_IinC)); // :115-116
    Hc[0] = -blk.a00, Hc[1] = -blk.a01, Hc[2] = -blk.a02, Hc[3] = -R_CtoG.a00, Hc[4] = -R_CtoG.a01, Hc[5] = -R_CtoG.a02;
    Hc[6] = -blk.a10, Hc[7] = -blk.a11, Hc[8] = -blk.a12, Hc[9] = -R_CtoG.a10, Hc[10] = -R_CtoG.a11, Hc[11] = -R_CtoG.a12;
    Hc[12] = -blk.a20, Hc[13] = -blk.a21, Hc[14] = -blk.a22, Hc[15] = -R_CtoG.a20, Hc[16] = -R_CtoG.a21, Hc[17] = -R_CtoG.a22;
  }
  M3 d{1, 0, 0, 0, 1, 0, 0, 0, 1};
  if (rep == OVGPU_REP_ANCHORED_FULL_INVERSE_DEPTH) {
    double J[9];
    inv_depth_jac(p_FinA, J);
    d = M3{J[0], J[1], J[2], J[3], J[4], J[5], J[6], J[7], J[8]};
  } else if (rep == OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH) { // :157-175
    const double alpha = p_FinA.x / p_FinA.z, beta = p_FinA.y / p_FinA.z, rho = 1.0 / p_FinA.z;
    d = M3{1.0 / rho, 0.0, -(1.0 / (rho * rho)) * alpha, 0.0, 1.0 / rho, -(1.0 / (rho * rho)) * beta, 0.0, 0.0, -(1.0 / (rho * rho))};
  }
  store_m3(dl, mul(R_CtoG, d));
}

__device__ __forceinline__ double lane_bcast_d(double v, int lane) { // lane is wave-uniform
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
  return __hiloint2double(hi, lo);
}

// One 8-column panel of the gate Cholesky, factored by a single wavefront in registers: row r of the panel
// (rows kb .. n+3) lives in lane r & 63, slot r >> 6; pivots and multipliers are broadcast with v_readlane, so
// there is no LDS round trip and no workgroup barrier inside the panel.  CS = row slots per lane.
// LONG: the trapezoid has more rows than CS x 64 (a track of more than 254 observations): the pivots' reciprocals and the
// multipliers L[kb + jj][kb + k] of the panel are kept (wave-uniform) and the rows behind the first CS x 64 take the same column
// operations, 512 at a time (gate_chol_panel_rest) — the statistic of a track of ANY length the row store holds, UpdaterMSCKF.cpp:216-222
// (dof >= 500: the quantile computed on the fly there, the extended table here).
template <int CS, bool LONG = false>
__device__ __forceinline__ void gate_chol_panel(double *S, int n, int kb, int nb, int lane, double *pinv = nullptr, double *pmul = nullptr) {
  constexpr int CB = 8;
  double a[CS][CB];
  size_t base[CS];
  bool okr[CS];
#pragma unroll
  for (int sl = 0; sl < CS; sl++) {
    const int row = kb + lane + 64 * sl;
    okr[sl] = row < n + 4;
    base[sl] = (row < n ? (size_t)row * (row + 1) / 2 : (size_t)n * (n + 1) / 2 + (size_t)(row - n) * n) + kb; // &S[row][kb]
#pragma unroll
    for (int jj = 0; jj < CB; jj++) a[sl][jj] = (okr[sl] && jj < nb && (row >= n || kb + jj <= row)) ? S[base[sl] + jj] : 0.0;
  }
#pragma unroll
  for (int k = 0; k < CB; k++) {
    if (k < nb) {
      const double dkk = lane_bcast_d(a[0][k], k); // row kb + k lives in lane k, slot 0
      // pivot: d = sqrt(dkk), inv = 1 / d from one v_rsq_f64 seed + Newton steps (a few ulp; the chain of a column step)
      double inv = __builtin_amdgcn_rsq(dkk);
#pragma unroll
      for (int it = 0; it < 2; it++) inv = fma(0.5 * inv, fma(-dkk * inv, inv, 1.0), inv);
      double d = dkk * inv;
      d = fma(0.5 * inv, fma(-d, d, dkk), d);
      if (!(dkk > 0.0)) d = sqrt(dkk), inv = 1.0 / d; // keep the NaN / inf behaviour of a broken-down factorisation
      if (LONG) pinv[k] = inv;
#pragma unroll
      for (int sl = 0; sl < CS; sl++) a[sl][k] = (lane + 64 * sl == k) ? d : a[sl][k] * inv;
#pragma unroll
      for (int jj = k + 1; jj < CB; jj++) {
        if (jj < nb) {
          const double ljk = lane_bcast_d(a[0][k], jj); // L[kb + jj][kb + k]
          if (LONG) pmul[k * CB + jj] = ljk;
#pragma unroll
          for (int sl = 0; sl < CS; sl++) a[sl][jj] = fma(-a[sl][k], ljk, a[sl][jj]);
        }
      }
    }
  }
#pragma unroll
  for (int sl = 0; sl < CS; sl++) {
    const int row = kb + lane + 64 * sl;
#pragma unroll
    for (int jj = 0; jj < CB; jj++)
      if (okr[sl] && jj < nb && (row >= n || kb + jj <= row)) S[base[sl] + jj] = a[sl][jj];
  }
}

// the rows row0 .. row0 + 64 CS - 1 of the panel at kb (all of them below the pivot rows): the column operations of gate_chol_panel
// with its recorded reciprocals pinv[k] and multipliers pmul[k * 8 + jj]
template <int CS>
__device__ __forceinline__ void gate_chol_panel_rest(double *S, int n, int kb, int nb, int lane, int row0, const double *pinv, const double *pmul) {
  constexpr int CB = 8;
#pragma unroll 1
  for (int sl = 0; sl < CS; sl++) {
    const int row = row0 + lane + 64 * sl;
    if (row >= n + 4) continue;
    const size_t base = (row < n ? (size_t)row * (row + 1) / 2 : (size_t)n * (n + 1) / 2 + (size_t)(row - n) * n) + kb; // &S[row][kb]
    double a[CB];
#pragma unroll
    for (int jj = 0; jj < CB; jj++) a[jj] = jj < nb ? S[base + jj] : 0.0; // (row > kb + 7: the whole panel is left of the diagonal)
#pragma unroll
    for (int k = 0; k < CB; k++) {
      if (k < nb) {
        a[k] *= pinv[k];
#pragma unroll
        for (int jj = k + 1; jj < CB; jj++)
          if (jj < nb) a[jj] = fma(-a[k], pmul[k * CB + jj], a[jj]);
      }
    }
#pragma unroll
    for (int jj = 0; jj < CB; jj++)
      if (jj < nb) S[base + jj] = a[jj];
  }
}

// One measurement's two sparse Jacobian rows (UpdaterHelper.cpp:314-421) -> rd, its column / covariance bookkeeping -> mi[8]:
// camera, clone, first column of the clone / extrinsic / intrinsic blocks (-1: not estimated), their covariance ids.
// gm = index of the measurement in the batch; hq = the feature's representation scratch ([12..20] dpfg_dlambda, [21..38] H_anc,
// [39..56] H_calib).  Shared by the general per-feature kernel (k_system) and the MSCKF fast path (k_feat.h).
__device__ __forceinline__ void sys_measurement_rows(const SysParams &p, int gm, const V3 &p_FinG, const V3 &p_FinG_fej, bool relative, const double *hq,
                                                     int *mi, double *rd) {
    const int code = p.meas_cc[gm];
    const int cam = code >> 10, cl = code & 1023;
    const int ccol = p.clone_col[cl], pcol = p.calib_col[cam], icol = p.intr_col[cam];
    mi[0] = cam, mi[1] = cl, mi[2] = ccol, mi[3] = pcol, mi[4] = icol;
    mi[5] = p.col_cov[ccol];
    mi[6] = pcol >= 0 ? p.col_cov[pcol] : -1;
    mi[7] = icol >= 0 ? p.col_cov[icol] : -1;

    const M3 R_ItoC = load_m3(p.tab_cam + 12 * cam);
    const V3 p_IinC = load_v3(p.tab_cam + 12 * cam + 9);
    const CamIntr ci = load_cam(p.intr + 8 * cam);
    const bool fish = p.fisheye[cam] != 0;
    const double *tc = p.tab_clone + 24 * cl;
    M3 R_GtoIi = load_m3(tc);
    V3 p_IiinG = load_v3(tc + 9);
    V3 p_FinIi = mul(R_GtoIi, p_FinG - p_IiinG);  // :334
    V3 p_FinCi = mul(R_ItoC, p_FinIi) + p_IinC;   // :337
    const double un = p_FinCi.x / p_FinCi.z, vn = p_FinCi.y / p_FinCi.z;
    double ud, vd;
    if (fish)
      equi_distort_d(ci, un, vn, ud, vd); // :343 (float round trip, Q2)
    else
      radtan_distort_d(ci, un, vn, ud, vd);
    rd[RO_RES] = (double)p.uv[2 * gm] - ud; // :346-348
    rd[RO_RES + 1] = (double)p.uv[2 * gm + 1] - vd;
    if (p.opt.do_fej) { // :354-363  (uv_norm is deliberately NOT recomputed, Q4)
      R_GtoIi = load_m3(tc + 12);
      p_IiinG = load_v3(tc + 21);
      p_FinIi = mul(R_GtoIi, p_FinG_fej - p_IiinG); // p_FinG_fej == p_FinG for MSCKF features (Q5), the landmark's fej for SLAM
      p_FinCi = mul(R_ItoC, p_FinIi) + p_IinC;
    }
    double dzn[4], dze[16];
    if (fish)
      equi_jacobian(ci, un, vn, dzn, dze); // :367
    else
      radtan_jacobian(ci, un, vn, dzn, dze);
    const double iz = 1.0 / p_FinCi.z;
    const double n02 = -p_FinCi.x * iz * iz, n12 = -p_FinCi.y * iz * iz; // :370-371
    // dz_dpfc = dz_dzn * dzn_dpfc (2x3)
    const double a00 = dzn[0] * iz, a01 = dzn[1] * iz, a02 = dzn[0] * n02 + dzn[1] * n12;
    const double a10 = dzn[2] * iz, a11 = dzn[3] * iz, a12 = dzn[2] * n02 + dzn[3] * n12;
    const M3 dpfc_dpfg = mul(R_ItoC, R_GtoIi); // :374
    // dz_dpfg = dz_dpfc * dpfc_dpfg
    const double g00 = a00 * dpfc_dpfg.a00 + a01 * dpfc_dpfg.a10 + a02 * dpfc_dpfg.a20;
    const double g01 = a00 * dpfc_dpfg.a01 + a01 * dpfc_dpfg.a11 + a02 * dpfc_dpfg.a21;
    const double g02 = a00 * dpfc_dpfg.a02 + a01 * dpfc_dpfg.a12 + a02 * dpfc_dpfg.a22;
    const double g10 = a10 * dpfc_dpfg.a00 + a11 * dpfc_dpfg.a10 + a12 * dpfc_dpfg.a20;
    const double g11 = a10 * dpfc_dpfg.a01 + a11 * dpfc_dpfg.a11 + a12 * dpfc_dpfg.a21;
    const double g12 = a10 * dpfc_dpfg.a02 + a11 * dpfc_dpfg.a12 + a12 * dpfc_dpfg.a22;
    // H_f = dz_dpfg * dpfg_dlambda (:389)
    const double *dl = hq + 12;
    rd[RO_HF + 0] = g00 * dl[0] + g01 * dl[3] + g02 * dl[6];
    rd[RO_HF + 1] = g00 * dl[1] + g01 * dl[4] + g02 * dl[7];
    rd[RO_HF + 2] = g00 * dl[2] + g01 * dl[5] + g02 * dl[8];
    rd[RO_HF + 3] = g10 * dl[0] + g11 * dl[3] + g12 * dl[6];
    rd[RO_HF + 4] = g10 * dl[1] + g11 * dl[4] + g12 * dl[7];
    rd[RO_HF + 5] = g10 * dl[2] + g11 * dl[5] + g12 * dl[8];
    // clone block = dz_dpfc * [R_ItoC skew(p_FinIi), -dpfc_dpfg]  (:377-392)
    const M3 Rsk = mul(R_ItoC, skew_x(p_FinIi));
    rd[RO_CLONE + 0] = a00 * Rsk.a00 + a01 * Rsk.a10 + a02 * Rsk.a20;
    rd[RO_CLONE + 1] = a00 * Rsk.a01 + a01 * Rsk.a11 + a02 * Rsk.a21;
    rd[RO_CLONE + 2] = a00 * Rsk.a02 + a01 * Rsk.a12 + a02 * Rsk.a22;
    rd[RO_CLONE + 3] = -g00, rd[RO_CLONE + 4] = -g01, rd[RO_CLONE + 5] = -g02;
    rd[RO_CLONE + 6] = a10 * Rsk.a00 + a11 * Rsk.a10 + a12 * Rsk.a20;
    rd[RO_CLONE + 7] = a10 * Rsk.a01 + a11 * Rsk.a11 + a12 * Rsk.a21;
    rd[RO_CLONE + 8] = a10 * Rsk.a02 + a11 * Rsk.a12 + a12 * Rsk.a22;
    rd[RO_CLONE + 9] = -g10, rd[RO_CLONE + 10] = -g11, rd[RO_CLONE + 11] = -g12;
    // extrinsics: dz_dpfc * [skew(p_FinCi - p_IinC), I]  (:404-413)
    {
      const M3 sk = skew_x(p_FinCi - p_IinC);
      rd[RO_CPOSE + 0] = a00 * sk.a00 + a01 * sk.a10 + a02 * sk.a20;
      rd[RO_CPOSE + 1] = a00 * sk.a01 + a01 * sk.a11 + a02 * sk.a21;
      rd[RO_CPOSE + 2] = a00 * sk.a02 + a01 * sk.a12 + a02 * sk.a22;
      rd[RO_CPOSE + 3] = a00, rd[RO_CPOSE + 4] = a01, rd[RO_CPOSE + 5] = a02;
      rd[RO_CPOSE + 6] = a10 * sk.a00 + a11 * sk.a10 + a12 * sk.a20;
      rd[RO_CPOSE + 7] = a10 * sk.a01 + a11 * sk.a11 + a12 * sk.a21;
      rd[RO_CPOSE + 8] = a10 * sk.a02 + a11 * sk.a12 + a12 * sk.a22;
      rd[RO_CPOSE + 9] = a10, rd[RO_CPOSE + 10] = a11, rd[RO_CPOSE + 11] = a12;
    }
    // intrinsics (:416-418)
#pragma unroll
    for (int s = 0; s < 16; s++) rd[RO_CINTR + s] = dze[s];
    if (relative) { // representation extras (:396-398): dz_dpfg * H_anc, dz_dpfg * H_calib
      const double *Ha = hq + 21, *Hc = hq + 39;
#pragma unroll
      for (int s = 0; s < 6; s++) {
        rd[RO_ANC + s] = g00 * Ha[s] + g01 * Ha[6 + s] + g02 * Ha[12 + s];
        rd[RO_ANC + 6 + s] = g10 * Ha[s] + g11 * Ha[6 + s] + g12 * Ha[12 + s];
        rd[RO_ACAL + s] = g00 * Hc[s] + g01 * Hc[6 + s] + g02 * Hc[12 + s];
        rd[RO_ACAL + 6 + s] = g10 * Hc[s] + g11 * Hc[6 + s] + g12 * Hc[12 + s];
      }
    }
}

// Householder QR of H_f (2m x 3, stored in the row store) by ONE wavefront -> V (n x 3, unit lower trapezoidal), hq[0..2] = tau,
// hq[3..8] = T of the compact WY form Q = I - V T V^T, hq[58..60] = diag(R1)   (role of UpdaterHelper.cpp:426-454).
// Only the first nproj columns are reflected (a column that stays is a Jacobian column of the output stage).
__device__ __forceinline__ void sys_hf_householder(double *rows, int RS, double *V, double *hq, int n, int nproj, int lane) {
#pragma unroll
  for (int k = 0; k < 3; k++) {
    // column k, rows k..n-1 (H_f element (row, k) lives at rows[(row>>1)*RS + 3*(row&1) + k])
    double sig = 0.0;
    for (int r = k + 1 + lane; r < n; r += 64) {
      const double x = rows[(size_t)(r >> 1) * RS + RO_HF + 3 * (r & 1) + k];
      sig = fma(x, x, sig);
    }
    sig = wave_sum(sig);
    const double alpha = rows[(size_t)(k >> 1) * RS + RO_HF + 3 * (k & 1) + k];
    double beta = alpha, tau = 0.0, scale = 0.0;
    if (sig > 2.2250738585072014e-308 && k < nproj) { // a column that stays (the depth of a single-depth landmark): H_k = I
      beta = sqrt(alpha * alpha + sig);
      if (alpha >= 0.0) beta = -beta;
      scale = 1.0 / (alpha - beta);
      tau = (beta - alpha) / beta;
    }
    // v_k
    for (int r = lane; r < n; r += 64) {
      double v = 0.0;
      if (r == k) v = 1.0;
      else if (r > k) v = rows[(size_t)(r >> 1) * RS + RO_HF + 3 * (r & 1) + k] * scale;
      V[(size_t)r * 3 + k] = v;
    }
    if (lane == 0) hq[k] = tau, hq[58 + k] = beta; // beta_k = R1[k][k]
    // apply H_k to the remaining PROJECTED columns of H_f (a column that stays is a Jacobian column of the output stage: untouched here)
    for (int c = k + 1; c < nproj; c++) {
      double w = 0.0;
      for (int r = k + lane; r < n; r += 64) w = fma(V[(size_t)r * 3 + k], rows[(size_t)(r >> 1) * RS + RO_HF + 3 * (r & 1) + c], w);
      w = wave_sum(w) * tau;
      for (int r = k + lane; r < n; r += 64) {
        double *x = &rows[(size_t)(r >> 1) * RS + RO_HF + 3 * (r & 1) + c];
        *x = fma(-w, V[(size_t)r * 3 + k], *x);
      }
    }
  }
  // T of the compact WY form Q = I - V T V^T (forward, column-wise)
  double v01 = 0, v02 = 0, v12 = 0;
  for (int r = lane; r < n; r += 64) {
    const double a = V[(size_t)r * 3], b = V[(size_t)r * 3 + 1], c = V[(size_t)r * 3 + 2];
    v01 = fma(a, b, v01), v02 = fma(a, c, v02), v12 = fma(b, c, v12);
  }
  v01 = wave_sum(v01), v02 = wave_sum(v02), v12 = wave_sum(v12);
  if (lane == 0) {
    const double t0 = hq[0], t1 = hq[1], t2 = hq[2];
    const double T00 = t0, T11 = t1, T22 = t2;
    const double T01 = -t1 * (T00 * v01);
    const double T02 = -t2 * (T00 * v02 + T01 * v12);
    const double T12 = -t2 * (T11 * v12);
    hq[3] = T00, hq[4] = T01, hq[5] = T02, hq[6] = T11, hq[7] = T12, hq[8] = T22;
  }
}

#ifndef OVG_TU_FEATY // (a non-template kernel: defined in the library's main translation unit only, see ovgpu_featy_tu.hip)
// RG: the feature's Jacobian records in the global workspace SysParams::rows_ws instead of LDS (long tracks at many columns; round 5)
template <bool RG> __global__ void __launch_bounds__(SYS_NT) k_system_t(SysParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int tid = threadIdx.x;
  const int D = p.D, LD = p.LD, N = p.N, RS = p.row_stride;
  // ---- LDS carve
  int *minfo = reinterpret_cast<int *>(smem);
  double *rows_lds = reinterpret_cast<double *>(smem + (((size_t)p.m_max * 8 * sizeof(int) + 15) & ~(size_t)15));
  double *rows = RG ? p.rows_ws + (size_t)blockIdx.x * p.m_max * RS : rows_lds;
  double *V = RG ? rows_lds : rows_lds + (size_t)p.m_max * RS;
  double *hq = V + (size_t)2 * p.m_max * 3; // [0..2] tau, [3..11] T, [12..20] dpfg_dlambda, [21..38] H_anc, [39..56] H_calib, [57..] flags
  double *Tch = hq + 64;
  double *S_lds = Tch + (size_t)2 * SYS_RCM * D;

#ifdef SYS_PROFILE
  long long sys_tacc[10] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}, sys_tlast = clock64();
#define SYS_T(i) { __syncthreads(); const long long tn = clock64(); sys_tacc[i] += tn - sys_tlast; sys_tlast = tn; }
#else
#define SYS_T(i)
#endif
  for (int slot = p.f_begin + blockIdx.x; slot < p.f_end; slot += gridDim.x) {
    // longest tracks first: workgroups are handed out in blockIdx order, so the tail of the launch is made of short features
    const int f = p.order ? p.order[slot] : slot;
    const int m0 = p.meas_offsets[f];
    const int m = p.meas_offsets[f + 1] - m0;
    const int64_t orow0 = p.row_off[f];
    const int n_out = (int)(p.row_off[f + 1] - orow0); // 2m - 3 (0 when m < 2)
    const int status_in = p.status[f];
    __syncthreads(); // previous feature's LDS fully consumed

    if (status_in != OVGPU_FEAT_USED) {
      // failed before the gate: its rows of the stacked system are zero
      for (int64_t e = tid; e < (int64_t)n_out * LD; e += SYS_NT) p.Hbig[orow0 * LD + e] = 0.0;
      if (p.init && tid == 0) p.init_flag[0] = 0;
      continue;
    }
    const int n = 2 * m;
    double *S = (m <= p.m_lds_max) ? S_lds : (p.ws + (size_t)blockIdx.x * p.ws_stride);

    // ------------------------------------------------------------------
    SYS_T(0)
    // (a0) representation Jacobian — UpdaterHelper.cpp:32-190 (once per feature)
    // ------------------------------------------------------------------
    V3 p_FinG = load_v3(p.p_FinG + 3 * f);
    const bool slam = p.slam != 0; // UpdaterSLAM::update: the feature is a landmark of the state
    // Columns of H_f that are projected out (nproj) / kept as the landmark's state columns (lm_size, starting at lm_off):
    //   MSCKF feature, delayed init          nproj 3, no landmark column
    //   3-dof SLAM landmark                  nproj 0, columns 0..2                                  (UpdaterSLAM.cpp:381-383)
    //   ANCHORED_INVERSE_DEPTH_SINGLE        nproj 2 (the bearing is marginalised), column 2 = depth (UpdaterSLAM.cpp:371-379)
    // SLAM: the representation is the LANDMARK's (UpdaterSLAM.cpp:336-341: SLAM landmarks and ArUco corners of different representations
    // are stacked in one update, :427-447); the single depth takes the Jacobians of the MSCKF inverse depth (:338-341)
    const int lm_id = slam ? p.feat_lm[f] : -1, lm_col = slam ? p.feat_lmcol[f] : -1, lm_cov = slam ? p.feat_lmcov[f] : -1;
    const int lrep = (slam && p.lm_rep) ? p.lm_rep[lm_id] : -1;
    const int rep_f = lrep < 0 ? p.opt.feat_rep : (lrep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE ? OVGPU_REP_ANCHORED_MSCKF_INVERSE_DEPTH : lrep);
    const bool relative = rep_is_relative(rep_f);
    const int lm_size = slam ? (lrep < 0 ? p.lm_size : (lrep == OVGPU_REP_ANCHORED_INVERSE_DEPTH_SINGLE ? 1 : 3)) : 0, lm_off = 3 - lm_size, nproj = 3 - lm_size;
    // per-feature measurement noise and gate multiplier (UpdaterSLAM's ArUco options, UpdaterSLAM.cpp:227-232, :392-409): the
    // rows leave the kernel scaled by sigma / sigma_f, so that the stacked system keeps ONE isotropic noise level (a QR of the
    // stack only preserves R = sigma^2 I)
    const double sig2_f = p.feat_sigma ? p.feat_sigma[f] * p.feat_sigma[f] : p.opt.sigma_pix_sq;
    const double mult_f = p.feat_chi2mult ? p.feat_chi2mult[f] : p.opt.chi2_multipler;
    const double oscale = p.feat_sigma ? sqrt(p.opt.sigma_pix_sq) / p.feat_sigma[f] : 1.0;
    V3 p_FinG_fej = slam ? load_v3(p.p_fej + 3 * f) : p_FinG; // fej == value for MSCKF features (UpdaterMSCKF.cpp:186-194)
    int anchor_cam = -1, anchor_clone = -1;
    if (relative) {
      // MSCKF / delayed init: the anchor of the triangulation; SLAM: the landmark's (UpdaterSLAM.cpp:345-348)
      const int ac = slam ? p.feat_anchor[f] : p.meas_cc[p.anchor_meas[f]];
      anchor_cam = ac >> 10, anchor_clone = ac & 1023;
      const V3 p_FinA = load_v3(p.p_FinA + 3 * f);
      const M3 R_ItoC = load_m3(p.tab_cam + 12 * anchor_cam);
      const V3 p_IinC = load_v3(p.tab_cam + 12 * anchor_cam + 9);
      const M3 R_GtoI = load_m3(p.tab_clone + 24 * anchor_clone);
      const V3 p_IinG = load_v3(p.tab_clone + 24 * anchor_clone + 9);
      p_FinG = mulT(R_GtoI, mulT(R_ItoC, p_FinA - p_IinC)) + p_IinG; // UpdaterHelper.cpp:274
      p_FinG_fej = p_FinG;                                           // :279-283: the "best" estimate, p_FinA_fej is not used
    }
    if (tid == 0) {
      double *dl = hq + 12;
      if (rep_f == OVGPU_REP_GLOBAL_3D) {
        dl[0] = 1, dl[1] = 0, dl[2] = 0, dl[3] = 0, dl[4] = 1, dl[5] = 0, dl[6] = 0, dl[7] = 0, dl[8] = 1;
      } else if (rep_f == OVGPU_REP_GLOBAL_FULL_INVERSE_DEPTH) {
        inv_depth_jac(p.opt.do_fej ? p_FinG_fej : p_FinG, dl); // UpdaterHelper.cpp:46 (fej == value for MSCKF features)
      } else {
        // anchored (UpdaterHelper.cpp:84-189)
        anchored_rep_jacobian(rep_f, p.opt.do_fej, p.tab_cam + 12 * anchor_cam, p.tab_clone + 24 * anchor_clone, load_v3(p.p_FinA + 3 * f), dl,
                              hq + 21, hq + 39);
      }
    }
    __syncthreads();

    // ------------------------------------------------------------------
    SYS_T(1)
    // (a) per-measurement sparse Jacobian rows — UpdaterHelper.cpp:314-421
    // ------------------------------------------------------------------
    for (int i = tid; i < m; i += SYS_NT) sys_measurement_rows(p, m0 + i, p_FinG, p_FinG_fej, relative, hq, minfo + 8 * i, rows + (size_t)i * RS);
    __syncthreads();

    const int anc_ccol = relative ? p.clone_col[anchor_clone] : -1;
    const int anc_pcol = (relative && p.opt.do_calib_pose) ? p.calib_col[anchor_cam] : -1;
    const int anc_ccov = anc_ccol >= 0 ? p.col_cov[anc_ccol] : -1;
    const int anc_pcov = anc_pcol >= 0 ? p.col_cov[anc_pcol] : -1;

    // ------------------------------------------------------------------
    SYS_T(2)
    // (c) chi2 gate: S0 = H P H^T + sigma^2 I (packed lower) with RHS rows [res ; H_f^T]
    // ------------------------------------------------------------------
    // P entries this thread (= column c of T) keeps across measurements: the 14 calibration rows of a camera are
    // shared by all of that camera's measurements, the anchor rows by the whole feature — loaded once, not per row
    int cache_c = -1, cache_cam = -1;
    double pcp[6] = {0, 0, 0, 0, 0, 0}, pci[8] = {0, 0, 0, 0, 0, 0, 0, 0}, pac[6] = {0, 0, 0, 0, 0, 0}, pap[6] = {0, 0, 0, 0, 0, 0};
    double plm[3] = {0, 0, 0}; // SLAM: the landmark's 3 rows of P
    for (int i0 = 0; i0 < m; i0 += SYS_RCM) {
      const int mc = min(SYS_RCM, m - i0);
      // phase 1: T = H P for the chunk's 2*mc rows, all D columns
      for (int c = tid; c < D; c += SYS_NT) {
        const double *Pc = p.P + p.col_cov[c];
        // the clone rows of every measurement of the chunk: all loads in flight before the first FMA
        double pcl[SYS_RCM][6];
#pragma unroll
        for (int ii = 0; ii < SYS_RCM; ii++) {
          const double *Pr = Pc + (size_t)minfo[8 * (i0 + min(ii, mc - 1)) + 5] * N;
#pragma unroll
          for (int s = 0; s < 6; s++) pcl[ii][s] = Pr[(size_t)s * N];
        }
        if (c != cache_c) {
          cache_c = c, cache_cam = -1;
          if (anc_ccov >= 0) {
#pragma unroll
            for (int s = 0; s < 6; s++) pac[s] = Pc[(size_t)(anc_ccov + s) * N];
          }
          if (anc_pcov >= 0) {
#pragma unroll
            for (int s = 0; s < 6; s++) pap[s] = Pc[(size_t)(anc_pcov + s) * N];
          }
          if (lm_cov >= 0) {
#pragma unroll
            for (int s = 0; s < 3; s++) plm[s] = s >= lm_off ? Pc[(size_t)(lm_cov + s - lm_off) * N] : 0.0; // indexed by H_f column
          }
        }
#pragma unroll
        for (int ii = 0; ii < SYS_RCM; ii++) {
          if (ii < mc) {
            const int *mi = minfo + 8 * (i0 + ii);
            const double *rd = rows + (size_t)(i0 + ii) * RS;
            if (mi[0] != cache_cam) {
              cache_cam = mi[0];
              if (mi[6] >= 0) {
#pragma unroll
                for (int s = 0; s < 6; s++) pcp[s] = Pc[(size_t)(mi[6] + s) * N];
              }
              if (mi[7] >= 0) {
#pragma unroll
                for (int s = 0; s < 8; s++) pci[s] = Pc[(size_t)(mi[7] + s) * N];
              }
            }
            double t0 = 0.0, t1 = 0.0;
#pragma unroll
            for (int s = 0; s < 6; s++) t0 = fma(rd[RO_CLONE + s], pcl[ii][s], t0), t1 = fma(rd[RO_CLONE + 6 + s], pcl[ii][s], t1);
            if (mi[6] >= 0) {
#pragma unroll
              for (int s = 0; s < 6; s++) t0 = fma(rd[RO_CPOSE + s], pcp[s], t0), t1 = fma(rd[RO_CPOSE + 6 + s], pcp[s], t1);
            }
            if (mi[7] >= 0) {
#pragma unroll
              for (int s = 0; s < 8; s++) t0 = fma(rd[RO_CINTR + s], pci[s], t0), t1 = fma(rd[RO_CINTR + 8 + s], pci[s], t1);
            }
            if (anc_ccov >= 0) {
#pragma unroll
              for (int s = 0; s < 6; s++) t0 = fma(rd[RO_ANC + s], pac[s], t0), t1 = fma(rd[RO_ANC + 6 + s], pac[s], t1);
            }
            if (anc_pcov >= 0) {
#pragma unroll
              for (int s = 0; s < 6; s++) t0 = fma(rd[RO_ACAL + s], pap[s], t0), t1 = fma(rd[RO_ACAL + 6 + s], pap[s], t1);
            }
            if (lm_cov >= 0) {
#pragma unroll
              for (int s = 0; s < 3; s++) t0 = fma(rd[RO_HF + s], plm[s], t0), t1 = fma(rd[RO_HF + 3 + s], plm[s], t1);
            }
            Tch[(size_t)(2 * ii) * D + c] = t0;
            Tch[(size_t)(2 * ii + 1) * D + c] = t1;
          }
        }
      }
      __syncthreads();
      SYS_T(9)
      // phase 2: S0[r][q] = T[r] . H[q]  for r in the chunk, q <= r.  Thread = (column q of S0, row parity): the sparse
      // row H[q] and its column offsets sit in registers for all rows of the chunk, so the only LDS traffic of the inner
      // loop is T[r] at addresses known up front (no index load -> address -> value chains).
      const int r0 = 2 * i0, nr = 2 * mc;
      {
        const int tq = tid & (SYS_NT / 2 - 1), tp = tid / (SYS_NT / 2);
        for (int q = tq; q < n && q <= r0 + nr - 1; q += SYS_NT / 2) {
          const int qi = q >> 1, qa = q & 1;
          const int *mi = minfo + 8 * qi;
          const double *rd = rows + (size_t)qi * RS;
          const int c_cl = mi[2], c_po = mi[3], c_in = mi[4];
          double hcl[6], hpo[6], hin[8], han[6], hac[6], hlm[3];
#pragma unroll
          for (int k = 0; k < 3; k++) hlm[k] = (lm_col >= 0 && k >= lm_off) ? rd[RO_HF + 3 * qa + k] : 0.0; // indexed by H_f column
#pragma unroll
          for (int k = 0; k < 6; k++) hcl[k] = rd[RO_CLONE + 6 * qa + k];
#pragma unroll
          for (int k = 0; k < 6; k++) hpo[k] = c_po >= 0 ? rd[RO_CPOSE + 6 * qa + k] : 0.0;
#pragma unroll
          for (int k = 0; k < 8; k++) hin[k] = c_in >= 0 ? rd[RO_CINTR + 8 * qa + k] : 0.0;
#pragma unroll
          for (int k = 0; k < 6; k++) han[k] = anc_ccol >= 0 ? rd[RO_ANC + 6 * qa + k] : 0.0;
#pragma unroll
          for (int k = 0; k < 6; k++) hac[k] = anc_pcol >= 0 ? rd[RO_ACAL + 6 * qa + k] : 0.0;
          for (int rr = tp; rr < nr; rr += 2) {
            const int r = r0 + rr;
            if (q > r) continue;
            const double *Tr = Tch + (size_t)rr * D;
            double sv = (q == r) ? sig2_f : 0.0;
#pragma unroll
            for (int k = 0; k < 6; k++) sv = fma(Tr[c_cl + k], hcl[k], sv);
            if (c_po >= 0) {
#pragma unroll
              for (int k = 0; k < 6; k++) sv = fma(Tr[c_po + k], hpo[k], sv);
            }
            if (c_in >= 0) {
#pragma unroll
              for (int k = 0; k < 8; k++) sv = fma(Tr[c_in + k], hin[k], sv);
            }
            if (anc_ccol >= 0) {
#pragma unroll
              for (int k = 0; k < 6; k++) sv = fma(Tr[anc_ccol + k], han[k], sv);
            }
            if (anc_pcol >= 0) {
#pragma unroll
              for (int k = 0; k < 6; k++) sv = fma(Tr[anc_pcol + k], hac[k], sv);
            }
            if (lm_col >= 0) {
#pragma unroll
              for (int k = 0; k < 3; k++) sv = fma(Tr[lm_col + max(k - lm_off, 0)], hlm[k], sv);
            }
            S[sidx(r, q, n)] = sv;
          }
        }
      }
      __syncthreads();
      SYS_T(3)
    }
    // right-hand sides: row n = res, rows n+1..n+3 = H_f columns
    for (int q = tid; q < n; q += SYS_NT) {
      const double *rd = rows + (size_t)(q >> 1) * RS;
      const int qa = q & 1;
      S[sidx(n, q, n)] = rd[RO_RES + qa];
      S[sidx(n + 1, q, n)] = rd[RO_HF + 3 * qa + 0];
      S[sidx(n + 2, q, n)] = rd[RO_HF + 3 * qa + 1];
      S[sidx(n + 3, q, n)] = nproj == 3 ? rd[RO_HF + 3 * qa + 2] : 0.0; // only the projected columns of H_f enter the statistic
    }
    __syncthreads();
    SYS_T(4)
    // Blocked right-looking Cholesky of the (n + 4) x n lower trapezoid; the 4 extra rows end as (L^-1 b)^T.
    // Per block of 8 columns: wave 0 factors the (rows x 8) panel in registers (row r of the panel in lane
    // r & 63, pivots and multipliers broadcast with v_readlane: no LDS round trip, no workgroup barrier inside the
    // panel), then all waves apply the rank-8 update to the trailing trapezoid.  2 barriers per 8 columns.
    {
      constexpr int CB = 8; // block width; the panel routine handles n + 4 <= 512 rows
      const int lane = tid & 63, wv = tid >> 6;
      for (int kb = 0; kb < n; kb += CB) {
        const int nb = min(CB, n - kb);
        if (wv == 0) {
          if (n + 4 - kb <= 128) gate_chol_panel<2>(S, n, kb, nb, lane);
          else if (n + 4 - kb <= 512) gate_chol_panel<8>(S, n, kb, nb, lane);
          else { // a track of more than 254 observations: the first 512 rows of the trapezoid, then the rest with the recorded pivots
            double pinv[8], pmul[64];
            gate_chol_panel<8, true>(S, n, kb, nb, lane, pinv, pmul);
            for (int row0 = kb + 512; row0 < n + 4; row0 += 512) gate_chol_panel_rest<8>(S, n, kb, nb, lane, row0, pinv, pmul);
          }
        }
        __syncthreads();
        SYS_T(0)
        // trailing update: S[i][j] -= sum_l L[i][kb + l] L[j][kb + l]   for kb + nb <= j <= min(i, n - 1), i < n + 4.
        // Register tile of 4 rows per thread: the 8 multipliers of row j are read from LDS once for 4 rows (the update is
        // LDS-bandwidth bound: 9 reads per element with one row per thread, 3.25 with four).
        {
          const int ti = tid >> 4, tj = tid & 15;
          for (int i0 = kb + nb + ti; i0 < n + 4; i0 += 64) {
            double li[4][CB];
            int jm[4]; // last column of each row, -1 for a row past the end
#pragma unroll
            for (int u = 0; u < 4; u++) {
              const int i = i0 + 16 * u;
              jm[u] = i < n + 4 ? min(i, n - 1) : -1;
#pragma unroll
              for (int l2 = 0; l2 < CB; l2++) li[u][l2] = (jm[u] >= 0 && l2 < nb) ? S[sidx(i, kb + l2, n)] : 0.0;
            }
            const int jall = max(max(jm[0], jm[1]), max(jm[2], jm[3]));
            for (int j = kb + nb + tj; j <= jall; j += 16) {
              double lj[CB];
#pragma unroll
              for (int l2 = 0; l2 < CB; l2++) lj[l2] = l2 < nb ? S[sidx(j, kb + l2, n)] : 0.0;
#pragma unroll
              for (int u = 0; u < 4; u++) {
                if (j <= jm[u]) {
                  const size_t e = sidx(i0 + 16 * u, j, n);
                  double acc = S[e];
#pragma unroll
                  for (int l2 = 0; l2 < CB; l2++) acc = fma(-li[u][l2], lj[l2], acc);
                  S[e] = acc;
                }
              }
            }
          }
        }
        __syncthreads();
        SYS_T(4)
      }
    }
    SYS_T(5)
    // chi2 = |y_r|^2 - g^T G^-1 g,  y_r = L^-1 res, Y_f = L^-1 H_f, G = Y_f^T Y_f, g = Y_f^T y_r
    if (tid < 64) {
      double a = 0, G00 = 0, G01 = 0, G02 = 0, G11 = 0, G12 = 0, G22 = 0, g0 = 0, g1 = 0, g2 = 0;
      for (int j = tid; j < n; j += 64) {
        const double yr = S[sidx(n, j, n)], y0 = S[sidx(n + 1, j, n)], y1 = S[sidx(n + 2, j, n)], y2 = S[sidx(n + 3, j, n)];
        a = fma(yr, yr, a);
        G00 = fma(y0, y0, G00), G01 = fma(y0, y1, G01), G02 = fma(y0, y2, G02);
        G11 = fma(y1, y1, G11), G12 = fma(y1, y2, G12), G22 = fma(y2, y2, G22);
        g0 = fma(y0, yr, g0), g1 = fma(y1, yr, g1), g2 = fma(y2, yr, g2);
      }
      a = wave_sum(a);
      G00 = wave_sum(G00), G01 = wave_sum(G01), G02 = wave_sum(G02), G11 = wave_sum(G11), G12 = wave_sum(G12), G22 = wave_sum(G22);
      g0 = wave_sum(g0), g1 = wave_sum(g1), g2 = wave_sum(g2);
      // two projected columns: the third row / column of G is zero, an identity entry keeps the 3 x 3 solve well posed
      const M3 G{G00, G01, G02, G01, G11, G12, G02, G12, nproj == 3 ? G22 : 1.0};
      const V3 g{g0, g1, g2};
      const V3 x = colpiv_qr_solve3(G, g);
      // 3-dof SLAM landmark: a state variable, the gate is on all n rows (UpdaterSLAM.cpp:390-405)
      const double chi2 = nproj == 0 ? a : a - dot(g, x);
      // StateHelper::initialize gates the 2m-3 projected rows against the quantile of the res.rows() it was handed: 2m, or
      // 2m-2 when the bearing was projected out before (StateHelper.cpp:466, UpdaterSLAM.cpp:181-196)
      const int dof = p.init ? n - p.init_dof_less : n - nproj;
      const double thr = mult_f * p.chi2_table[min(dof, p.chi2_table_len - 1)]; // UpdaterMSCKF.cpp:216-222
      if (tid == 0) {
        p.chi2[f] = chi2;
        p.chi2_thresh[f] = thr;
        const bool reject = chi2 > thr; // :225
        hq[57] = reject ? 1.0 : 0.0;
        if (reject) p.status[f] = OVGPU_FEAT_CHI2_REJECTED;
        else if (p.rows_used) atomicAdd(p.rows_used, n_out);
        if (p.init) p.init_flag[0] = reject ? 0 : 1;
      }
    }
    __syncthreads();
    if (hq[57] != 0.0) {
      for (int64_t e = tid; e < (int64_t)n_out * LD; e += SYS_NT) p.Hbig[orow0 * LD + e] = 0.0;
      continue;
    }

    // ------------------------------------------------------------------
    SYS_T(6)
    // (b) Householder QR of H_f (2m x 3) -> V, tau, T   (role of UpdaterHelper.cpp:426-454)
    // ------------------------------------------------------------------
    if (nproj > 0 && tid < 64) sys_hf_householder(rows, RS, V, hq, n, nproj, tid);
    __syncthreads();

    // ------------------------------------------------------------------
    SYS_T(7)
    // (d) stack: thread-per-column, rows 3..n-1 of Q^T [H_x | res] -> Hbig   (UpdaterMSCKF.cpp:237-255)
    // ------------------------------------------------------------------
    // Whitened output (p.Lw != nullptr, the Gram route): the rows leave as Q^T [H_x L | res] with P_DD = L L^T the Cholesky
    // factor of the prior block of the involved variables (L = U1^T of k_ekf.h).  The update works on T = I + (H L)^T (H L) / s^2;
    // forming H L row by row BEFORE the Gram accumulation keeps every rounding error relative to the prior: along the
    // unobservable directions (H x = 0 only by cancellation across columns, and the prior is LARGE exactly there) the rows of
    // H L vanish to eps |H| |L| and their Gram contribution to eps^2, where U1 (H^T H) U1^T would carry eps |H|^2 |U1|^2
    // (measured: |dP| / |P| 5e-8 at a 10 m gauge sigma, against 3e-13 for this form and 6e-13 for the reference's own S = H P H^T + R).
    const bool whiten = p.Lw != nullptr;
    {
      const double T00 = hq[3], T01 = hq[4], T02 = hq[5], T11 = hq[6], T12 = hq[7], T22 = hq[8];
      double *wv = Tch; // [3][LD]  V^T [H_x | res] before the whitening (the T chunk is free after the gate)
      for (int c = tid; c < LD; c += SYS_NT) {
        int kind = COL_RESIDUAL, var = 0, sub = 0;
        if (c < D) kind = p.col_kind[c], var = p.col_var[c], sub = p.col_sub[c];
        const bool anc_hit_c = (kind == COL_CLONE && var == anchor_clone && relative);
        const bool anc_hit_p = (kind == COL_CALIB_POSE && var == anchor_cam && relative && p.opt.do_calib_pose);
        // branch-free element of the sparse Jacobian: every load address is known up front, so the compiler can keep
        // several rows in flight (the branchy form was LDS-latency bound: 2 dependent reads per element)
        const int kclone = (kind == COL_CLONE) ? var : -1000;
        const int kcam = (kind == COL_CALIB_POSE || kind == COL_CALIB_INTR) ? var : -1000; // such a column exists only for a calibrated camera
        const bool whole = kind == COL_RESIDUAL || (kind == COL_LANDMARK && var == lm_id); // columns every row of the feature touches
        const int off0 = kind == COL_RESIDUAL ? RO_RES
                                              : (kind == COL_CLONE ? RO_CLONE + sub
                                                                   : (kind == COL_CALIB_POSE ? RO_CPOSE + sub : (kind == COL_CALIB_INTR ? RO_CINTR + sub : RO_HF + sub)));
        const int astr = kind == COL_RESIDUAL ? 1 : (kind == COL_CALIB_INTR ? 8 : (kind == COL_LANDMARK ? 3 : 6));
        auto hval = [&](int i, int a) -> double {
          const int2 key = *reinterpret_cast<const int2 *>(minfo + 8 * i); // (camera, clone)
          const double *rd = rows + (size_t)i * RS;
          const double hv = rd[off0 + astr * a];
          double h = (whole || key.y == kclone || key.x == kcam) ? hv : 0.0;
          if (anc_hit_c) h += rd[RO_ANC + 6 * a + sub];
          if (anc_hit_p) h += rd[RO_ACAL + 6 * a + sub];
          return h;
        };
        if (nproj == 0) { // UpdaterSLAM.cpp:381-383, :427-447: the rows go into the stack as they are, landmark columns included
          if (whiten) continue; // written by the whitening pass below
          double *out = p.Hbig + orow0 * LD + c;
#pragma unroll 4
          for (int r = 0; r < n; r++) out[(size_t)r * LD] = oscale * hval(r >> 1, r & 1);
          continue;
        }
        // y = V^T h
        double y0 = 0, y1 = 0, y2 = 0;
#pragma unroll 4
        for (int i = 0; i < m; i++) {
          const double h0 = hval(i, 0), h1 = hval(i, 1);
          const double *v = V + (size_t)6 * i;
          y0 = fma(v[0], h0, y0), y1 = fma(v[1], h0, y1), y2 = fma(v[2], h0, y2);
          y0 = fma(v[3], h1, y0), y1 = fma(v[4], h1, y1), y2 = fma(v[5], h1, y2);
        }
        if (whiten) { // V^T (H L) = (V^T H) L: the three rows are whitened below, as one dense product
          wv[c] = y0, wv[LD + c] = y1, wv[2 * LD + c] = y2;
          continue;
        }
        // z = T^T y
        const double z0 = T00 * y0, z1 = T01 * y0 + T11 * y1, z2 = T02 * y0 + T12 * y1 + T22 * y2;
        if (p.init) { // StateHelper.cpp:445-448: the 3 rows that determine the new variable, [Hxinit | resinit]
#pragma unroll
          for (int r = 0; r < 3; r++) {
            const double *v = V + (size_t)3 * r;
            p.init_out[(size_t)r * LD + c] = oscale * (hval(r >> 1, r & 1) - (v[0] * z0 + v[1] * z1 + v[2] * z2));
          }
        }
        double *out = p.Hbig + orow0 * LD + c;
#pragma unroll 4
        for (int r = nproj; r < n; r++) {
          const double h = hval(r >> 1, r & 1);
          const double *v = V + (size_t)3 * r;
          out[(size_t)(r - nproj) * LD] = oscale * (h - (v[0] * z0 + v[1] * z1 + v[2] * z2));
        }
      }
      if (whiten) {
        __syncthreads(); // wv complete
        // thread c owns column c of [H_x L | res]: (H L)[r][c] = sum over the blocks of row r of H[r][s] L[s][c].  The access
        // pattern of T = H P above with L in place of P: consecutive threads read consecutive addresses of a row of L; the 14
        // calibration rows are shared by all measurements of a camera, the anchor / landmark rows by the whole feature.
        for (int c = tid; c < LD; c += SYS_NT) {
          const bool isres = c >= D;
          const int cq = isres ? D - 1 : c; // clamped: the residual column is not whitened, its loads are masked below
          const double *Lc = p.Lw + cq;
          double z0 = 0.0, z1 = 0.0, z2 = 0.0;
          if (nproj > 0) {
            double y0 = 0.0, y1 = 0.0, y2 = 0.0;
            if (isres) {
              y0 = wv[D], y1 = wv[LD + D], y2 = wv[2 * LD + D];
            } else {
              // L is lower triangular: rows s < c of column c are zero
              double a0 = 0.0, a1 = 0.0, a2 = 0.0, b0 = 0.0, b1 = 0.0, b2 = 0.0;
              int s = c;
              for (; s + 1 < D; s += 2) {
                const double l0 = Lc[(size_t)s * D], l1 = Lc[(size_t)(s + 1) * D];
                a0 = fma(wv[s], l0, a0), a1 = fma(wv[LD + s], l0, a1), a2 = fma(wv[2 * LD + s], l0, a2);
                b0 = fma(wv[s + 1], l1, b0), b1 = fma(wv[LD + s + 1], l1, b1), b2 = fma(wv[2 * LD + s + 1], l1, b2);
              }
              if (s < D) {
                const double l0 = Lc[(size_t)s * D];
                a0 = fma(wv[s], l0, a0), a1 = fma(wv[LD + s], l0, a1), a2 = fma(wv[2 * LD + s], l0, a2);
              }
              y0 = a0 + b0, y1 = a1 + b1, y2 = a2 + b2;
            }
            z0 = T00 * y0, z1 = T01 * y0 + T11 * y1, z2 = T02 * y0 + T12 * y1 + T22 * y2; // z = T^T y
          }
          double lcp[6] = {0, 0, 0, 0, 0, 0}, lci[8] = {0, 0, 0, 0, 0, 0, 0, 0}, lac[6] = {0, 0, 0, 0, 0, 0}, lap[6] = {0, 0, 0, 0, 0, 0};
          double llm[3] = {0, 0, 0};
          if (anc_ccol >= 0) {
#pragma unroll
            for (int s = 0; s < 6; s++) lac[s] = Lc[(size_t)(anc_ccol + s) * D];
          }
          if (anc_pcol >= 0) {
#pragma unroll
            for (int s = 0; s < 6; s++) lap[s] = Lc[(size_t)(anc_pcol + s) * D];
          }
          if (lm_col >= 0) {
#pragma unroll
            for (int s = 0; s < 3; s++) llm[s] = s >= lm_off ? Lc[(size_t)(lm_col + s - lm_off) * D] : 0.0; // indexed by H_f column
          }
          int lcam = -1;
          double *out = p.Hbig + orow0 * LD + c;
          for (int i0 = 0; i0 < m; i0 += SYS_RCM) {
            const int mc = min(SYS_RCM, m - i0);
            double lcl[SYS_RCM][6]; // the clone rows of the chunk: all loads in flight before the first FMA
#pragma unroll
            for (int ii = 0; ii < SYS_RCM; ii++) {
              const double *Lr = Lc + (size_t)minfo[8 * (i0 + min(ii, mc - 1)) + 2] * D;
#pragma unroll
              for (int s = 0; s < 6; s++) lcl[ii][s] = Lr[(size_t)s * D];
            }
#pragma unroll
            for (int ii = 0; ii < SYS_RCM; ii++) {
              if (ii < mc) {
                const int i = i0 + ii;
                const int *mi = minfo + 8 * i;
                const double *rd = rows + (size_t)i * RS;
                if (mi[0] != lcam) {
                  lcam = mi[0];
                  if (mi[3] >= 0) {
#pragma unroll
                    for (int s = 0; s < 6; s++) lcp[s] = Lc[(size_t)(mi[3] + s) * D];
                  }
                  if (mi[4] >= 0) {
#pragma unroll
                    for (int s = 0; s < 8; s++) lci[s] = Lc[(size_t)(mi[4] + s) * D];
                  }
                }
                double t0 = 0.0, t1 = 0.0;
#pragma unroll
                for (int s = 0; s < 6; s++) t0 = fma(rd[RO_CLONE + s], lcl[ii][s], t0), t1 = fma(rd[RO_CLONE + 6 + s], lcl[ii][s], t1);
                if (mi[3] >= 0) {
#pragma unroll
                  for (int s = 0; s < 6; s++) t0 = fma(rd[RO_CPOSE + s], lcp[s], t0), t1 = fma(rd[RO_CPOSE + 6 + s], lcp[s], t1);
                }
                if (mi[4] >= 0) {
#pragma unroll
                  for (int s = 0; s < 8; s++) t0 = fma(rd[RO_CINTR + s], lci[s], t0), t1 = fma(rd[RO_CINTR + 8 + s], lci[s], t1);
                }
                if (anc_ccol >= 0) {
#pragma unroll
                  for (int s = 0; s < 6; s++) t0 = fma(rd[RO_ANC + s], lac[s], t0), t1 = fma(rd[RO_ANC + 6 + s], lac[s], t1);
                }
                if (anc_pcol >= 0) {
#pragma unroll
                  for (int s = 0; s < 6; s++) t0 = fma(rd[RO_ACAL + s], lap[s], t0), t1 = fma(rd[RO_ACAL + 6 + s], lap[s], t1);
                }
                if (lm_col >= 0) {
#pragma unroll
                  for (int s = 0; s < 3; s++) t0 = fma(rd[RO_HF + s], llm[s], t0), t1 = fma(rd[RO_HF + 3 + s], llm[s], t1);
                }
                if (isres) t0 = rd[RO_RES], t1 = rd[RO_RES + 1];
                const double *v = V + (size_t)6 * i;
                const int r = 2 * i;
                if (nproj > 0) t0 -= v[0] * z0 + v[1] * z1 + v[2] * z2, t1 -= v[3] * z0 + v[4] * z1 + v[5] * z2; // V is not formed otherwise
                if (r >= nproj) out[(size_t)(r - nproj) * LD] = oscale * t0;
                if (r + 1 >= nproj) out[(size_t)(r + 1 - nproj) * LD] = oscale * t1;
              }
            }
          }
        }
      }
    }
    if (p.init && tid < 9) { // H_finit = R1 (3 x 3 upper triangular): Q^T H_f, diagonal = beta
      const int i = tid / 3, j = tid % 3;
      p.init_out[(size_t)3 * LD + tid] = oscale * (j < i ? 0.0 : (j == i ? hq[58 + i] : rows[(size_t)(i >> 1) * RS + RO_HF + 3 * (i & 1) + j]));
    }
    SYS_T(8)
  }
#ifdef SYS_PROFILE
  if (p.dbg && tid == 0 && blockIdx.x == 0)
    for (int i = 0; i < 10; i++) p.dbg[100 + i] = sys_tacc[i];
#endif
}
#endif // OVG_TU_FEATY

} // namespace ovg
