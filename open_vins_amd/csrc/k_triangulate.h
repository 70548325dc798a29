// k_triangulate.h — pose tables + per-feature triangulation and Gauss-Newton refinement.
//
//   k_build_tables   UpdaterMSCKF.cpp:97-115  (clone-camera pose table, FeatureInitializer.h:51-82)
//   k_triangulate    FeatureInitializer.cpp:30-112 (single_triangulation), :114-195 (_1d),
//                    :197-375 (single_gaussnewton), :377-423 (compute_error)
//
// Mapping: ONE FEATURE PER WAVEFRONT.  The 64 lanes stride over the feature's measurements,
// the 3x3 normal equations / GN Hessian are summed with xor-butterfly shuffles, and every lane
// then solves the same 3x3 system redundantly, so control flow stays wave-uniform.  The
// clone-camera pose table (K*C x 12 doubles, <= 19 KB) is staged in LDS once per workgroup.
#pragma once
#include "device_math.h"
#include "ovgpu_types.h"

namespace ovg {

// ---------------------------------------------------------------------------
// tables: per clone [24]: R_GtoI(9) p_IinG(3) R_GtoI_fej(9) p_IinG_fej(3)
//         per cam   [12]: R_ItoC(9) p_IinC(3)
//         per (cam, clone) [12]: R_GtoC(9) p_CinG(3)        index k*C + c
// ---------------------------------------------------------------------------
__device__ __forceinline__ void build_tables_item(int t, int C, int K, const double *clone_qp, const double *clone_fej, const double *calib_qp,
                                                  double *tab_clone, double *tab_cam, double *tab_cc) {
  if (t < C) {
    const double *q = clone_qp + 7 * t;
    const double *qf = clone_fej + 7 * t;
    double *o = tab_clone + 24 * t;
    store_m3(o, quat_2_Rot(q[0], q[1], q[2], q[3]));
    o[9] = q[4], o[10] = q[5], o[11] = q[6];
    store_m3(o + 12, quat_2_Rot(qf[0], qf[1], qf[2], qf[3]));
    o[21] = qf[4], o[22] = qf[5], o[23] = qf[6];
  }
  if (t < K) {
    const double *q = calib_qp + 7 * t;
    double *o = tab_cam + 12 * t;
    store_m3(o, quat_2_Rot(q[0], q[1], q[2], q[3]));
    o[9] = q[4], o[10] = q[5], o[11] = q[6];
  }
  if (t < K * C) {
    const int k = t / C, c = t % C;
    const double *qc = calib_qp + 7 * k;
    const double *q = clone_qp + 7 * c;
    const M3 R_ItoC = quat_2_Rot(qc[0], qc[1], qc[2], qc[3]);
    const M3 R_GtoI = quat_2_Rot(q[0], q[1], q[2], q[3]);
    const M3 R_GtoC = mul(R_ItoC, R_GtoI);                                      // UpdaterMSCKF.cpp:106
    const V3 p = v3(q[4], q[5], q[6]) - mulT(R_GtoC, v3(qc[4], qc[5], qc[6])); // :107
    store_m3(tab_cc + 12 * t, R_GtoC);
    store_v3(tab_cc + 12 * t + 9, p);
  }
}

__global__ void k_build_tables(int C, int K, const double *__restrict__ clone_qp, const double *__restrict__ clone_fej,
                               const double *__restrict__ calib_qp, double *__restrict__ tab_clone, double *__restrict__ tab_cam,
                               double *__restrict__ tab_cc) {
  build_tables_item(blockIdx.x * blockDim.x + threadIdx.x, C, K, clone_qp, clone_fej, calib_qp, tab_clone, tab_cam, tab_cc);
}

struct RelPose {
  M3 R_AtoCi;
  V3 p_CiinA;
};

__device__ __forceinline__ RelPose rel_pose(const double *lds_cc, int cc, const M3 &R_GtoA, const V3 &p_AinG) {
  const double *t = lds_cc + 12 * cc;
  const M3 R_GtoC = load_m3(t);
  const V3 p_CinG = load_v3(t + 9);
  RelPose r;
  r.R_AtoCi = mulABt(R_GtoC, R_GtoA);          // FeatureInitializer.cpp:73
  r.p_CiinA = mul(R_GtoA, p_CinG - p_AinG);    // :75
  return r;
}

// FeatureInitializer::compute_error — FeatureInitializer.cpp:377-423 (float residual, Q3)
__device__ __forceinline__ double gn_cost(const double *lds_cc, const uint16_t *__restrict__ meas_cc, const float2 *__restrict__ uvn, int m0, int m1,
                                          int lane, const M3 &R_GtoA, const V3 &p_AinG, int C, double alpha, double beta, double rho) {
  double err = 0.0;
  for (int i = m0 + lane; i < m1; i += 64) {
    const int code = meas_cc[i];
    const RelPose rp = rel_pose(lds_cc, (code >> 10) * C + (code & 1023), R_GtoA, p_AinG);
    const V3 pA = -1.0 * mul(rp.R_AtoCi, rp.p_CiinA); // p_AinCi
    const M3 &R = rp.R_AtoCi;
    const double hi1 = R.a00 * alpha + R.a01 * beta + R.a02 + rho * pA.x;
    const double hi2 = R.a10 * alpha + R.a11 * beta + R.a12 + rho * pA.y;
    const double hi3 = R.a20 * alpha + R.a21 * beta + R.a22 + rho * pA.z;
    const float2 z = uvn[i];
    const float r0 = sub_f32(z.x, (float)(hi1 / hi3)), r1 = sub_f32(z.y, (float)(hi2 / hi3));
    const float n = sqrtf_rn(add_f32(mul_f32(r0, r0), mul_f32(r1, r1)));
    err += (double)n * (double)n;
  }
  return wave_sum(err);
}

// grid: ceil(F / w) blocks of w wavefronts = w features per workgroup (enqueue_triangulate)
__global__ void __launch_bounds__(1024) k_triangulate(TriParams p) {
  extern __shared__ __attribute__((aligned(16))) double lds_cc[]; // [K*C*12]
  for (int i = threadIdx.x; i < p.K * p.C * 12; i += blockDim.x) lds_cc[i] = p.tab_cc[i];
  __syncthreads();

  const int lane = threadIdx.x & 63;
  const int f = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (f >= p.F) return; // whole wavefront leaves together
  const int m0 = p.meas_offsets[f], m1 = p.meas_offsets[f + 1];
  const int m = m1 - m0;
  const float2 *uvn = reinterpret_cast<const float2 *>(p.uvn);

  auto finish = [&](int status, int anchor, const V3 &pA, const V3 &pG) {
    if (lane == 0) {
      p.status[f] = status;
      p.anchor_meas[f] = anchor;
      p.p_FinA[3 * f + 0] = pA.x, p.p_FinA[3 * f + 1] = pA.y, p.p_FinA[3 * f + 2] = pA.z;
      p.p_FinG[3 * f + 0] = pG.x, p.p_FinG[3 * f + 1] = pG.y, p.p_FinG[3 * f + 2] = pG.z;
    }
  };
  const double qnan = __longlong_as_double(0x7ff8000000000000LL);
  const V3 vnan{qnan, qnan, qnan};
  if (m < 2) { // UpdaterMSCKF.cpp:87-93
    finish(OVGPU_FEAT_TOO_FEW_MEAS, -1, vnan, vnan);
    return;
  }

  // ---- anchor rule (FeatureInitializer.cpp:36-46): a property of the batch, found when it was laid out (feat::k_batch_layout, k_featy.h; until round 5
  //      every wavefront scanned its feature's packed codes here, ~50 dependent loads in front of its first floating-point instruction)
  const bool seeded = p.seed_pA != nullptr; // single_gaussnewton on a caller's estimate: its anchor and position stand
  const int anchor = __builtin_amdgcn_readfirstlane(seeded ? p.seed_anchor[f] : p.anchor_pre[f]);
  if (seeded && (anchor < m0 || anchor >= m1)) {
    finish(OVGPU_FEAT_TRI_FAILED, -1, vnan, vnan);
    return;
  }
  const int acode = p.meas_cc[anchor];
  const int acc = (acode >> 10) * p.C + (acode & 1023);
  const M3 R_GtoA = load_m3(lds_cc + 12 * acc);
  const V3 p_AinG = load_v3(lds_cc + 12 * acc + 9);

  V3 p_f;
  if (seeded) {
    p_f = load_v3(p.seed_pA + 3 * f);
  } else if (!p.opt.triangulate_1d) {
    // ---- single_triangulation: A = sum Bperp^T Bperp, b = sum A_i p_CiinA   (:58-85)
    double A00 = 0, A01 = 0, A02 = 0, A11 = 0, A12 = 0, A22 = 0, b0 = 0, b1 = 0, b2 = 0;
    for (int i = m0 + lane; i < m1; i += 64) {
      const int code = p.meas_cc[i];
      const RelPose rp = rel_pose(lds_cc, (code >> 10) * p.C + (code & 1023), R_GtoA, p_AinG);
      const float2 z = uvn[i];
      V3 bi = mulT(rp.R_AtoCi, v3((double)z.x, (double)z.y, 1.0)); // :78-80
      const double inv = 1.0 / norm(bi);
      bi = v3(bi.x * inv, bi.y * inv, bi.z * inv);
      // Ai = skew(bi)^T skew(bi) = |bi|^2 I - bi bi^T  evaluated entry-wise as the reference's product does
      const double a00 = bi.z * bi.z + bi.y * bi.y, a11 = bi.z * bi.z + bi.x * bi.x, a22 = bi.y * bi.y + bi.x * bi.x;
      const double a01 = -bi.x * bi.y, a02 = -bi.x * bi.z, a12 = -bi.y * bi.z;
      A00 += a00, A01 += a01, A02 += a02, A11 += a11, A12 += a12, A22 += a22;
      const V3 &q = rp.p_CiinA;
      b0 += a00 * q.x + a01 * q.y + a02 * q.z;
      b1 += a01 * q.x + a11 * q.y + a12 * q.z;
      b2 += a02 * q.x + a12 * q.y + a22 * q.z;
    }
    A00 = wave_sum(A00), A01 = wave_sum(A01), A02 = wave_sum(A02), A11 = wave_sum(A11), A12 = wave_sum(A12), A22 = wave_sum(A22);
    b0 = wave_sum(b0), b1 = wave_sum(b1), b2 = wave_sum(b2);
    const M3 A{A00, A01, A02, A01, A11, A12, A02, A12, A22};
    p_f = colpiv_qr_solve3(A, v3(b0, b1, b2)); // :88
    const double condA = cond_sym3(A);         // :91-95
    const double pn = norm(p_f);
    if (fabs(condA) > p.opt.max_cond_number || p_f.z < p.opt.min_dist || p_f.z > p.opt.max_dist || isnan(pn)) { // :103-106
      finish(OVGPU_FEAT_TRI_FAILED, anchor, vnan, vnan);
      return;
    }
  } else {
    // ---- single_triangulation_1d: depth along the anchor bearing (:142-184)
    const float2 za = uvn[anchor];
    V3 bearing = v3((double)za.x, (double)za.y, 1.0);
    const double binv = 1.0 / norm(bearing);
    bearing = v3(bearing.x * binv, bearing.y * binv, bearing.z * binv);
    double A = 0.0, b = 0.0;
    for (int i = m0 + lane; i < m1; i += 64) {
      if (i == anchor) continue; // :160-161
      const int code = p.meas_cc[i];
      const RelPose rp = rel_pose(lds_cc, (code >> 10) * p.C + (code & 1023), R_GtoA, p_AinG);
      const float2 z = uvn[i];
      V3 bi = mulT(rp.R_AtoCi, v3((double)z.x, (double)z.y, 1.0));
      const double inv = 1.0 / norm(bi);
      bi = v3(bi.x * inv, bi.y * inv, bi.z * inv);
      const M3 Bp = skew_x(bi);
      const V3 Ba = mul(Bp, bearing);
      const V3 Bq = mul(Bp, rp.p_CiinA);
      A += dot(Ba, Ba);
      b += dot(Ba, Bq);
    }
    A = wave_sum(A), b = wave_sum(b);
    const double depth = b / A;
    p_f = v3(depth * bearing.x, depth * bearing.y, depth * bearing.z);
    if (p_f.z < p.opt.min_dist || p_f.z > p.opt.max_dist || isnan(norm(p_f))) {
      finish(OVGPU_FEAT_TRI_FAILED, anchor, vnan, vnan);
      return;
    }
  }

  V3 p_FinA = p_f;
  if (p.opt.refine_features) {
    // ---- single_gaussnewton (:197-375): LM in (alpha, beta, rho)
    double rho = 1.0 / p_FinA.z, alpha = p_FinA.x / p_FinA.z, beta = p_FinA.y / p_FinA.z;
    double lam = p.opt.init_lamda, eps = 10000.0;
    int runs = 0;
    bool recompute = true;
    double H00 = 0, H01 = 0, H02 = 0, H11 = 0, H12 = 0, H22 = 0, g0 = 0, g1 = 0, g2 = 0;
    double cost_old = gn_cost(lds_cc, p.meas_cc, uvn, m0, m1, lane, R_GtoA, p_AinG, p.C, alpha, beta, rho); // :217
    while (runs < p.opt.max_runs && lam < p.opt.max_lamda && eps > p.opt.min_dx) {                          // :227
      if (recompute) {
        H00 = H01 = H02 = H11 = H12 = H22 = g0 = g1 = g2 = 0.0;
        for (int i = m0 + lane; i < m1; i += 64) {
          const int code = p.meas_cc[i];
          const RelPose rp = rel_pose(lds_cc, (code >> 10) * p.C + (code & 1023), R_GtoA, p_AinG);
          const V3 pA = -1.0 * mul(rp.R_AtoCi, rp.p_CiinA);
          const M3 &R = rp.R_AtoCi;
          const double hi1 = R.a00 * alpha + R.a01 * beta + R.a02 + rho * pA.x; // :260-262
          const double hi2 = R.a10 * alpha + R.a11 * beta + R.a12 + rho * pA.y;
          const double hi3 = R.a20 * alpha + R.a21 * beta + R.a22 + rho * pA.z;
          const double h3sq = hi3 * hi3;
          const double j0 = (R.a00 * hi3 - hi1 * R.a20) / h3sq, j1 = (R.a01 * hi3 - hi1 * R.a21) / h3sq, j2 = (pA.x * hi3 - hi1 * pA.z) / h3sq;
          const double j3 = (R.a10 * hi3 - hi2 * R.a20) / h3sq, j4 = (R.a11 * hi3 - hi2 * R.a21) / h3sq, j5 = (pA.y * hi3 - hi2 * pA.z) / h3sq;
          const float2 z = uvn[i];
          const double r0 = (double)sub_f32(z.x, (float)(hi1 / hi3)), r1 = (double)sub_f32(z.y, (float)(hi2 / hi3)); // :273-275
          g0 += j0 * r0 + j3 * r1, g1 += j1 * r0 + j4 * r1, g2 += j2 * r0 + j5 * r1; // :282
          H00 += j0 * j0 + j3 * j3, H01 += j0 * j1 + j3 * j4, H02 += j0 * j2 + j3 * j5; // :283
          H11 += j1 * j1 + j4 * j4, H12 += j1 * j2 + j4 * j5, H22 += j2 * j2 + j5 * j5;
        }
        H00 = wave_sum(H00), H01 = wave_sum(H01), H02 = wave_sum(H02), H11 = wave_sum(H11), H12 = wave_sum(H12), H22 = wave_sum(H22);
        g0 = wave_sum(g0), g1 = wave_sum(g1), g2 = wave_sum(g2);
      }
      const double dl = 1.0 + lam; // :289-292
      const M3 Hl{H00 * dl, H01, H02, H01, H11 * dl, H12, H02, H12, H22 * dl};
      const V3 dx = colpiv_qr_solve3(Hl, v3(g0, g1, g2)); // :294
      const double cost = gn_cost(lds_cc, p.meas_cc, uvn, m0, m1, lane, R_GtoA, p_AinG, p.C, alpha + dx.x, beta + dx.y, rho + dx.z);
      if (cost <= cost_old && (cost_old - cost) / cost_old < p.opt.min_dcost) { // :306
        alpha += dx.x, beta += dx.y, rho += dx.z;
        eps = 0;
        break;
      }
      if (cost <= cost_old) { // :316
        recompute = true;
        cost_old = cost;
        alpha += dx.x, beta += dx.y, rho += dx.z;
        runs++;
        lam = lam / p.opt.lam_mult;
        eps = norm(dx);
      } else {
        recompute = false;
        lam = lam * p.opt.lam_mult;
      }
    }
    p_FinA = v3(alpha / rho, beta / rho, 1.0 / rho); // :332-335

    // max baseline orthogonal to the bearing (:338-357).  The reference builds the tangent
    // plane from a Householder QR of p_FinA; the norm of the projection onto that plane is
    // basis-independent: |x|^2 - (x . p/|p|)^2, evaluated through the same Householder vector.
    const double pn = norm(p_FinA);
    double hb = sqrt(p_FinA.x * p_FinA.x + (p_FinA.y * p_FinA.y + p_FinA.z * p_FinA.z));
    if (p_FinA.x >= 0.0) hb = -hb;
    const double tail = p_FinA.y * p_FinA.y + p_FinA.z * p_FinA.z;
    double tau = 0.0, e1 = 0.0, e2 = 0.0;
    if (tail > 2.2250738585072014e-308) {
      const double inv = 1.0 / (p_FinA.x - hb);
      e1 = p_FinA.y * inv, e2 = p_FinA.z * inv;
      tau = (hb - p_FinA.x) / hb;
    }
    // Q = I - tau v v^T, v = (1, e1, e2); columns 1 and 2
    const V3 q1 = v3(-tau * e1, 1.0 - tau * e1 * e1, -tau * e2 * e1);
    const V3 q2 = v3(-tau * e2, -tau * e1 * e2, 1.0 - tau * e2 * e2);
    double base_line_max = 0.0;
    for (int i = m0 + lane; i < m1; i += 64) {
      const int code = p.meas_cc[i];
      const RelPose rp = rel_pose(lds_cc, (code >> 10) * p.C + (code & 1023), R_GtoA, p_AinG);
      const double d1 = dot(q1, rp.p_CiinA), d2 = dot(q2, rp.p_CiinA);
      base_line_max = fmax(base_line_max, sqrt(d1 * d1 + d2 * d2));
    }
    base_line_max = wave_max(base_line_max);
    if (p_FinA.z < p.opt.min_dist || p_FinA.z > p.opt.max_dist || (pn / base_line_max) > p.opt.max_baseline || isnan(pn)) { // :367-370
      finish(OVGPU_FEAT_GN_FAILED, anchor, vnan, vnan);
      return;
    }
  }
  const V3 p_FinG = mulT(R_GtoA, p_FinA) + p_AinG; // :373 / :110
  finish(OVGPU_FEAT_USED, anchor, p_FinA, p_FinG);
}

} // namespace ovg
