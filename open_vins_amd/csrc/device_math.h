// device_math.h — small fixed-size linear algebra, JPL quaternion helpers and the
// camera models evaluated inside the MSCKF kernels (gfx950).
//
// Everything is written for fully unrolled, register-resident use: no dynamic
// indexing of local arrays (it would go to scratch, cdna_hip_programming.md §5.4 rule 20).
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>

namespace ovg {

struct V3 {
  double x, y, z;
};
struct M3 { // row-major
  double a00, a01, a02, a10, a11, a12, a20, a21, a22;
};

__device__ __forceinline__ V3 v3(double x, double y, double z) { return V3{x, y, z}; }
__device__ __forceinline__ V3 operator+(const V3 &a, const V3 &b) { return V3{a.x + b.x, a.y + b.y, a.z + b.z}; }
__device__ __forceinline__ V3 operator-(const V3 &a, const V3 &b) { return V3{a.x - b.x, a.y - b.y, a.z - b.z}; }
__device__ __forceinline__ V3 operator*(double s, const V3 &a) { return V3{s * a.x, s * a.y, s * a.z}; }
__device__ __forceinline__ double dot(const V3 &a, const V3 &b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__device__ __forceinline__ double norm(const V3 &a) { return sqrt(dot(a, a)); }

__device__ __forceinline__ V3 mul(const M3 &A, const V3 &v) {
  return V3{A.a00 * v.x + A.a01 * v.y + A.a02 * v.z, A.a10 * v.x + A.a11 * v.y + A.a12 * v.z, A.a20 * v.x + A.a21 * v.y + A.a22 * v.z};
}
__device__ __forceinline__ V3 mulT(const M3 &A, const V3 &v) { // A^T v
  return V3{A.a00 * v.x + A.a10 * v.y + A.a20 * v.z, A.a01 * v.x + A.a11 * v.y + A.a21 * v.z, A.a02 * v.x + A.a12 * v.y + A.a22 * v.z};
}
__device__ __forceinline__ M3 mul(const M3 &A, const M3 &B) {
  M3 C;
  C.a00 = A.a00 * B.a00 + A.a01 * B.a10 + A.a02 * B.a20;
  C.a01 = A.a00 * B.a01 + A.a01 * B.a11 + A.a02 * B.a21;
  C.a02 = A.a00 * B.a02 + A.a01 * B.a12 + A.a02 * B.a22;
  C.a10 = A.a10 * B.a00 + A.a11 * B.a10 + A.a12 * B.a20;
  C.a11 = A.a10 * B.a01 + A.a11 * B.a11 + A.a12 * B.a21;
  C.a12 = A.a10 * B.a02 + A.a11 * B.a12 + A.a12 * B.a22;
  C.a20 = A.a20 * B.a00 + A.a21 * B.a10 + A.a22 * B.a20;
  C.a21 = A.a20 * B.a01 + A.a21 * B.a11 + A.a22 * B.a21;
  C.a22 = A.a20 * B.a02 + A.a21 * B.a12 + A.a22 * B.a22;
  return C;
}
__device__ __forceinline__ M3 mulABt(const M3 &A, const M3 &B) { // A * B^T
  M3 C;
  C.a00 = A.a00 * B.a00 + A.a01 * B.a01 + A.a02 * B.a02;
  C.a01 = A.a00 * B.a10 + A.a01 * B.a11 + A.a02 * B.a12;
  C.a02 = A.a00 * B.a20 + A.a01 * B.a21 + A.a02 * B.a22;
  C.a10 = A.a10 * B.a00 + A.a11 * B.a01 + A.a12 * B.a02;
  C.a11 = A.a10 * B.a10 + A.a11 * B.a11 + A.a12 * B.a12;
  C.a12 = A.a10 * B.a20 + A.a11 * B.a21 + A.a12 * B.a22;
  C.a20 = A.a20 * B.a00 + A.a21 * B.a01 + A.a22 * B.a02;
  C.a21 = A.a20 * B.a10 + A.a21 * B.a11 + A.a22 * B.a12;
  C.a22 = A.a20 * B.a20 + A.a21 * B.a21 + A.a22 * B.a22;
  return C;
}
__device__ __forceinline__ M3 transpose(const M3 &A) { return M3{A.a00, A.a10, A.a20, A.a01, A.a11, A.a21, A.a02, A.a12, A.a22}; }

// ov_core/src/utils/quat_ops.h:135-139
__device__ __forceinline__ M3 skew_x(const V3 &w) { return M3{0.0, -w.z, w.y, w.z, 0.0, -w.x, -w.y, w.x, 0.0}; }

// ov_core/src/utils/quat_ops.h:153-158   R = (2 q4^2 - 1) I - 2 q4 [q x] + 2 q q^T
__device__ __forceinline__ M3 quat_2_Rot(double qx, double qy, double qz, double qw) {
  const double s = 2.0 * qw * qw - 1.0;
  M3 R;
  R.a00 = s + 2.0 * qx * qx;
  R.a01 = 2.0 * qw * qz + 2.0 * qx * qy;
  R.a02 = -2.0 * qw * qy + 2.0 * qx * qz;
  R.a10 = -2.0 * qw * qz + 2.0 * qy * qx;
  R.a11 = s + 2.0 * qy * qy;
  R.a12 = 2.0 * qw * qx + 2.0 * qy * qz;
  R.a20 = 2.0 * qw * qy + 2.0 * qz * qx;
  R.a21 = -2.0 * qw * qx + 2.0 * qz * qy;
  R.a22 = s + 2.0 * qz * qz;
  return R;
}

__device__ __forceinline__ M3 load_m3(const double *p) { return M3{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7], p[8]}; }
__device__ __forceinline__ V3 load_v3(const double *p) { return V3{p[0], p[1], p[2]}; }
__device__ __forceinline__ void store_m3(double *p, const M3 &m) {
  p[0] = m.a00, p[1] = m.a01, p[2] = m.a02, p[3] = m.a10, p[4] = m.a11, p[5] = m.a12, p[6] = m.a20, p[7] = m.a21, p[8] = m.a22;
}
__device__ __forceinline__ void store_v3(double *p, const V3 &v) { p[0] = v.x, p[1] = v.y, p[2] = v.z; }

// ---------------------------------------------------------------------------
// wavefront (64 lanes) all-reduce: every lane ends with the sum.
// Round 4: on the DPP network instead of an xor butterfly of __shfl_xor.  A shuffle of a double is two ds_bpermute_b32 — the LDS
// path, ~100 cycles of latency each level, six dependent levels: ~1.4 k cycles per reduction, and the wavefront-per-feature kernels
// (k_triangulate: 23 reductions per Gauss-Newton pass structure, k_feat_vt, the chi2 tail of the gate) are chains of them.  DPP
// moves are plain vector instructions: pairs (quad_perm), quads, half rows (row_half_mirror), rows (row_mirror), then row_bcast15 /
// row_bcast31 carry the row sums upwards; lane 63 holds the total and two v_readlane make it wave-uniform.  The order of the
// additions is fixed (pairs, quads, eights, rows, row pairs, halves): bit-reproducible, and different from the butterfly's.
// ---------------------------------------------------------------------------
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_mov_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_bcast63(double v) {
  return __hiloint2double(__builtin_amdgcn_readlane(__double2hiint(v), 63), __builtin_amdgcn_readlane(__double2loint(v), 63));
}
__device__ __forceinline__ double wave_sum(double v) {
  v += dpp_mov_f64<0xB1, 0xF>(v);  // quad_perm [1, 0, 3, 2]
  v += dpp_mov_f64<0x4E, 0xF>(v);  // quad_perm [2, 3, 0, 1]
  v += dpp_mov_f64<0x141, 0xF>(v); // row_half_mirror
  v += dpp_mov_f64<0x140, 0xF>(v); // row_mirror: every lane of a row holds the row's sum
  v += dpp_mov_f64<0x142, 0xA>(v); // row_bcast15 into rows 1 and 3 (the other rows add 0)
  v += dpp_mov_f64<0x143, 0xC>(v); // row_bcast31 into rows 2 and 3: lane 63 = the total
  return wave_bcast63(v);
}
// (the same network; the rows a broadcast does not reach keep their value: fmax(v, v))
template <int CTRL, int ROW_MASK> __device__ __forceinline__ double dpp_mov_keep_f64(double v) {
  const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), CTRL, ROW_MASK, 0xF, false);
  const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), CTRL, ROW_MASK, 0xF, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double wave_max(double v) {
  v = fmax(v, dpp_mov_f64<0xB1, 0xF>(v));
  v = fmax(v, dpp_mov_f64<0x4E, 0xF>(v));
  v = fmax(v, dpp_mov_f64<0x141, 0xF>(v));
  v = fmax(v, dpp_mov_f64<0x140, 0xF>(v));
  v = fmax(v, dpp_mov_keep_f64<0x142, 0xA>(v));
  v = fmax(v, dpp_mov_keep_f64<0x143, 0xC>(v));
  return wave_bcast63(v);
}

// ---------------------------------------------------------------------------
// 3x3 solve with column-pivoted Householder QR — the role of
// Eigen's A.colPivHouseholderQr().solve(b) at FeatureInitializer.cpp:88,294.
// Register-only: the three columns are kept as named vectors and pivoting is
// done by conditional swaps.
// ---------------------------------------------------------------------------
__device__ __forceinline__ void swap3(V3 &a, V3 &b, bool c) {
  if (c) {
    V3 t = a;
    a = b;
    b = t;
  }
}

__device__ __forceinline__ V3 colpiv_qr_solve3(const M3 &A, const V3 &bin) {
  // columns
  V3 c0{A.a00, A.a10, A.a20}, c1{A.a01, A.a11, A.a21}, c2{A.a02, A.a12, A.a22};
  V3 b = bin;
  int p0 = 0, p1 = 1, p2 = 2; // permutation: solution index of each working column
  const double eps = 2.220446049250313e-16;
  double n0 = dot(c0, c0), n1 = dot(c1, c1), n2 = dot(c2, c2);
  const double maxn2 = fmax(n0, fmax(n1, n2));
  const double thr = maxn2 * eps * eps / 3.0; // Eigen: (max col norm * eps)^2 / rows
  int rank = 3;

  // ---- step 0: pivot = largest column
  {
    bool s1 = (n1 > n0) && (n1 >= n2);
    bool s2 = (n2 > n0) && (n2 > n1);
    swap3(c0, c1, s1);
    if (s1) { int t = p0; p0 = p1; p1 = t; }
    swap3(c0, c2, s2);
    if (s2) { int t = p0; p0 = p2; p2 = t; }
  }
  if (dot(c0, c0) < thr * 3.0) rank = 0;
  double r00, r01, r02, r11, r12, r22;
  {
    // Householder on c0 (rows 0..2)
    double tail = c0.y * c0.y + c0.z * c0.z;
    double beta, tau, e1 = 0.0, e2 = 0.0;
    if (tail <= 2.2250738585072014e-308) {
      tau = 0.0;
      beta = c0.x;
    } else {
      beta = sqrt(c0.x * c0.x + tail);
      if (c0.x >= 0.0) beta = -beta;
      const double inv = 1.0 / (c0.x - beta);
      e1 = c0.y * inv;
      e2 = c0.z * inv;
      tau = (beta - c0.x) / beta;
    }
    r00 = beta;
    // apply to c1, c2, b
    double t1 = tau * (c1.x + e1 * c1.y + e2 * c1.z);
    c1.x -= t1, c1.y -= t1 * e1, c1.z -= t1 * e2;
    double t2 = tau * (c2.x + e1 * c2.y + e2 * c2.z);
    c2.x -= t2, c2.y -= t2 * e1, c2.z -= t2 * e2;
    double tb = tau * (b.x + e1 * b.y + e2 * b.z);
    b.x -= tb, b.y -= tb * e1, b.z -= tb * e2;
  }
  // ---- step 1: pivot among the remaining two on rows 1..2
  {
    double m1 = c1.y * c1.y + c1.z * c1.z, m2 = c2.y * c2.y + c2.z * c2.z;
    bool s = m2 > m1;
    swap3(c1, c2, s);
    if (s) { int t = p1; p1 = p2; p2 = t; }
    if (rank == 3 && fmax(m1, m2) < thr * 2.0) rank = 1;
  }
  r01 = c1.x;
  r02 = c2.x;
  {
    double tail = c1.z * c1.z;
    double beta, tau, e1 = 0.0;
    if (tail <= 2.2250738585072014e-308) {
      tau = 0.0;
      beta = c1.y;
    } else {
      beta = sqrt(c1.y * c1.y + tail);
      if (c1.y >= 0.0) beta = -beta;
      e1 = c1.z / (c1.y - beta);
      tau = (beta - c1.y) / beta;
    }
    r11 = beta;
    double t2 = tau * (c2.y + e1 * c2.z);
    c2.y -= t2, c2.z -= t2 * e1;
    double tb = tau * (b.y + e1 * b.z);
    b.y -= tb, b.z -= tb * e1;
  }
  r12 = c2.y;
  r22 = c2.z;
  if (rank == 3 && r22 * r22 < thr * 1.0) rank = 2;
  // ---- back substitution on the leading rank x rank block
  double y0 = 0.0, y1 = 0.0, y2 = 0.0;
  if (rank == 3) {
    y2 = b.z / r22;
    y1 = (b.y - r12 * y2) / r11;
    y0 = (b.x - r01 * y1 - r02 * y2) / r00;
  } else if (rank == 2) {
    y1 = b.y / r11;
    y0 = (b.x - r01 * y1) / r00;
  } else if (rank == 1) {
    y0 = b.x / r00;
  }
  V3 out{0.0, 0.0, 0.0};
  // out[perm[i]] = y_i
  out.x = (p0 == 0) ? y0 : ((p1 == 0) ? y1 : y2);
  out.y = (p0 == 1) ? y0 : ((p1 == 1) ? y1 : y2);
  out.z = (p0 == 2) ? y0 : ((p1 == 2) ? y1 : y2);
  return out;
}

// sigma_max / sigma_min of a symmetric 3x3 (the triangulation normal matrix,
// FeatureInitializer.cpp:91-95) by cyclic Jacobi eigenvalue sweeps.
__device__ __forceinline__ void jacobi_rot(double &app, double &aqq, double &apq, double &arp, double &arq) {
  // annihilates apq; r is the third index
  if (fabs(apq) > 1e-300) {
    double theta = (aqq - app) / (2.0 * apq);
    double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
    double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
    app -= t * apq;
    aqq += t * apq;
    apq = 0.0;
    double nrp = c * arp - s * arq;
    double nrq = s * arp + c * arq;
    arp = nrp;
    arq = nrq;
  }
}
__device__ __forceinline__ double cond_sym3(const M3 &A) {
  double a00 = A.a00, a11 = A.a11, a22 = A.a22, a01 = A.a01, a02 = A.a02, a12 = A.a12;
  for (int sweep = 0; sweep < 12; sweep++) {
    double off = fabs(a01) + fabs(a02) + fabs(a12);
    double diag = fabs(a00) + fabs(a11) + fabs(a22);
    if (off <= 1e-18 * diag) break;
    jacobi_rot(a00, a11, a01, a02, a12); // (p,q)=(0,1), r=2: arp=a02, arq=a12
    jacobi_rot(a00, a22, a02, a01, a12); // (0,2), r=1: arp=a01(=a10), arq=a12(=a21)
    jacobi_rot(a11, a22, a12, a01, a02); // (1,2), r=0: arp=a01, arq=a02
  }
  double e0 = fabs(a00), e1 = fabs(a11), e2 = fabs(a22);
  double mx = fmax(e0, fmax(e1, e2)), mn = fmin(e0, fmin(e1, e2));
  return mx / mn;
}

// Float arithmetic that must round after every operation, as the reference's x86-64 build does.
// HIP's __fmul_rn / __fadd_rn are plain operators to the optimiser and get contracted into
// v_fma_f32 under the default -ffp-contract=fast, which rounds once instead of twice; measured on
// MI355X: 1 % of the distorted pixels came out one float ulp off.  The pragma removes the
// `contract` flag from these instructions, so they cannot be fused even after inlining.
__device__ __forceinline__ float mul_f32(float a, float b) {
#pragma clang fp contract(off)
  return a * b;
}
__device__ __forceinline__ float add_f32(float a, float b) {
#pragma clang fp contract(off)
  return a + b;
}
__device__ __forceinline__ float sub_f32(float a, float b) {
#pragma clang fp contract(off)
  return a - b;
}

// Correctly rounded float square root.  __fsqrt_rn / sqrtf lower to a bare v_sqrt_f32 (1 ulp) on
// gfx950; the reference's x86 sqrtss is IEEE-exact and its result feeds float-rounded pixels and
// GN costs, so a 1-ulp difference becomes visible.  sqrt in double (correctly rounded, 53 >= 2*24+2
// bits) followed by one rounding to float is the correctly rounded float result.
// The empty asm keeps LLVM from shrinking (float)sqrt((double)x) back to sqrtf(x).
__device__ __forceinline__ float sqrtf_rn(float x) {
  double d = (double)x;
  asm volatile("" : "+v"(d));
  return (float)sqrt(d);
}

// ---------------------------------------------------------------------------
// camera models
// ---------------------------------------------------------------------------
struct CamIntr {
  double fx, fy, cx, cy, d0, d1, d2, d3;
};
__device__ __forceinline__ CamIntr load_cam(const double *p) { return CamIntr{p[0], p[1], p[2], p[3], p[4], p[5], p[6], p[7]}; }

// CamBase::distort_d -> CamRadtan::distort_f (CamBase.h:130-135, CamRadtan.h:127-146).
// The reference evaluates this on an Eigen::Vector2f: products of two of its
// coefficients and the sqrt argument are float, the polynomial is double, and the
// pixel is rounded to float.  The explicit _rn intrinsics stop the compiler from
// contracting the float products into FMAs (which would round differently).
__device__ __forceinline__ void radtan_distort_d(const CamIntr &c, double xn, double yn, double &u, double &v) {
  const float x = (float)xn, y = (float)yn;
  const float rf = sqrtf_rn(add_f32(mul_f32(x, x), mul_f32(y, y)));
  const double r = (double)rf;
  const double r_2 = r * r;
  const double r_4 = r_2 * r_2;
  const double xd = (double)x, yd = (double)y;
  const double twoxx = (double)mul_f32(mul_f32(2.0f, x), x); // 2 * x * x in float
  const double twoyy = (double)mul_f32(mul_f32(2.0f, y), y);
  const double rad = 1.0 + c.d0 * r_2 + c.d1 * r_4;
  const double x1 = xd * rad + 2.0 * c.d2 * xd * yd + c.d3 * (r_2 + twoxx);
  const double y1 = yd * rad + c.d2 * (r_2 + twoyy) + 2.0 * c.d3 * xd * yd;
  u = (double)(float)(c.fx * x1 + c.cx);
  v = (double)(float)(c.fy * y1 + c.cy);
}

// CamEqui::distort_f through distort_d (CamEqui.h:136-158)
__device__ __forceinline__ void equi_distort_d(const CamIntr &c, double xn, double yn, double &u, double &v) {
  const float x = (float)xn, y = (float)yn;
  const float rf = sqrtf_rn(add_f32(mul_f32(x, x), mul_f32(y, y)));
  const double r = (double)rf;
  const double th = atan(r);
  const double th2 = th * th;
  const double th3 = th2 * th, th5 = th3 * th2, th7 = th5 * th2, th9 = th7 * th2;
  const double theta_d = th + c.d0 * th3 + c.d1 * th5 + c.d2 * th7 + c.d3 * th9;
  const double inv_r = (r > 1e-8) ? 1.0 / r : 1.0;
  const double cdist = (r > 1e-8) ? theta_d * inv_r : 1.0;
  const double x1 = (double)x * cdist, y1 = (double)y * cdist;
  u = (double)(float)(c.fx * x1 + c.cx);
  v = (double)(float)(c.fy * y1 + c.cy);
}

// CamRadtan::compute_distort_jacobian (CamRadtan.h:154-198): dz_dzn row-major 2x2, dz_dzeta 2x8
__device__ __forceinline__ void radtan_jacobian(const CamIntr &c, double x, double y, double *dzn, double *dze) {
  const double r = sqrt(x * x + y * y);
  const double r_2 = r * r, r_4 = r_2 * r_2;
  const double x_2 = x * x, y_2 = y * y, x_y = x * y;
  const double rad = 1.0 + c.d0 * r_2 + c.d1 * r_4;
  dzn[0] = c.fx * (rad + (2 * c.d0 * x_2 + 4 * c.d1 * x_2 * r_2) + 2 * c.d2 * y + (2 * c.d3 * x + 4 * c.d3 * x));
  dzn[1] = c.fx * (2 * c.d0 * x_y + 4 * c.d1 * x_y * r_2 + 2 * c.d2 * x + 2 * c.d3 * y);
  dzn[2] = c.fy * (2 * c.d0 * x_y + 4 * c.d1 * x_y * r_2 + 2 * c.d2 * x + 2 * c.d3 * y);
  dzn[3] = c.fy * (rad + (2 * c.d0 * y_2 + 4 * c.d1 * y_2 * r_2) + 2 * c.d3 * x + (2 * c.d2 * y + 4 * c.d2 * y));
  const double x1 = x * rad + 2 * c.d2 * x * y + c.d3 * (r_2 + 2 * x * x);
  const double y1 = y * rad + c.d2 * (r_2 + 2 * y * y) + 2 * c.d3 * x * y;
  dze[0] = x1, dze[1] = 0, dze[2] = 1, dze[3] = 0;
  dze[4] = c.fx * x * r_2, dze[5] = c.fx * x * r_4, dze[6] = 2 * c.fx * x * y, dze[7] = c.fx * (r_2 + 2 * x * x);
  dze[8] = 0, dze[9] = y1, dze[10] = 0, dze[11] = 1;
  dze[12] = c.fy * y * r_2, dze[13] = c.fy * y * r_4, dze[14] = c.fy * (r_2 + 2 * y * y), dze[15] = 2 * c.fy * x * y;
}

// CamEqui::compute_distort_jacobian (CamEqui.h:166-230)
__device__ __forceinline__ void equi_jacobian(const CamIntr &c, double x, double y, double *dzn, double *dze) {
  const double r = sqrt(x * x + y * y);
  const double th = atan(r);
  const double th2 = th * th;
  const double th3 = th2 * th, th4 = th2 * th2, th5 = th3 * th2, th6 = th4 * th2, th7 = th5 * th2, th8 = th4 * th4, th9 = th7 * th2;
  const double theta_d = th + c.d0 * th3 + c.d1 * th5 + c.d2 * th7 + c.d3 * th9;
  const double inv_r = (r > 1e-8) ? 1.0 / r : 1.0;
  const double cdist = (r > 1e-8) ? theta_d * inv_r : 1.0;
  const double dxy_dxyn = theta_d * inv_r;
  const double dxy_dr0 = -x * theta_d * inv_r * inv_r, dxy_dr1 = -y * theta_d * inv_r * inv_r;
  const double dr0 = x * inv_r, dr1 = y * inv_r;
  const double dthd_dth = 1 + 3 * c.d0 * th2 + 5 * c.d1 * th4 + 7 * c.d2 * th6 + 9 * c.d3 * th8;
  const double dth_dr = 1.0 / (r * r + 1.0);
  const double v0 = dxy_dr0 + dr0 * dthd_dth * dth_dr, v1 = dxy_dr1 + dr1 * dthd_dth * dth_dr;
  dzn[0] = c.fx * (dxy_dxyn + v0 * dr0);
  dzn[1] = c.fx * (v0 * dr1);
  dzn[2] = c.fy * (v1 * dr0);
  dzn[3] = c.fy * (dxy_dxyn + v1 * dr1);
  const double x1 = x * cdist, y1 = y * cdist;
  dze[0] = x1, dze[1] = 0, dze[2] = 1, dze[3] = 0;
  dze[4] = c.fx * x * inv_r * th3, dze[5] = c.fx * x * inv_r * th5, dze[6] = c.fx * x * inv_r * th7, dze[7] = c.fx * x * inv_r * th9;
  dze[8] = 0, dze[9] = y1, dze[10] = 0, dze[11] = 1;
  dze[12] = c.fy * y * inv_r * th3, dze[13] = c.fy * y * inv_r * th5, dze[14] = c.fy * y * inv_r * th7, dze[15] = c.fy * y * inv_r * th9;
}

} // namespace ovg
